"""TEST / MEASUREMENT INFRASTRUCTURE -- not part of the product (see oracle/__init__.py).

One worker of bench.py's `cpu_baseline` leg: whole RLEPSO episodes of the bench workload (bbob d=10 pop=100, functions
round-robin, reference stop rule) on the C oracle, with actions sampled from the actor's (mu, sigma) table the parent
process wrote to an .npy file (rows indexed by fes; action = clip(N(mu, sigma), 0, 1) like the reference's
Actor.forward, src/agent/rlepso_agent.py:38-44).  The worker imports numpy and the oracle only, so bench.py can start one
per host core.  Prints one JSON line {"steps", "episodes", "seconds"}.

    python oracle/cpu_workload.py --table /tmp/t.npy --seconds 10 --worker 3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(table, seconds, worker, np_=100, dim=10, max_fes=20000, n_log=50):
    from metabox_amd.problem.bbob import BBOB_Dataset
    from oracle import oracle
    tr, te = BBOB_Dataset.get_datasets('bbob', dim, 5.0)
    ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    ps = ps[worker % len(ps):] + ps[:worker % len(ps)]              # workers start on different functions
    cfg = oracle.make_cfg(1, np_, dim, max_fes, max_fes // n_log, n_log)
    adim = table.shape[1] // 2
    mu, sigma = table[:, :adim], table[:, adim:]
    rng = np.random.default_rng(977 + worker)
    steps, episodes, run_id = 0, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for p in ps:
            o = oracle.RlepsoOracle(p.desc(), p.bias, cfg, seed=1000 + 4096 * worker + run_id)
            s = o.reset()
            done = False
            while not done:
                k = min(int(round(s * max_fes)), table.shape[0] - 1)
                a = np.clip(mu[k] + sigma[k] * rng.standard_normal(adim, dtype=np.float32), 0, 1)
                s, _, done = o.step(a)
                steps += 1
            episodes += 1
            if time.perf_counter() - t0 >= seconds:
                break
        run_id += 1
    return {'steps': steps, 'episodes': episodes, 'seconds': time.perf_counter() - t0}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--table', required=True)
    ap.add_argument('--seconds', type=float, default=10.0)
    ap.add_argument('--worker', type=int, default=0)
    a = ap.parse_args()
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    print(json.dumps(run(np.load(a.table), a.seconds, a.worker)))
