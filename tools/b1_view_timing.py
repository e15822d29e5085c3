#!/usr/bin/env python
"""Cost of the single-instance compatibility view (PBO_Env + RLEPSO_Optimizer.init_population / update: the reference's protocol, one
instance, host round trip every generation) next to the reference's own 2.5 ms per generation (432 env-steps/s on one
host core).   python tools/b1_view_timing.py [--episodes 5]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metabox_amd.config import get_config
from metabox_amd.environment import PBO_Env
from metabox_amd.optimizer import RLEPSO_Optimizer
from metabox_amd.problem.bbob import BBOB_Dataset

ap = argparse.ArgumentParser(); ap.add_argument('--episodes', type=int, default=5)
a = ap.parse_args()
cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0)
opt = RLEPSO_Optimizer(cfg)
rs = np.random.RandomState(0)
np.random.seed(0)
resets, steps, n_steps = [], [], 0
for ep in range(a.episodes + 1):
    p = tr.data[ep % len(tr.data)]
    env = PBO_Env(p, opt)
    t0 = time.perf_counter(); env.reset(); t1 = time.perf_counter()
    done, n = False, 0
    while not done:
        _, _, done = env.step(rs.uniform(0, 1, 35).astype(np.float32)); n += 1
    t2 = time.perf_counter()
    if ep:                                   # the first episode pays library load / allocation
        resets.append(t1 - t0); steps.append((t2 - t1) / n); n_steps += n
print(json.dumps({'path': 'B = 1 compatibility view: PBO_Env.reset / step through RLEPSO_Optimizer (one instance, host round trip per generation)',
                  'ms_per_env_step': float(np.mean(steps)) * 1e3, 'env_steps_per_s': 1. / float(np.mean(steps)), 'ms_per_reset': float(np.mean(resets)) * 1e3,
                  'episodes': a.episodes, 'env_steps': n_steps, 'reference_ms_per_env_step': 2.5,
                  'note': 'reference figure: SURVEY.md section 6 (432 env-steps/s, one host core of the build container)'}))
