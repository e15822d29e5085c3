"""Kernel micro-benchmark of the resident rollout (mbx_rlepso_rollout) against one launch per generation (mbx_rlepso_act_step) on the
bench workload (RLEPSO bbob d=10 pop=100, 24 functions round-robin, exported policy):
   python tools/kbench_rollout.py [--B 4096] [--fids 1,16] [--episode]        (MBX_LIB selects an ablation build)
default: generations 1..40 of an episode (every instance live) as ONE launch / as 40 launches; --episode: whole episodes in chunks."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from metabox_amd.suite import Batch, Suite
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.problem.bbob import BBOB_Dataset

ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=4096)
ap.add_argument('--fids', default='')
ap.add_argument('--gens', type=int, default=40)
ap.add_argument('--episode', action='store_true')
ap.add_argument('--modes', default='')
a = ap.parse_args()
cfg = bench.make_config(); cfg.device = 'cuda'
agent = bench.load_agent(cfg, 'cuda')
actor = agent.actor
h1, h2 = actor.hidden_sizes()
tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
if a.fids:
    want = [int(x) for x in a.fids.split(',')]
    ps = [p for p in ps if p.func_id in want]
B = a.B
s = Suite(ps)
pidx = np.arange(B) % len(ps); seeds = np.arange(B, dtype=np.uint64) + 1000
lib = os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so'))
modes = a.modes.split(',') if a.modes else (['step', 'run20', 'run50', 'run199', 'step'] if a.episode else ['step', 'run', 'run', 'step'])
for mode in modes:
    b = Batch(s, ALGO_RLEPSO, pidx, seeds, 100, 20000, 400, 50)
    table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    b.reset(); torch.cuda.synchronize()
    G = 199 if a.episode else a.gens
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if mode == 'step':
        for g in range(G): b.act_step(table)
    else:
        n = int(mode[3:]) if len(mode) > 3 else G
        g = 0
        while g < G:
            b.rlepso_rollout(table, min(n, G - g)); g += n
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    r = b.results(); steps = int(r['steps'].sum().item())
    print(json.dumps({'lib': lib, 'mode': mode, 'fids': a.fids or 'all', 'B': B, 'gens': G, 'us_per_gen': round(ms * 1e3 / G, 1), 'env_steps': steps,
                      'env_steps_per_s': round(steps / ms * 1e3), 'cost_sum': float(r['cost'][:, -1].sum().item())}), flush=True)
    b.close()
