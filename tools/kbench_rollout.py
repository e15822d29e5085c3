# timing: resident rollout vs per-generation act_step on the bench workload
import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from metabox_amd.suite import Batch, Suite
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.problem.bbob import BBOB_Dataset
cfg = bench.make_config(); cfg.device = 'cuda'
agent = bench.load_agent(cfg, 'cuda')
actor = agent.actor
h1, h2 = actor.hidden_sizes()
tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = Suite(ps)
pidx = np.arange(B) % len(ps); seeds = np.arange(B, dtype=np.uint64) + 1000
def ev():
    return torch.cuda.Event(enable_timing=True)
for mode in ('step', 'run20', 'run50', 'run199', 'step'):
    b = Batch(s, ALGO_RLEPSO, pidx, seeds, 100, 20000, 400, 50)
    table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    b.reset(); torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    e0.record()
    if mode == 'step':
        for g in range(199): b.act_step(table)
    else:
        n = int(mode[3:]); g = 0
        while g < 199:
            b.rlepso_rollout(table, min(n, 199 - g)); g += n
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    r = b.results(); steps = int(r['steps'].sum().item())
    print(json.dumps({'mode': mode, 'B': B, 'ms_episode': ms, 'us_per_gen': ms * 1e3 / 199, 'env_steps': steps, 'env_steps_per_s': steps / ms * 1e3,
                      'cost_sum': float(r['cost'][:, -1].sum().item())}), flush=True)
    b.close()
