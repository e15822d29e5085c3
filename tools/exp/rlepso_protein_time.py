"""RLEPSO on protein docking (the reference's --problem protein: D = 12, NP = 100, 9 generations per episode), one GPU's share of config 4's table (35 problems x 64 runs):
ms per generation, whole episodes through mbx_reset + mbx_rlepso_rollout.   python tools/exp/rlepso_protein_time.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.suite import Batch, Suite
from metabox_amd.utils import construct_problem_set
cfg = get_config(['--problem', 'protein', '--device', 'cuda']); cfg.agent_save_dir = None
agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
actor = agent.actor; h1, h2 = actor.hidden_sizes()
tr, te = construct_problem_set(cfg)
ps = (tr + te).data[:35]
B = 35 * 64
b = Batch(Suite(ps), ALGO_RLEPSO, np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1, 100, 1000, 200, 5)
table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
def episodes(n):
    for _ in range(n):
        b.reset(); b.rlepso_rollout(table, 9)
episodes(2); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); episodes(10); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 90)
print(json.dumps({'lib': os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so')), 'resident': b.rollout_is_resident(), 'ms_per_generation': sorted(ts)[1] * 1e3,
                  'final_cost_mean': float(b.results()['cost'][:, -1].mean())}))
