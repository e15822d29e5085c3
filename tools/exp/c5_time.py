"""Config 5's bench leg alone (8192 instances, 54 functions, D = 40, NP = 128, resident launches of 20 generations, median of 5):  MBX_LIB=... python tools/exp/c5_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite
ps = []
for suite in ('bbob', 'bbob-noisy'):
    tr, te = BBOB_Dataset.get_datasets(suite, 40, 5.0)
    ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
s = Suite(ps)
cfg = get_config(['--problem', 'bbob', '--dim', '40', '--device', 'cuda']); cfg.agent_save_dir = None
agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
actor = agent.actor; h1, h2 = actor.hidden_sizes()
B = 8192
b = Batch(s, ALGO_RLEPSO, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 3, 128, 80000, 1600, 50, early_stop=False)
table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
b.reset(); b.rlepso_rollout(table, 4); torch.cuda.synchronize()
ts = []; G = int(os.environ.get("C5_GENS", 20))
for _ in range(5):
    t0 = time.perf_counter(); b.rlepso_rollout(table, G); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / G)
print(json.dumps({'lib': os.path.basename(os.environ.get('MBX_LIB', 'libmbx.so')), 'config5_ms_per_generation': round(sorted(ts)[2] * 1e3, 4), 'all': [round(t * 1e3, 4) for t in ts]}))
