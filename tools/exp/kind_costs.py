"""Per-function cost of a generation at the large geometries, for the launch-order weights (upload_launch_order in mbx.hip):
   python tools/exp/kind_costs.py     -> RLEPSO D = 40 NP = 128 (resident, 20 generations per launch) for the 24 bbob functions, LDE D = 30 NP = 50 per noisy group"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite

tr, te = BBOB_Dataset.get_datasets('bbob', 40, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
s = Suite(ps)
cfg = get_config(['--problem', 'bbob', '--dim', '40', '--device', 'cuda']); cfg.agent_save_dir = None
agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
actor = agent.actor; h1, h2 = actor.hidden_sizes()
B = 4096
out = {}
for k, p in enumerate(ps):
    b = Batch(s, ALGO_RLEPSO, np.full(B, k), np.arange(B, dtype=np.uint64) + 3, 128, 80000, 1600, 50, early_stop=False)
    table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    b.reset(); b.rlepso_rollout(table, 4); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); b.rlepso_rollout(table, 10); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10)
    out[p.func_id] = round(sorted(ts)[1] * 1e6, 1)
    b.close()
print(json.dumps({'rlepso_d40_np128_us_per_generation_of_4096': out}))
