"""Config 5 for the PMC passes (tools/exp/pmc.sh): four mbx_rlepso_rollout launches of TEN generations each on 8192 instances -- every k_rlepso_run<1024, 128, 40, 5, true>
dispatch of the process is a 10-generation one, so the per-dispatch counter means divide by 10 (the r05 file averaged a 2- and a 10-generation dispatch)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite
ps = []
for suite in ('bbob', 'bbob-noisy'):
    tr, te = BBOB_Dataset.get_datasets(suite, 40, 5.0)
    ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
cfg = get_config(['--problem', 'bbob', '--dim', '40', '--device', 'cuda']); cfg.agent_save_dir = None
agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', '..', 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
actor = agent.actor; h1, h2 = actor.hidden_sizes()
B = 8192
b = Batch(Suite(ps), ALGO_RLEPSO, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 3, 128, 80000, 1600, 50, early_stop=False)
table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
b.reset()
for _ in range(4):
    b.rlepso_rollout(table, 10)
torch.cuda.synchronize()
print('done', b.rollout_is_resident())
