#!/usr/bin/env python
"""BASELINE.json config 5, one GPU's share: RLEPSO on the mixed suite (24 bbob + 30 bbob-noisy) at D = 40 with NP = 128, 8192
instances (65536 / 8), policy fused (act + step in one launch).   python tools/kbench_config5.py [--B 8192] [--steps 20]"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite

ap = argparse.ArgumentParser(); ap.add_argument('--B', type=int, default=8192); ap.add_argument('--steps', type=int, default=20); ap.add_argument('--fids', default='')
a = ap.parse_args()
ps = []
for suite in ('bbob', 'bbob-noisy'):
    tr, te = BBOB_Dataset.get_datasets(suite, 40, 5.0)
    ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
if a.fids:
    want = [int(x) for x in a.fids.split(',')]
    ps = [p for p in ps if p.func_id in want]
s = Suite(ps)
cfg = get_config(['--problem', 'bbob', '--dim', '40', '--device', 'cuda']); cfg.agent_save_dir = None
agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
actor = agent.actor; h1, h2 = actor.hidden_sizes()
b = Batch(s, ALGO_RLEPSO, np.arange(a.B) % len(ps), np.arange(a.B, dtype=np.uint64) + 3, 128, 80000, 1600, 50, early_stop=False)
table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
b.reset()
for _ in range(3): b.act_step(table)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): b.act_step(table)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({'fids': a.fids, 'path': 'config 5: RLEPSO mixed suite D=40 NP=128, %d instances on one GPU, one launch per generation' % a.B, 'ms_per_step': dt * 1e3, 'env_steps_per_s': a.B / dt}))
# the same generations as ONE resident launch (mbx_rlepso_rollout)
b.rlepso_rollout(table, 2)
torch.cuda.synchronize(); t0 = time.perf_counter()
b.rlepso_rollout(table, a.steps)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({'fids': a.fids, 'path': 'config 5, resident: %d generations in one mbx_rlepso_rollout launch' % a.steps, 'ms_per_step': dt * 1e3, 'env_steps_per_s': a.B / dt}))
