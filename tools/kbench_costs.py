#!/usr/bin/env python
"""Per-function cost table for the inter-rank partition (metabox_amd/distributed.py: COST_US): one RLEPSO generation of a batch that holds a single
BBOB function kind, fixed horizon, resident rollout, at D = 10 / NP = 100 (configs 1-2) and D = 40 / NP = 128 (config 5).
   python tools/kbench_costs.py [--dims 10,40]  ->  one JSON object {dim: {kind: us per instance-generation x 1e3}}"""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite

ap = argparse.ArgumentParser(); ap.add_argument('--dims', default='10,40'); ap.add_argument('--gens', type=int, default=12); ap.add_argument('--kinds', default='')
a = ap.parse_args()
out = {}
for dim in [int(x) for x in a.dims.split(',')]:
    NP, B = (100, 4096) if dim <= 16 else (128 if dim == 40 else 100, 2048)
    tr, te = BBOB_Dataset.get_datasets('bbob', dim, 5.0)
    ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    s = Suite(ps)
    cfg = get_config(['--problem', 'bbob', '--dim', str(dim), '--device', 'cuda']); cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(os.path.dirname(__file__), '..', 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
    actor = agent.actor; h1, h2 = actor.hidden_sizes()
    res = {}
    for k, p in enumerate(ps):
        if a.kinds and str(p.kind) not in a.kinds.split(','): continue
        b = Batch(s, ALGO_RLEPSO, np.full(B, k), np.arange(B, dtype=np.uint64) + 3, NP, 2000 * dim, 40 * dim, 50, early_stop=False)
        table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
        b.reset(); b.rlepso_rollout(table, 2); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.rlepso_rollout(table, a.gens); e1.record(); torch.cuda.synchronize()
        res[p.kind] = round(e0.elapsed_time(e1) / a.gens / B * 1e6, 2)           # ns per instance-generation
        b.close()
    out[dim] = res
print(json.dumps(out))
