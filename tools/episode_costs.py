"""Whole-episode cost of one RLEPSO instance per function (bbob 1-24 and bbob-noisy 101-130), exact FDR kernels, shipped policy, the reference's stop rule: the weights of the
inter-rank partition (metabox_amd/distributed.py: EPISODE_COST_US).  One batch per function, resident rollout, stream time by HIP events / instances.
    python tools/episode_costs.py [--dims 10,30,40] [--instances 1024]  ->  one JSON object {dim: {func_id: us per instance-episode}}"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.agent import RLEPSO_Agent
from metabox_amd.config import get_config
from metabox_amd.problem.bbob import BBOB_Dataset
from metabox_amd.suite import Batch, Suite

ap = argparse.ArgumentParser(); ap.add_argument('--dims', default='10,30,40'); ap.add_argument('--instances', type=int, default=1024)
a = ap.parse_args()
out = {}
for dim in [int(x) for x in a.dims.split(',')]:
    NP = 128 if dim == 40 else 100
    ps = []
    for suite in ('bbob', 'bbob-noisy'):
        tr, te = BBOB_Dataset.get_datasets(suite, dim, 5.0)
        ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
    s = Suite(ps)
    cfg = get_config(['--problem', 'bbob', '--dim', str(dim), '--device', 'cuda']); cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
    actor = agent.actor; h1, h2 = actor.hidden_sizes()
    maxfes = 2000 * dim
    gens = (maxfes - 1) // NP + 1
    B = a.instances
    res, table = {}, None
    for k, p in enumerate(ps):
        b = Batch(s, ALGO_RLEPSO, np.full(B, k), np.arange(B, dtype=np.uint64) * 7 + 3, NP, maxfes, maxfes // 50, 50)
        if table is None:
            table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma).clone()
        ts = []
        for rep in range(2):                                   # the second episode of the same instances counts (clocks, caches)
            b.reset(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            done = 0
            while done < gens:
                n = min(50, gens - done); b.rlepso_rollout(table, n); done += n
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[int(p.func_id)] = round(ts[-1] / B * 1e3, 3)          # us per instance-episode
        b.close()
    out[dim] = res
    print(json.dumps({dim: res}), flush=True)
