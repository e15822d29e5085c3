"""A ONE-GPU proxy for the 8-GPU scaling curve (it is NOT a SCALE line: no second GPU, no RCCL).

An epoch on N GPUs ends with its slowest rank: instances are independent, shards are contiguous cost-weighted cuts of the (problem x run) table
(`metabox_amd.distributed.partition_bounds`, the enumeration of /root/reference/src/tester.py:190-202), the step path has no collective.  So the scaling
efficiency a node can reach is bounded by how evenly the eight shards' kernel times come out.  This tool builds the FULL tables of BASELINE.json configs
2 / 4 / 5 (8 x 4096, 17 920 and 65 536 instances), cuts them with partition_bounds(.., 8) exactly as Tester.run_pairs does, runs the eight shards one after
another on cuda:0 through the resident routes the bench times (whole episodes), and reports per shard: instances, env-steps, stream time by HIP events; and
    balance = mean / max of the shard times   (= the projected 8-GPU efficiency of the compute part; the end-of-epoch all-gather is 3.5 MB per GPU)
It also runs a SAMPLE of global rows again in one unsharded batch and checks that the gathered table holds bit-identical rows for them (results depend on
(problem, Philox key) only, never on the shard an instance ran in).

    python tools/shard_balance.py [--configs 2,4,5] [--world 8] [--out profiles/r06_shard_balance.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metabox_amd import distributed as mdist                                     # noqa: E402
from metabox_amd._abi import ALGO_RLEPSO                                         # noqa: E402
from metabox_amd.suite import Batch, Suite                                       # noqa: E402


def table_of(n_problems, total):
    """problem-major (problem, run) table like instance_table, truncated to `total` rows (the last problem gets the remainder)."""
    runs = -(-total // n_problems)
    p, r = mdist.instance_table(n_problems, runs)
    return p[:total], r[:total]


def rlepso_case(dim, np_, total, suites):
    from metabox_amd.agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.problem.bbob import BBOB_Dataset
    ps = []
    for suite in suites:
        tr, te = BBOB_Dataset.get_datasets(suite, dim, 5.0)
        ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
    cfg = get_config(['--problem', 'bbob', '--dim', str(dim), '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
    actor = agent.actor
    h1, h2 = actor.hidden_sizes()
    maxfes = 2000 * dim
    gens = (maxfes - np_ + np_ - 1) // np_ + 1                                   # an episode never runs longer (every generation bills >= NP evaluations)
    suite = Suite(ps)
    state = {}

    def run_shard(pidx, seeds):
        b = Batch(suite, ALGO_RLEPSO, pidx, seeds, np_, maxfes, maxfes // 50, 50)
        assert b.rollout_is_resident()
        if 'table' not in state:
            state['table'] = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma).clone()
        table = state['table']
        b.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.rlepso_rollout(table, gens)                                           # the whole episode in ONE launch, like RLEPSO_Agent.rollout_batch; instances that finish leave it
        e1.record()
        torch.cuda.synchronize()
        rows = mdist.pack_rows(b.results()).clone()
        assert bool((b.done != 0).all())
        b.close()
        return e0.elapsed_time(e1), rows

    return ps, total, run_shard


def ddqn_case(total_runs=64):
    from metabox_amd.agent import DE_DDQN_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import DE_DDQN_Optimizer
    from metabox_amd.utils import construct_problem_set
    cfg = get_config(['--problem', 'protein', '--device', 'cuda'])
    cfg.agent_save_dir = None
    torch.manual_seed(0)
    agent = DE_DDQN_Agent(cfg).to('cuda')
    packed = agent.packed_weights()
    tr, te = construct_problem_set(cfg)
    ps = (tr + te).data
    suite = Suite(ps)

    def run_shard(pidx, seeds):
        env = BatchedPBO_Env(ps, DE_DDQN_Optimizer(cfg), pidx, seeds, early_stop=False, suite=suite)
        env.reset()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        with torch.no_grad():
            for _ in range(cfg.maxFEs - 100):                                   # one trial vector per step: maxFEs - NP steps to the budget
                env.step(env.batch.ddqn_qnet(packed))
        e1.record()
        torch.cuda.synchronize()
        rows = mdist.pack_rows(env.results()).clone()
        assert bool((env.batch.done != 0).all())
        env.batch.close()
        return e0.elapsed_time(e1), rows

    return ps, len(ps) * total_runs, run_shard


def measure(name, ps, total, run_shard, world, sample):
    pidx, runs = table_of(len(ps), total)
    gid = np.arange(total, dtype=np.uint64)
    seeds = mdist.philox_seed(runs, gid)
    bounds = mdist.partition_bounds(ps, pidx, world)
    shards, rows = [], []
    run_shard(pidx[:min(256, total)], seeds[:min(256, total)])                   # warm-up: module load, table build, clocks
    for r in range(world):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        ms, rw = run_shard(pidx[lo:hi], seeds[lo:hi])
        rows.append(rw)
        steps = int(rw[:, -1].sum().item())
        shards.append({'rank': r, 'instances': hi - lo, 'problems': int(len(np.unique(pidx[lo:hi]))), 'env_steps': steps, 'stream_ms': ms,
                       'env_steps_per_s': steps / (ms * 1e-3)})
    table = torch.cat(rows, dim=0)                                               # what gather_rows assembles: rank order == global-id order
    assert table.shape[0] == total
    # a sample of global rows once more, in ONE unsharded batch, in a different batch order
    pick = np.sort(np.random.RandomState(7).choice(total, min(sample, total), replace=False))[::-1].copy()
    _, again = run_shard(pidx[pick], seeds[pick])
    identical = bool(torch.equal(again.cpu(), table[torch.from_numpy(pick.astype(np.int64))].cpu()))
    t = np.array([s['stream_ms'] for s in shards])
    eq = mdist.cost_partition(np.ones(total), world)                             # what an equal-count split would have cut
    out = {'config': name, 'instances': total, 'world': world, 'bounds': [int(x) for x in bounds], 'equal_count_bounds': [int(x) for x in eq],
           'shards': shards, 'shard_ms_max': float(t.max()), 'shard_ms_mean': float(t.mean()),
           'max_over_mean': float(t.max() / t.mean()), 'projected_efficiency': float(t.mean() / t.max()),
           'sample_rows_rerun_unsharded': int(len(pick)), 'sample_rows_bit_identical': identical,
           'one_gpu_serial_ms': float(t.sum())}
    print(json.dumps({k: out[k] for k in ('config', 'instances', 'shard_ms_max', 'shard_ms_mean', 'max_over_mean', 'projected_efficiency', 'sample_rows_bit_identical')}), flush=True)
    assert identical, 'a row depends on the shard it ran in'
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--configs', default='2,4,5')
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--sample', type=int, default=512)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r06_shard_balance.json'))
    args = ap.parse_args()
    want = {int(c) for c in args.configs.split(',')}
    res, t0 = [], time.time()
    if 2 in want:
        res.append(measure('config 2 x 8: RLEPSO bbob d=10 pop=100, 32 768 instances (24 functions x runs), whole episodes, mbx_rlepso_rollout',
                           *rlepso_case(10, 100, 8 * 4096, ('bbob',)), args.world, args.sample))
    if 4 in want:
        res.append(measure('config 4: DE-DDQN protein docking d=12 pop=100, 17 920 instances = 280 problems x 64 runs, 900 steps, mbx_ddqn_qnet + mbx_step',
                           *ddqn_case(), args.world, args.sample))
    if 5 in want:
        res.append(measure('config 5: RLEPSO mixed suite (24 bbob + 30 noisy) d=40 pop=128, 65 536 instances, whole episodes, mbx_rlepso_rollout',
                           *rlepso_case(40, 128, 65536, ('bbob', 'bbob-noisy')), args.world, args.sample))
    doc = {'what': 'one-GPU proxy of the 8-GPU scaling curve: the eight cost-weighted shards of each full table run one after another on cuda:0; an epoch on 8 GPUs '
                   'ends with the slowest shard, so projected_efficiency = mean / max of the shard times (compute part only; no RCCL, no second GPU was involved)',
           'partition': 'metabox_amd.distributed.partition_bounds (COST_NS weights), the cut Tester.run_pairs makes', 'device': torch.cuda.get_device_name(0),
           'wall_s': time.time() - t0, 'results': res}
    with open(args.out, 'w') as f:
        json.dump(doc, f, indent=1)
    print('wrote', args.out)


if __name__ == '__main__':
    main()
