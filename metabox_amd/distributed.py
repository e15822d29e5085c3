"""Instance sharding across the GPUs of one node and end-of-epoch result aggregation.

Instances are independent (nothing is shared between (problem x run) pairs,
src/tester.py:190-202), so the step path has NO collective.  Each rank owns a contiguous block of global instance
ids; the Philox key of an instance is a function of its global id only, so the gathered result is identical for any
number of ranks.  The only communication is one all-gather of the per-instance result rows
([n_local, n_logpoint+1 + 3] float64: cost curve, fes, return, steps) at the end of an epoch — RCCL over xGMI when
the process group is "nccl", gloo in the CPU tests.
"""
import os

import numpy as np
import torch


def _single_process(group=None):
    """True when there is nothing to communicate with: no process group, or a group of ONE rank -- unless MBX_FORCE_COLLECTIVES=1 asks for the
    collectives to be issued anyway (tests/test_gpu_shards.py pushes real result rows and gradients through a world-size-1 RCCL group, so that the
    first RCCL call this code ever makes is not the 8-GPU run's)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and os.environ.get('MBX_FORCE_COLLECTIVES', '0') != '1'


def shard_range(n_total, rank, world):
    """Contiguous block partition: rank r gets [lo, hi).  Sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def instance_table(n_problems, runs, sort_by_problem=True):
    """Global (problem_idx, run) table of a test/rollout epoch: problems x runs, problem-major like the reference's
    nested loops (tester.py:190-196), which also keeps a shard's problems few and its constants GPU-local."""
    p = np.repeat(np.arange(n_problems, dtype=np.int32), runs)
    r = np.tile(np.arange(runs, dtype=np.int64), n_problems)
    return p, r


# ns per instance-generation of the RLEPSO generation kernel by BBOB function kind (1..24), measured on one MI355X with tools/kbench_costs.py
# (a batch that holds one kind only, fixed horizon, resident rollout; re-measured at the round-4 head with the per-kind kernel bodies: profiles/r04e_kind_costs_ns.json; D = 30: before them) at D = 10 / NP = 100, D = 30 / NP = 100 and
# D = 40 / NP = 128.  Only the RATIOS matter: they weight the inter-rank partition below and -- the same numbers, rounded -- the launch order inside a batch
# (upload_launch_order in csrc/mbx.hip).  Noisy functions cost what their base kind costs (the noise is O(NP) per generation).
COST_NS = {
    10: {1: 26.01, 2: 29.65, 3: 32.93, 4: 29.37, 5: 20.05, 6: 24.52, 7: 25.34, 8: 23.1, 9: 22.82, 10: 26.95, 11: 26.83, 12: 26.39, 13: 22.29, 14: 25.01, 15:
         31.54, 16: 32.72, 17: 31.07, 18: 31.06, 19: 24.38, 20: 24.31, 21: 39.61, 22: 28.99, 23: 32.19, 24: 25.47},
    30: {1: 84.86, 2: 100.99, 3: 112.79, 4: 102.7, 5: 73.29, 6: 81.22, 7: 90.62, 8: 78.32, 9: 78.39, 10: 91.78, 11: 91.66, 12: 93.63, 13: 76.91, 14: 84.95, 15:
         112.8, 16: 118.18, 17: 113.17, 18: 113.63, 19: 83.24, 20: 80.5, 21: 144.75, 22: 97.63, 23: 104.35, 24: 86.63},
    40: {1: 158.85, 2: 178.7, 3: 207.74, 4: 198.05, 5: 133.42, 6: 156.57, 7: 188.54, 8: 153.28, 9: 152.78, 10: 178.33, 11: 178.44, 12: 183.99, 13: 149.94, 14:
         165.65, 15: 228.47, 16: 234.17, 17: 225.47, 18: 228.29, 19: 160.72, 20: 152.6, 21: 301.57, 22: 237.76, 23: 204.9, 24: 168.18},
}

# us per WHOLE EPISODE of one RLEPSO instance by function id (bbob 1-24, bbob-noisy 101-130), measured on one MI355X with tools/episode_costs.py at the round-6 head
# (D = 10: one-function batches of 4096 instances, which fill the chip like a shard does -- with 1024-instance batches the expensive functions look 1.6x dearer than they are
# in a full launch; D = 30 / 40: 1024 instances, one workgroup per CU either way)
# (exact FDR kernels, shipped policy, the reference's stop rule: Sphere and linear slope stop early, the noisy functions carry their noise models, the step ellipsoid's
# plateaus re-initialise).  An epoch of Tester / rollout runs whole episodes, so THESE weight the inter-rank partition (relative_cost); COST_NS above (per generation, by
# kind) remains the fallback for functions outside the table.  Only the ratios matter.  (tools/shard_balance.py: with the per-generation weights the eight shards of
# 8 x config 2 came out at max / mean = 1.096 -- the shards that hold Sphere / linear slope finish early -- and config 5 at 1.035.)
EPISODE_COST_US = {
    10: {1: 2.06, 2: 5.49, 3: 6.599, 4: 6.06, 5: 0.882, 6: 4.785, 7: 4.8, 8: 4.582, 9: 4.534, 10: 5.31, 11: 5.267, 12: 5.705, 13: 5.18, 14: 5.018, 15: 6.403,
         16: 6.37, 17: 6.179, 18: 6.174, 19: 4.902, 20: 5.029, 21: 6.566, 22: 6.418, 23: 6.187, 24: 4.828, 101: 2.084, 102: 2.589, 103: 4.86, 104: 4.928, 105:
         4.98, 106: 5.01, 107: 4.783, 108: 4.85, 109: 4.864, 110: 4.926, 111: 4.977, 112: 4.997, 113: 5.225, 114: 5.277, 115: 5.539, 116: 5.649, 117: 5.709,
         118: 5.72, 119: 5.368, 120: 5.409, 121: 5.437, 122: 6.474, 123: 6.524, 124: 6.546, 125: 5.222, 126: 5.289, 127: 5.298, 128: 8.12, 129: 8.19, 130:
         8.244},
    30: {1: 49.839, 2: 57.49, 3: 65.793, 4: 61.253, 5: 18.096, 6: 50.424, 7: 87.195, 8: 48.966, 9: 48.867, 10: 57.466, 11: 57.787, 12: 59.006, 13: 48.687, 14:
         53.691, 15: 71.181, 16: 72.724, 17: 73.554, 18: 70.707, 19: 52.503, 20: 51.597, 21: 94.011, 22: 68.812, 23: 64.733, 24: 54.919, 101: 50.834, 102:
         50.966, 103: 51.019, 104: 51.478, 105: 51.826, 106: 52.148, 107: 50.466, 108: 50.751, 109: 50.965, 110: 51.432, 111: 51.655, 112: 51.918, 113: 57.088,
         114: 57.126, 115: 69.928, 116: 60.076, 117: 60.395, 118: 60.718, 119: 56.352, 120: 56.582, 121: 56.877, 122: 71.95, 123: 72.264, 124: 72.417, 125:
         55.371, 126: 55.529, 127: 55.637, 128: 92.662, 129: 92.97, 130: 92.919},
    40: {1: 96.322, 2: 114.884, 3: 133.6, 4: 127.253, 5: 42.755, 6: 100.916, 7: 174.791, 8: 98.739, 9: 97.95, 10: 115.614, 11: 115.238, 12: 117.948, 13:
         96.142, 14: 106.594, 15: 147.226, 16: 149.02, 17: 145.344, 18: 145.246, 19: 104.588, 20: 97.767, 21: 192.195, 22: 151.999, 23: 130.564, 24: 109.576,
         101: 102.836, 102: 103.422, 103: 103.867, 104: 105.042, 105: 105.406, 106: 105.627, 107: 103.087, 108: 103.709, 109: 103.705, 110: 105.092, 111:
         106.204, 112: 105.789, 113: 120.459, 114: 121.348, 115: 149.978, 116: 122.797, 117: 123.647, 118: 124.065, 119: 115.076, 120: 115.811, 121: 115.873,
         122: 148.381, 123: 149.596, 124: 149.642, 125: 112.751, 126: 112.987, 127: 112.725, 128: 198.153, 129: 199.259, 130: 198.812},
}


def relative_cost(problem):
    """Predicted cost of one instance of `problem` (whole episode where EPISODE_COST_US knows the function, else one generation from COST_NS; the table of the nearest
    measured dimension: 10 / 30 / 40; a problem that knows its own relative cost -- protein docking: `relative_step_cost` -- says so; anything else costs 1).  Relative within one suite / dimension only."""
    own = getattr(problem, 'relative_step_cost', None)
    if own is not None:                                   # protein docking: the energy walks the problem's own number of close atom pairs
        return float(own())
    kind = getattr(problem, 'kind', None)
    if kind is None:
        return 1.0
    dim = getattr(problem, 'dim', 10)
    near = 10 if dim <= 20 else (30 if dim <= 35 else 40)
    fid = getattr(problem, 'func_id', None)
    ep = EPISODE_COST_US[near]
    if fid is not None and int(fid) in ep:
        return float(ep[int(fid)])
    # a function outside the table: its kind's per-generation cost, scaled to the table's units by the mean ratio of the two tables
    table = COST_NS[near]
    scale = np.mean([ep[k] / table[k] for k in table if k in ep])
    return float(table.get(int(kind), np.mean(list(table.values()))) * scale)


def cost_partition(costs, world):
    """Cost-weighted CONTIGUOUS partition of the instance table over `world` ranks: rank r gets [bounds[r], bounds[r + 1]) such that every
    rank's predicted cost is within one instance of total / world.  The table is problem-major (instance_table), so a rank still sees few
    problems, rank order == global-id order (gather_rows needs no permutation) and the Philox key of an instance is a function of its
    global id only: the gathered table is bit-identical for any number of ranks, only the cut points move.  The equal-count split this
    replaces gave the rank that owns Gallagher / Weierstrass instances ~3x the work of the one that owns Sphere / Linear slope at config 5
    ("what's missing" 3); an epoch ends with the slowest rank.
    -> int64 array of world + 1 ascending bounds, bounds[0] = 0, bounds[-1] = len(costs)."""
    c = np.asarray(costs, dtype=np.float64)
    n, world = len(c), int(world)
    if n == 0 or world <= 1:
        return np.array([0] + [n] * max(world, 1), dtype=np.int64)
    cum = np.cumsum(c)
    mid = cum - 0.5 * c                                   # an instance belongs to the rank whose quantile its midpoint falls into
    targets = cum[-1] * np.arange(1, world) / world
    inner = np.searchsorted(mid, targets, side='left')
    return np.concatenate([[0], inner, [n]]).astype(np.int64)


def partition_bounds(problems, pidx, world):
    """cost_partition over the (problem x run) table `pidx` of `problems`."""
    per_problem = np.array([relative_cost(p) for p in problems], dtype=np.float64)
    return cost_partition(per_problem[np.asarray(pidx)], world)


def philox_seed(run, global_id, epoch_salt=0):
    """64-bit Philox key of an instance: depends on (run seed, global instance id, salt) only."""
    with np.errstate(over='ignore'):                    # arithmetic modulo 2^64 is the point
        x = (np.asarray(run, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             ^ (np.asarray(global_id, dtype=np.uint64) + np.uint64(0xD1B54A32D192ED03)) * np.uint64(0xBF58476D1CE4E5B9)
             ^ np.uint64(epoch_salt) * np.uint64(0x94D049BB133111EB))
        x ^= x >> np.uint64(31)
    return x


def pack_rows(res):
    """dict of device tensors from Batch.results() -> one [n, n_logpoint+1+3] float64 tensor."""
    return torch.cat([res['cost'], res['fes'][:, None], res['return'][:, None],
                      res['steps'].to(torch.float64)[:, None]], dim=1).contiguous()


def unpack_rows(rows):
    n = rows.shape[1] - 3
    return {'cost': rows[:, :n], 'fes': rows[:, n], 'return': rows[:, n + 1], 'steps': rows[:, n + 2].to(torch.int64)}


def gather_rows(rows, n_total, group=None, bounds=None):
    """All-gather the local result rows of every rank into the global [n_total, C] table (rank order == global id
    order because shards are contiguous).  Shards differ in size (by one row for the equal-count split, by the cost ratio for
    `bounds` = cost_partition(...)): rows are padded to the maximum."""
    import torch.distributed as dist
    if _single_process(group):
        return rows
    world = dist.get_world_size(group)
    if bounds is not None:
        assert len(bounds) == world + 1 and int(bounds[-1]) == n_total
        sizes = [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]
    else:
        sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(max_n, rows.shape[1], dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def average_gradients(params, group=None, weight=None):
    """Data-parallel training: combine the gradients of `params` over all ranks with ONE flat all-reduce (the RLEPSO actor + critic are 6.9 k
    parameters, the LDE LSTM 32.6 k -- latency-bound on xGMI, nothing to bucket).

    weight: the number of samples the rank's loss was a MEAN over (live (step, instance) pairs for PPO / REINFORCE; may be 0 for a rank whose shard
    has finished).  Shards are cost-weighted (`cost_partition`), so ranks own different numbers of instances; the synchronised gradient is then
    sum_r weight_r grad_r / sum_r weight_r = the gradient of the mean loss over ALL ranks' samples, i.e. the same objective for any world size
    (the unweighted mean of per-rank means gave instances of expensive functions up to ~2x the weight of cheap ones).  The weight travels
    as one extra element of the same all-reduce.  weight=None: plain mean over ranks (equal-size mini-batches, e.g. DE-DDQN's replay samples)."""
    import torch.distributed as dist
    if _single_process(group):
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    if weight is None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= dist.get_world_size(group)
    else:
        w = torch.as_tensor(weight, dtype=flat.dtype, device=flat.device).reshape(1)
        flat = torch.cat([flat * w, w])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat = flat[:-1] / flat[-1].clamp_min(torch.finfo(flat.dtype).tiny)
    o = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[o:o + n].view_as(g))
        o += n


def all_ranks_any(flag, device=None, group=None):
    """Logical OR of a per-rank Python bool over all ranks (one scalar all-reduce; the local value when torch.distributed is not
    initialised).  The batched training loops use it so that every rank issues the same number of gradient all-reduces even though
    shards finish their episodes at different generations."""
    import torch.distributed as dist
    if _single_process(group):
        return bool(flag)
    dev = device if (device is not None and dist.get_backend(group) == 'nccl') else torch.device('cpu')
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())
