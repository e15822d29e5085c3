"""Instance sharding across the GPUs of one node and end-of-epoch result aggregation.

Instances are independent (nothing is shared between (problem x run) pairs,
src/tester.py:190-202), so the step path has NO collective.  Each rank owns a contiguous block of global instance
ids; the Philox key of an instance is a function of its global id only, so the gathered result is identical for any
number of ranks.  The only communication is one all-gather of the per-instance result rows
([n_local, n_logpoint+1 + 3] float64: cost curve, fes, return, steps) at the end of an epoch — RCCL over xGMI when
the process group is "nccl", gloo in the CPU tests.
"""
import numpy as np
import torch


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


def gather_rows(rows, n_total, group=None):
    """All-gather the local result rows of every rank into the global [n_total, C] table (rank order == global id
    order because shards are contiguous).  Shards may differ in size by one row: rows are padded to the maximum."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rows
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(max_n, rows.shape[1], dtype=rows.dtype, device=rows.device)
    pad[:rows.shape[0]] = rows
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0)


def average_gradients(params, group=None):
    """Data-parallel training: average the gradients of `params` over all ranks (one flat all-reduce; the RLEPSO actor +
    critic are 6.9 k parameters, the LDE LSTM 32.6 k — latency-bound on xGMI, nothing to bucket)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
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
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(flag)
    dev = device if (device is not None and dist.get_backend(group) == 'nccl') else torch.device('cpu')
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(t.item())
