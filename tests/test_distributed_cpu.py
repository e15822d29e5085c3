"""CPU, world_size 2, gloo: sharded epochs gather to exactly the unsharded table."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_results(lo, hi, nlog=5):
    """Deterministic stand-in for Batch.results(): a pure function of the global instance id (like the kernel)."""
    gid = torch.arange(lo, hi, dtype=torch.float64)
    cost = gid[:, None] * 10 + torch.arange(nlog + 1, dtype=torch.float64)[None, :]
    return {'cost': cost, 'fes': gid * 3 + 1, 'return': -gid, 'steps': (gid % 7).to(torch.int32)}


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    from metabox_amd.distributed import gather_rows, pack_rows, shard_range, unpack_rows
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    rows = pack_rows(_fake_results(lo, hi))
    full = gather_rows(rows, n_total)
    out = unpack_rows(full)
    if rank == 0:
        q.put({k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_total', [10, 11])
def test_gather_equals_unsharded(n_total):
    from metabox_amd.distributed import pack_rows, unpack_rows
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_total) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = unpack_rows(pack_rows(_fake_results(0, n_total)))
    for k in want:
        assert np.array_equal(got[k], want[k].numpy()), k


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from metabox_amd.distributed import average_gradients
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Linear(3, 2)
    x = torch.full((4, 3), float(rank + 1))
    net(x).sum().backward()
    average_gradients(list(net.parameters()))
    if rank == 0:
        q.put([p.grad.clone().numpy() for p in net.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_averaging_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gw, gb = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # rank r sees x = r+1: d/dW sum(Wx+b) = 4*(r+1) per entry, averaged over ranks 1 and 2 -> 6; bias grad 4
    assert np.allclose(gw, 6.0) and np.allclose(gb, 4.0)
