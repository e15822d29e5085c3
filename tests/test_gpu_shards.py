"""GPU, world_size 2 on ONE device: two ranks run the real fused RLEPSO generation kernel on their `shard_range` of an instance table,
all-gather their result rows (`gather_rows`) and must reproduce the unsharded batch bit for bit (reference property:
runs are independent, src/tester.py:190-202).  The process group is tried with backend "nccl" (= RCCL) first -- two ranks sharing one
GPU is not a configuration RCCL has to support -- and falls back to gloo; which one ran is printed and returned to the test.  The 8-GPU
scaling run is the driver's; this test pins that the multi-rank path is correct with the real kernel, not only with fake rows."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PROBLEMS, RUNS, GENS = 24, 5, 25          # 120 instances, uneven split impossible: 60 / 60; see n_total below for the ragged case


def _run_rows(pidx, seeds, gens):
    """The real path: Suite + Batch(ALGO_RLEPSO) + fused act/step, returns the packed result rows (device tensor)."""
    from metabox_amd._abi import ALGO_RLEPSO
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.distributed import pack_rows
    from metabox_amd.problem.bbob import BBOB_Dataset
    from metabox_amd.suite import Batch, Suite
    tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0)
    ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
    cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
    cfg.agent_save_dir = None
    agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
    actor = agent.actor
    h1, h2 = actor.hidden_sizes()
    b = Batch(Suite(ps), ALGO_RLEPSO, pidx, seeds, 100, 20000, 400, 50)
    table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
    b.reset()
    for _ in range(gens):
        b.act_step(table)
    rows = pack_rows(b.results()).clone()
    b.close()
    return rows


def _table(n_total):
    from metabox_amd.distributed import instance_table, philox_seed
    pidx, run = instance_table(N_PROBLEMS, RUNS)
    pidx, run = pidx[:n_total], run[:n_total]
    return pidx, philox_seed(run, np.arange(n_total), epoch_salt=7)


def _worker(rank, world, port, n_total, backend, q):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    from metabox_amd.distributed import gather_rows, shard_range
    torch.cuda.set_device(0)
    try:
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        pidx, seeds = _table(n_total)
        lo, hi = shard_range(n_total, rank, world)
        rows = _run_rows(pidx[lo:hi], seeds[lo:hi], GENS)
        if backend != 'nccl':
            rows = rows.cpu()
        full = gather_rows(rows, n_total)
        torch.cuda.synchronize()
        q.put((rank, 'ok', full.cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:                                   # reported to the parent instead of a silent non-zero exit
        q.put((rank, 'error', repr(exc)))


def _spawn(n_total, backend, port):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, backend, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in range(2):
            outs.append(q.get(timeout=240))
    except Exception:
        outs.append((-1, 'error', 'timeout'))
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    return outs


@pytest.mark.parametrize('n_total', [120, 119])
def test_two_ranks_on_one_gpu_gather_the_unsharded_table(n_total):
    sys.path.insert(0, ROOT)
    pidx, seeds = _table(n_total)
    want = _run_rows(pidx, seeds, GENS).cpu().numpy()
    torch.cuda.synchronize()
    used, outs = None, None
    for backend in ('nccl', 'gloo'):
        outs = _spawn(n_total, backend, 35500 + (os.getpid() + n_total + (7 if backend == 'gloo' else 0)) % 2000)
        if all(o[1] == 'ok' for o in outs) and len(outs) == 2:
            used = backend
            break
        print(f'backend {backend} with 2 ranks on one device did not work: {[o[2] for o in outs if o[1] != "ok"]}')
    assert used is not None, outs
    print(f'2 ranks on cuda:0, process group backend = {used}')
    for rank, _, full in outs:
        assert full.shape == want.shape
        assert np.array_equal(full, want, equal_nan=True), (rank, 'gathered table differs from the unsharded batch')
    assert want[:, -1].min() == GENS or want[:, -1].min() > 0          # every instance stepped
