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


# ------------------------------------------------------------------------------------------------------------ RCCL, one rank
def _nccl1_worker(port, q):
    """world size 1, backend nccl (= RCCL) on cuda:0, collectives forced (MBX_FORCE_COLLECTIVES=1): real result rows through gather_rows, real
    gradients through average_gradients, the loop-control flag through all_ranks_any -- the device-tensor branch of every collective this
    framework issues."""
    try:
        sys.path.insert(0, ROOT)
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        os.environ['MBX_FORCE_COLLECTIVES'] = '1'
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        import torch.distributed as dist
        from metabox_amd.distributed import all_ranks_any, average_gradients, gather_rows
        torch.cuda.set_device(0)
        dev = torch.device('cuda', 0)
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
        n = 48
        pidx, seeds = _table(n)
        rows = _run_rows(pidx, seeds, 12)
        assert rows.is_cuda
        full = gather_rows(rows, n, bounds=np.array([0, n]))                       # all_gather over RCCL (padded shard, cost-weighted bounds form)
        full_eq = gather_rows(rows, n)                                             # ... and the equal-count form
        torch.manual_seed(0)
        net = torch.nn.Linear(35, 7).to(dev)
        net(torch.rand(5, 35, device=dev)).sum().backward()
        before = [p.grad.clone() for p in net.parameters()]
        average_gradients(list(net.parameters()), weight=torch.tensor(5., device=dev))   # weighted flat all-reduce over RCCL
        after_w = [p.grad.clone() for p in net.parameters()]
        average_gradients(list(net.parameters()))                                  # plain mean
        flags = (all_ranks_any(True, dev), all_ranks_any(False, dev))              # scalar MAX all-reduce on a device tensor
        t = torch.arange(8, dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        q.put(('ok', dist.get_backend(), torch.equal(full, rows) and torch.equal(full_eq, rows),
               all(torch.allclose(a, b, rtol=1e-6, atol=0) for a, b in zip(before, after_w)) and
               all(torch.allclose(a, p.grad, rtol=1e-6, atol=0) for a, p in zip(before, net.parameters())),
               flags, t.cpu().tolist()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:
        import traceback
        q.put(('error', repr(exc), traceback.format_exc()))


def test_rccl_world_size_one_runs_every_collective_of_the_framework():
    """no test or bench had ever executed an `nccl` process group (two ranks on one device fall back to gloo above), so the first RCCL
    call of this code would have been the driver's 8-GPU run.  One rank, backend nccl, on cuda:0: result rows, gradients and the loop-control flag go
    through RCCL's all_gather / all_reduce on device tensors."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_nccl1_worker, args=(39500 + os.getpid() % 2000, q))
    p.start()
    try:
        out = q.get(timeout=300)
    finally:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    assert out[0] == 'ok', out
    _, backend, rows_ok, grads_ok, flags, summed = out
    print(f'world-size-1 process group, backend = {backend}')
    assert backend == 'nccl' and rows_ok and grads_ok and flags == (True, False) and summed == list(map(float, range(8)))


def test_bench_under_torchrun_one_rank_nccl():
    """The driver's launch line at N = 1 through torch.distributed.run with the RCCL backend: bench.py builds the process group, the max-over-ranks /
    sum-over-ranks reductions and the per-rank diagnostics run through RCCL on device tensors (over_ranks), and the line says `"backend": "nccl"`."""
    import json
    import subprocess
    port = 41500 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '12', '--warmup', '3',
                          '--dist-backend', 'nccl', '--no-cpu-baseline', '--no-other-configs', '--no-pmc', '--repeats', '3'],
                         capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-400:]
    d = json.loads(lines[0])
    assert d['backend'] == 'nccl' and d['n_gpus'] == 1 and d['ranks_seen'] == [{'rank': 0, 'device': 0}]
    assert d['value'] > 1e6 and len(d['per_rank']['ms_per_step']) == 1
