#!/usr/bin/env python
"""Headline benchmark: RLEPSO rollout throughput on bbob d=10 pop=100, 4096 lock-step instances per MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A *step* is one lock-step generation of the whole instance batch: one batched policy forward (PyTorch-ROCm) and
one fused RLEPSO generation kernel (mbx_step).  An episode is 199 generations (maxFEs = 20 000, pop 100); when it
ends the batch is re-initialised (mbx_reset, a new Philox episode) and stepping continues, so K steps may span
several episodes.  value = env-steps/s = sum over the K timed steps of the number of instances that were not yet
done (reference stop rule gbest <= 1e-8 kept) / wall time, aggregated over all ranks (weak scaling: 4096
instances per GPU).  Inputs (problem constants, policy weights, instance state) are resident in HBM before the
timed region.

Also reported on the same JSON line:
  roofline     — fused generation kernel: algorithmic bytes (54.0 KB per env-step) x live instances
                 per launch / average kernel duration measured with HIP events on the launch stream, vs 8 TB/s.
  cpu_baseline — the C oracle (a float64 port of the reference path, oracle/mbx_oracle.c) on all host cores (one
                 single-threaded worker per core) and on one core, on a bounded sample of the same workload
                 (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NP_, DIM, MAXFES, NLOG = 100, 10, 20000, 50
LOGI = MAXFES // NLOG
EPISODE_GENS = -(-(MAXFES - NP_) // NP_)            # 199
INSTANCES_PER_GPU = 4096
WORD = 8
# per env-step  2*S + 4*35 + (D^2+D+2)*w + 13,  S = (3*NP*D + 3*NP + D + 1)*w + 16
_S = (3 * NP_ * DIM + 3 * NP_ + DIM + 1) * WORD + 16
ALGO_BYTES_PER_STEP = 2 * _S + 4 * 35 + (DIM * DIM + DIM + 2) * WORD + 13
HBM_PEAK_GBS = 8000.0


def make_config():
    from metabox_amd.config import get_config
    return get_config(['--problem', 'bbob', '--difficulty', 'easy', '--dim', str(DIM), '--device', 'cuda'])


def load_agent(config, device):
    from metabox_amd.agent.rlepso_agent import RLEPSO_Agent
    config.agent_save_dir = None
    agent = RLEPSO_Agent(config)
    w = os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz')
    agent.load_exported_weights(np.load(w))
    return agent.to(device)


def cpu_baseline(seconds_budget=20.0):
    """The oracle (C port of the reference path, oracle/mbx_oracle.c) timed on the host over whole episodes of the same workload:
    one worker process per host core (oracle/cpu_workload.py; actions sampled from the actor's (mu, sigma) table), and the
    same worker alone for the single-core figure."""
    import subprocess
    import tempfile
    from metabox_amd.agent.rlepso_agent import ActorTable
    torch.set_num_threads(1)
    config = make_config()
    config.device = 'cpu'
    agent = load_agent(config, 'cpu')
    table = ActorTable(agent.actor, MAXFES, NP_, 'cpu').table.numpy()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                                       # a cgroup CPU quota (cpu.max = "<quota> <period>") bounds the useful workers
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            cores = min(cores, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    cores = max(1, min(cores, 128))
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
    script = os.path.join(ROOT, 'oracle', 'cpu_workload.py')

    def leg(n, seconds):
        procs = [subprocess.Popen([sys.executable, script, '--table', path, '--seconds', str(seconds), '--worker', str(k)],
                                  stdout=subprocess.PIPE, env=env, cwd=ROOT) for k in range(n)]
        outs = []
        for pr in procs:
            out, _ = pr.communicate(timeout=seconds * 6 + 120)
            if pr.returncode != 0:
                raise RuntimeError('cpu_baseline worker failed')
            outs.append(json.loads(out.decode().strip().splitlines()[-1]))
        return outs

    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'actor_table.npy')
        np.save(path, table)
        one = leg(1, 0.4 * seconds_budget)[0]
        many = leg(cores, 0.6 * seconds_budget) if cores > 1 else [one]
    single = one['steps'] / one['seconds']
    # every worker runs for the same wall time; the aggregate rate is the sum of the workers' own rates
    value = sum(w['steps'] / w['seconds'] for w in many)
    steps, episodes = sum(w['steps'] for w in many), sum(w['episodes'] for w in many)
    wall = max(w['seconds'] for w in many)
    return {'value': value, 'unit': 'env-steps/s', 'cores': len(many), 'kind': 'port', 'single_core_value': single,
            'sample': f'{episodes} whole RLEPSO episodes (bbob d=10 pop=100, functions round-robin, same policy as the GPU run: '
                      f'actions drawn from the actor\'s (mu, sigma) table), {steps} env-steps in {wall:.1f} s on {len(many)} '
                      f'single-threaded worker processes (one per host core the container may use), C float64 oracle; one worker alone: '
                      f'{one["steps"]} env-steps in {one["seconds"]:.1f} s'}


def pmc_traffic_per_launch(live_per_launch):
    """HBM bytes per launch of the generation kernel from the committed rocprofv3 PMC passes
    (2 x FETCH_SIZE + WRITE_SIZE, calibration in profiles/*_pmc_hbm_traffic.json; collected with every instance
    live), scaled to this run's average number of live instances per launch.  None when no profile is committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                per_launch = json.load(f)['calibration']['hbm_bytes_per_launch']
        except (OSError, ValueError, KeyError, TypeError):
            continue                                   # an incomplete profile must never take the bench line down
        return per_launch / INSTANCES_PER_GPU * live_per_launch
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2 * EPISODE_GENS)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--instances', type=int, default=INSTANCES_PER_GPU, help='instances per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--functions', default='all24',
                    help="all24 (default: every bbob function round-robin), train18 (the bbob-easy train split), or a comma list of "
                         "function ids (SURVEY §8(d) C2 asks for the split and per-function figures next to the headline)")
    ap.add_argument('--fixed-horizon', action='store_true',
                    help='disable the reference stop rule gbest <= 1e-8: every instance runs all 199 generations')
    ap.add_argument('--policy', choices=['fused', 'hip', 'torch', 'table'], default='fused',
                    help='fused: the generation kernel draws its own action from the actor table (mbx_rlepso_act_step, default); '
                         'hip: mbx_gauss_policy + mbx_step; torch: the two MLPs as batched PyTorch ops; table: (mu, sigma) gathered '
                         'from the per-fes table with PyTorch ops')
    ap.add_argument('--graph-policy', action='store_true', help='with --policy torch / table: replay the policy as one hipGraph')
    ap.add_argument('--event-stride', type=int, default=8, help='bracket every n-th generation kernel with HIP events (1 = all)')
    ap.add_argument('--dist-backend', default='nccl', help='process-group backend (nccl = RCCL; gloo only for single-GPU plumbing tests)')
    ap.add_argument('--same-device', action='store_true', help='plumbing test: every rank uses cuda:0')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.same_device:
            local = 0
        torch.cuda.set_device(local)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group(args.dist_backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import RLEPSO_Optimizer
    from metabox_amd.problem.bbob import BBOB_Dataset
    config = make_config()
    agent = load_agent(config, dev)
    optimizer = RLEPSO_Optimizer(config)
    tr, te = BBOB_Dataset.get_datasets('bbob', DIM, 5.0)
    ps = sorted(tr.data + te.data, key=lambda p: p.func_id)          # all 24 functions, every branch exercised
    if args.functions == 'train18':
        ps = sorted(tr.data, key=lambda p: p.func_id)
    elif args.functions != 'all24':
        want = [int(x) for x in args.functions.split(',')]
        ps = [p for p in ps if p.func_id in want]
        if len(ps) != len(set(want)):
            raise SystemExit(f'--functions: unknown bbob function id in {want}')
    B = args.instances
    gid = np.arange(B, dtype=np.int64) + rank * B                       # global instance ids: weak scaling
    pidx = (gid % len(ps)).astype(np.int32)
    seeds = (gid // len(ps)).astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(12345)
    env = BatchedPBO_Env(ps, optimizer, pidx, seeds, early_stop=not args.fixed_horizon)
    actor = agent.actor
    table = agent.actor_table(MAXFES, NP_, dev)

    h1, h2 = actor.hidden_sizes()
    if args.policy in ('fused', 'hip') and args.graph_policy:
        raise SystemExit('--graph-policy applies to --policy torch / table')
    # fused: the actor evaluated at every reachable state, once (rebuilt whenever the weights change; they do not during a rollout)
    fused_table = env.batch.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma) if args.policy == 'fused' else None

    def policy(st):
        if args.policy == 'hip':            # reads the batch's own state tensor (st is that tensor)
            return env.batch.gauss_policy(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
        return table.act(st) if args.policy == 'table' else actor.act_batch(st.to(torch.float32))

    def steps_sum():
        return int(env.results()['steps'].sum().item())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    K, W = args.steps, args.warmup
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    state = env.reset()
    # The policy forward is two 3-layer MLPs, tanh, Normal sampling and clamp: ~15 tiny kernels.  Launched eagerly they
    # are asynchronous and hide behind the previous generation kernel; --graph-policy captures them once into a hipGraph
    # (input = the batch's persistent state tensor, output = a static action tensor) and replays it every generation.
    # The generation kernel itself is always launched eagerly so that HIP events can bracket it.
    policy_graph, static_actions = None, None
    if args.graph_policy:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                static_actions = policy(state)
        torch.cuda.current_stream().wait_stream(side)
        policy_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(policy_graph), torch.no_grad():
            static_actions = policy(state)
    gen_in_ep, live, base = 0, 0, 0
    t0 = None
    with torch.no_grad():
        for it in range(W + K):
            if it == W:
                barrier()
                base = steps_sum()
                t0 = time.perf_counter()
            if gen_in_ep == EPISODE_GENS:
                if it > W:
                    live += steps_sum() - base
                    base = 0
                state = env.reset()
                gen_in_ep = 0
                if it <= W:
                    base = 0
            if fused_table is not None:
                actions = None
            elif policy_graph is not None:
                policy_graph.replay()
                actions = static_actions
            else:
                actions = policy(state)
            timed = it >= W and (it - W) % args.event_stride == 0
            if timed:
                ev0[it - W].record()
            if fused_table is not None:
                state, _, _ = env.batch.act_step(fused_table)
            else:
                state, _, _ = env.step(actions)
            if timed:
                ev1[it - W].record()
            gen_in_ep += 1
        barrier()
        elapsed = time.perf_counter() - t0
        live += steps_sum() - base
    n_timed = len(range(0, K, args.event_stride))
    kern_ms = sum(ev0[k].elapsed_time(ev1[k]) for k in range(0, K, args.event_stride)) * K / n_timed

    red_dev = dev if args.dist_backend == 'nccl' else torch.device('cpu')
    tot = torch.tensor([float(live), kern_ms], dtype=torch.float64, device=red_dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    live_all, kern_ms_all, elapsed_max = float(tot[0]), float(tot[1]), float(tmax[0])

    if rank == 0:
        fn_desc = {'all24': '24 bbob functions', 'train18': 'the 18 bbob-easy train functions'}.get(
            args.functions, 'bbob function(s) ' + args.functions)
        stop_desc = 'fixed horizon (stop rule disabled)' if args.fixed_horizon else 'reference stop rule'
        value = live_all / elapsed_max
        avg_kernel_s = (kern_ms_all / world) / K / 1e3
        live_per_launch = live_all / world / K
        bytes_per_launch = ALGO_BYTES_PER_STEP * live_per_launch
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        traffic = pmc_traffic_per_launch(live_per_launch)
        out = {
            'metric': 'env-steps/sec (instances x gens/s), RLEPSO bbob-easy d=10', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': elapsed_max / K * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'RLEPSO_Agent + RLEPSO_Optimizer, bbob dim=10 pop=100, {B} lock-step instances per GPU '
                                   f'({fn_desc} round-robin x seeds), maxFEs=20000 (199 generations/episode), '
                                   f'{stop_desc}, policy = exported bbob_easy RLEPSO weights sampled on device',
                       'instances_per_gpu': B, 'live_env_steps': live_all, 'parallelism': f'instances sharded x{world}',
                       'policy': {'fused': 'act + step in one launch (mbx_rlepso_act_step): the generation kernel draws its action from the '
                                           'actor (mu, sigma) table built by mbx_rlepso_policy_table (actor evaluated at every reachable '
                                           'state fes/maxFEs)',
                                  'hip': 'mbx_gauss_policy (both MLPs over the whole batch, one launch) + mbx_step per generation',
                                  'torch': 'both actor MLPs as batched PyTorch ops every generation' + (', hipGraph replay' if args.graph_policy else ''),
                                  'table': '(mu, sigma) gathered from the per-fes table with PyTorch ops' + (', hipGraph replay' if args.graph_policy else '')}[args.policy],
                       'kernel_timing': f'HIP events around every {args.event_stride}-th generation kernel'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'kernel': 'k_rlepso_step',
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'avg_kernel_us': avg_kernel_s * 1e6, 'algorithmic_bytes_per_env_step': ALGO_BYTES_PER_STEP,
                         'live_instances_per_launch': live_all / world / K},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
