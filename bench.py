#!/usr/bin/env python
"""Headline benchmark: RLEPSO rollout throughput on bbob d=10 pop=100, 4096 lock-step instances per MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A *step* is one lock-step generation of the whole instance batch: agent.act (the actor's (mu, sigma) looked up in the per-state table that ONE
mbx_rlepso_policy_table launch builds per weight set, sampled inside the generation kernel) + RLEPSO_Optimizer.update + problem.eval, i.e. one
generation of the resident kernel k_rlepso_run (mbx_rlepso_rollout, --gens-per-launch generations per launch; --policy selects the older routes: a
PyTorch / HIP policy forward + mbx_step per generation).  An episode is 199 generations (maxFEs = 20 000, pop 100); when it
ends the batch is re-initialised (mbx_reset, a new Philox episode) and stepping continues, so K steps may span
several episodes.  value = env-steps/s = sum over the K timed steps of the number of instances that were not yet
done (reference stop rule gbest <= 1e-8 kept) / wall time, aggregated over all ranks (weak scaling: 4096
instances per GPU).  Inputs (problem constants, policy weights, instance state) are resident in HBM before the
timed region.

Also reported on the same JSON line:
  roofline     — fused generation kernel: algorithmic bytes (54.0 KB per env-step) x live instances
                 per launch / average kernel duration measured with HIP events on the launch stream, vs 8 TB/s.
  roofline.valu — the kernel is VALU-issue bound, not bandwidth bound: wave-instructions per generation (one rocprofv3 --pmc child pass) priced at 4 cycles each
                 against the SIMD cycles of the TIMED window, whose shader clock is sampled by a probe wave beside the kernel, no profiler attached
                 (frac_timed_window; useful_f64_frac = that x float64 share x active lanes / 64); the profiled launch's own ratio stays as frac_profiled_launch.
  other_configs — one GPU's share of BASELINE.json configs 3 (LDE, NP = 50 and NP = 100), 4 (DE-DDQN on protein docking) and
                 5 (RLEPSO D = 40, NP = 128): ms per lock-step step, env-steps/s, algorithmic bytes and roofline fraction (N = 1 only).
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


def _child_env():
    """Environment of the rocprofv3 child processes: this process's, WITHOUT the rendezvous variables of torch.distributed.run (a child that inherited RANK /
    WORLD_SIZE / MASTER_* would join -- or collide with -- the parent's process group) and with the child marker set."""
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith('TORCHELASTIC_') or k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'MASTER_ADDR', 'MASTER_PORT'))}
    env.update(MBX_BENCH_CHILD='1', TMPDIR='/tmp')
    return env


def _latest_profile():
    """The newest committed PMC summary (tools/pmc_summary.py output), or None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')), reverse=True):
        try:
            with open(path) as f:
                return os.path.basename(path), json.load(f)
        except (OSError, ValueError):
            continue                                   # an incomplete profile must never take the bench line down
    return None, None


def _profile_for(resident):
    """(file name, profile, env-steps per profiled launch) of the newest committed PMC summary IF it was taken on the kernel this run
    uses (k_rlepso_run for --policy resident, k_rlepso_step otherwise); (None, None, None) otherwise."""
    name, prof = _latest_profile()
    try:
        cal = prof['calibration']
        kernel = cal.get('kernel', 'mbx::k_rlepso_step')
        if kernel != ('mbx::k_rlepso_run' if resident else 'mbx::k_rlepso_step'):
            return None, None, None
        return name, prof, float(cal.get('env_steps_per_launch', INSTANCES_PER_GPU))
    except (KeyError, TypeError, AttributeError):
        return None, None, None


def pmc_traffic_per_launch(live_per_launch, resident=False):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, calibration in
    profiles/*_pmc_hbm_traffic.json; collected with every instance live), scaled to the env-steps one launch of this run processes.
    (None, None) when no profile of this kernel is committed."""
    name, prof, per = _profile_for(resident)
    try:
        return prof['calibration']['hbm_bytes_per_launch'] / per * live_per_launch, name
    except (KeyError, TypeError):
        return None, None


def pmc_traffic_in_run(instances, timeout_s=150):
    """HBM bytes per env-step of k_rlepso_run AND its vector-ALU utilisation at the clock the chip really ran at, measured DURING this bench run the way
    MI355X_MICROARCH.md prescribes: separate rocprofv3 passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, one pass of SQ / GRBM counters; no trace domains) of
    a child `bench.py --steps 20 --warmup 2 --repeats 1` (generations 3-22 of an episode: every instance live, one 20-generation launch).
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of that launch (FETCH_SIZE under-reports these loads x2 on gfx950, WRITE_SIZE is 1:1: calibration in
    profiles/README.md), divided by its instances x 20 env-steps.  clock = GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's own duration (its timestamps in
    the counter file).  Returns (bytes per env-step, note, valu dict or None) or (None, reason, None)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found', None
    tmp = tempfile.mkdtemp(prefix='mbx_pmc_')
    got, valu = {}, None
    sq_pass = 'GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_SALU'
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE', sq_pass):
            out = os.path.join(tmp, counter.split()[0])
            env = _child_env()
            cmd = [exe, '--pmc', *counter.split(), '--output-format', 'csv', '-d', out, '-o', 'p', '--', sys.executable, os.path.abspath(__file__), '--steps', '20',
                   '--warmup', '2', '--repeats', '1', '--instances', str(instances), '--no-cpu-baseline', '--no-other-configs']
            r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
            rows = {}                                       # dispatch id -> {counter: value, 't0', 't1'}
            for path in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if 'k_rlepso_run' in row['Kernel_Name']:
                            d = rows.setdefault(row['Dispatch_Id'], {})
                            d[row['Counter_Name']] = float(row['Counter_Value'])
                            if row.get('Start_Timestamp') and row.get('End_Timestamp'):
                                d['t0'], d['t1'] = float(row['Start_Timestamp']), float(row['End_Timestamp'])
            first = counter.split()[0]
            rows = [d for d in rows.values() if first in d]
            if r.returncode != 0 or not rows:
                if counter is sq_pass:
                    break                                       # the traffic figure stands without the VALU pass
                return None, f'rocprofv3 --pmc {first} pass failed (rc {r.returncode})', None
            # the 20-generation launch (the other dispatch is the 2-generation warm-up): the LONGEST dispatch (GRBM_GUI_ACTIVE of a process's first dispatch can
            # exceed a later, longer one's: round 5 saw the warm-up picked by counter value)
            big = max(rows, key=lambda d: (d.get('t1', 0.) - d.get('t0', 0.), d.get('SQ_INSTS_VALU', 0.), d[first]))
            if counter is not sq_pass:
                got[counter] = big[counter]
                continue
            dur_ns = big.get('t1', 0.) - big.get('t0', 0.)
            cyc = big['GRBM_GUI_ACTIVE'] / 8.                    # summed over the 8 XCDs
            if dur_ns > 0 and cyc > 0 and 'SQ_ACTIVE_INST_VALU' in big:
                f64 = sum(big.get(k, 0.) for k in ('SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_FMA_F64'))
                valu = {'bound': 'valu', 'measured_in_run': True, 'clock_ghz': cyc / dur_ns,
                        'profiled_launch_us': dur_ns / 1e3, 'generations_in_profiled_launch': 20,
                        'wave_instructions_per_generation': big['SQ_INSTS_VALU'] / 20.,
                        'wave_instructions_per_env_step': big['SQ_INSTS_VALU'] / (instances * 20.),
                        'f64_share': f64 / big['SQ_INSTS_VALU'] if big.get('SQ_INSTS_VALU') else None,
                        'salu_per_valu': big.get('SQ_INSTS_SALU', 0.) / big['SQ_INSTS_VALU'] if big.get('SQ_INSTS_VALU') else None,
                        'active_lanes_per_instruction': big['SQ_THREAD_CYCLES_VALU'] / big['SQ_ACTIVE_INST_VALU'] if big.get('SQ_THREAD_CYCLES_VALU') else None,
                        # SQ_ACTIVE_INST_VALU counts quad-cycles in which a SIMD issues a vector instruction; 1024 SIMDs x the launch's cycles is all there is
                        'issue_busy_us_per_generation': big['SQ_ACTIVE_INST_VALU'] * 4. / 1024. / (cyc / dur_ns) / 1e3 / 20.,
                        'frac': big['SQ_ACTIVE_INST_VALU'] * 4. / (1024. * cyc),
                        'source': 'collected during this run: one rocprofv3 --pmc pass (GRBM_GUI_ACTIVE + 7 SQ counters) of the child window; clock = GRBM_GUI_ACTIVE / 8 XCDs / '
                                  'the dispatch\'s own duration; frac = SIMD cycles issuing a VALU instruction / all SIMD cycles of the launch'}
        return (2 * got['FETCH_SIZE'] + got['WRITE_SIZE']) * 1024 / (instances * 20), \
            f"2 x FETCH_SIZE ({got['FETCH_SIZE']:.0f} KB) + WRITE_SIZE ({got['WRITE_SIZE']:.0f} KB) of one 20-generation launch with {instances} live instances", valu
    except Exception as e:                                # a profiler hiccup must never take the bench line down
        return None, f'{type(e).__name__}: {e}', None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def qnet_child(launches=60):
    """config 4's policy GEMM alone: mbx_ddqn_qnet (k_qnet_argmax, Q-network 99 -> 100 x 4 -> 4 + argmax on the float32 matrix cores) over one GPU's share of
    config 4 (2240 instances); the parent runs this under one rocprofv3 --pmc pass."""
    import torch
    from metabox_amd.agent import DE_DDQN_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import DE_DDQN_Optimizer
    from metabox_amd.utils import construct_problem_set
    torch.cuda.set_device(0)
    cfg = get_config(['--problem', 'protein', '--device', 'cuda'])
    cfg.agent_save_dir = None
    torch.manual_seed(0)
    agent = DE_DDQN_Agent(cfg).to('cuda')
    tr, te = construct_problem_set(cfg)
    ps = (tr + te).data[:35]
    B = 35 * 64
    env = BatchedPBO_Env(ps, DE_DDQN_Optimizer(cfg), np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1)
    env.reset()
    packed = agent.packed_weights()
    for _ in range(launches):
        env.step(env.batch.ddqn_qnet(packed))
    torch.cuda.synchronize()
    env.close()


def policy_mfma_in_run(timeout_s=120):
    """Matrix-pipe utilisation of the policy GEMM that IS on a timed route -- config 4's k_qnet_argmax -- measured during this bench run: one rocprofv3 --pmc pass
    (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, GRBM_GUI_ACTIVE; no trace domains) over `bench.py --qnet-child`.  utilisation = busy cycles of the matrix pipes /
    (1024 SIMDs x the dispatch's cycles), both from the same dispatch.  -> dict or None."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix='mbx_mfma_')
    try:
        cmd = [exe, '--pmc', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_MFMA', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VALU', '--output-format', 'csv', '-d', tmp, '-o', 'p', '--',
               sys.executable, os.path.abspath(__file__), '--qnet-child']
        r = subprocess.run(cmd, cwd='/tmp', env=_child_env(), capture_output=True, text=True, timeout=timeout_s)
        acc = {}
        for path in glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if 'k_qnet_argmax' in row['Kernel_Name']:
                        acc.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
        if r.returncode != 0 or 'SQ_VALU_MFMA_BUSY_CYCLES' not in acc or 'GRBM_GUI_ACTIVE' not in acc:
            return None
        m = {k: float(np.mean(v)) for k, v in acc.items()}
        cyc = m['GRBM_GUI_ACTIVE'] / 8.
        return {'kernel': 'k_qnet_argmax<99, 100, 4>', 'dispatches': len(acc['GRBM_GUI_ACTIVE']), 'mfma_instructions_per_launch': m.get('SQ_INSTS_MFMA'),
                'vector_instructions_per_launch': m.get('SQ_INSTS_VALU'), 'shader_cycles_per_launch': cyc, 'utilisation': m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024. * cyc)}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def policy_mfma_profile():
    """Matrix-pipe utilisation of the policy kernels that use the matrix cores (north_star: 'MFMA utilisation on the policy GEMM'), from the newest committed
    profile (tools/exp/mfma_util.sh: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)).  Not collected during the run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_mfma_utilisation.json')), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            d = d.get('kernels', d)
            out = {k.replace('mbx::', ''): round(v['mfma_pipe_utilisation'], 4) for k, v in d.items()
                   if isinstance(v, dict) and 'mfma_pipe_utilisation' in v and ('k_lstm_policy' in k or 'k_qnet_argmax' in k)}
            if out:
                return {'measured_in_run': False, 'source': 'profiles/' + os.path.basename(path), 'peak': 'float32 MFMA 157.3 TFLOP/s (1 instruction per 32 cycles per SIMD)',
                        'utilisation': out,
                        'note': 'the headline policy is the memoised actor table (no matrix work on the step path); k_qnet_argmax is config 4\'s Q-network, k_lstm_policy LDE\'s '
                                'PolicyNet on the one-launch-per-generation route (mbx_lde_rollout evaluates it inside k_lde_run as float32 fma chains)'}
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None


def valu_roofline(live_per_gen, avg_gen_s, resident=False):
    """VALU side of the roofline ('report both achieved GB/s and VALU utilisation'), per GENERATION of the batch:
    wave-instructions by class from the committed PMC profile x the issue cost of each class measured with tools/ubench/valu_rates.hip,
    scaled to this run's live instances.  frac = issue-bound time / measured time (an interval: the counters do not split 2- and 4-cycle
    integer ops)."""
    name, prof, per = _profile_for(resident)
    try:
        v = prof['valu_issue_bound']
        scale = live_per_gen / per
        lo, hi = (x * scale for x in v['issue_bound_us_at_2.4GHz'])
        return {'bound': 'valu', 'wave_instructions_per_generation': v['wave_instructions_per_launch']['total'] * scale,
                'f64_share': v['wave_instructions_per_launch']['f64_add_mul_fma'] / v['wave_instructions_per_launch']['total'],
                'issue_bound_us': [lo, hi], 'measured_us': avg_gen_s * 1e6,
                'frac': [lo / (avg_gen_s * 1e6), hi / (avg_gen_s * 1e6)],
                'measured_in_run': False, 'active_lanes_per_instruction': v.get('active_lanes_per_valu_instruction'),
                'source': f'profiles/{name} (rocprofv3 --pmc passes of this command, not collected during this run) + profiles/r02_valu_issue_rates.txt'}
    except (KeyError, TypeError, ZeroDivisionError):
        return None


def _bracket(fn, steps, stream_sync=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def plugin_view_cost(episodes=3):
    """The single-instance compatibility view -- PBO_Env.reset()/step() through RLEPSO_Optimizer.init_population / update, the reference's own
    protocol, one instance, one host round trip per generation -- timed next to the reference's 2.5 ms per generation.
    This is what a user plugin written against MetaBox pays when it does not use the batched path."""
    from metabox_amd.environment import PBO_Env
    from metabox_amd.optimizer import RLEPSO_Optimizer
    from metabox_amd.problem.bbob import BBOB_Dataset
    cfg = make_config()
    tr, _ = BBOB_Dataset.get_datasets('bbob', DIM, 5.0)
    opt = RLEPSO_Optimizer(cfg)
    rs = np.random.RandomState(0)
    t_reset, t_step, n_steps = [], 0., 0
    for ep in range(episodes + 1):
        env = PBO_Env(tr.data[ep % len(tr.data)], opt)
        t0 = time.perf_counter()
        env.reset()
        t1 = time.perf_counter()
        done, n = False, 0
        while not done:
            _, _, done = env.step(rs.uniform(0, 1, 35).astype(np.float32))
            n += 1
        if ep:                                               # the first episode pays allocation
            t_reset.append(t1 - t0); t_step += time.perf_counter() - t1; n_steps += n
    return {'path': 'PBO_Env + RLEPSO_Optimizer, B = 1: H2D of the action, one generation kernel, one 536-byte D2H (mbx_read_public) per env.step',
            'ms_per_env_step': t_step / n_steps * 1e3, 'env_steps_per_s': n_steps / t_step, 'ms_per_reset': float(np.mean(t_reset)) * 1e3,
            'reference_ms_per_env_step': 2.5, 'reference_source': 'SURVEY.md section 6: 432 env-steps/s on one host core (build container)'}


def other_configs(budget_s=60.0):
    """One GPU's share of BASELINE.json configs 3, 4 and 5, policy included, each on its own lock-step batch: ms per step, env-steps/s,
    algorithmic bytes per env-step and the fraction of the 8 TB/s roofline they imply.  Bounded: a few dozen steps each."""
    from metabox_amd._abi import ALGO_RLEPSO
    from metabox_amd.agent import DE_DDQN_Agent, LDE_Agent, RLEPSO_Agent
    from metabox_amd.config import get_config
    from metabox_amd.environment import BatchedPBO_Env
    from metabox_amd.optimizer import DE_DDQN_Optimizer, LDE_Optimizer
    from metabox_amd.problem.bbob import BBOB_Dataset
    from metabox_amd.suite import Batch, Suite
    from metabox_amd.utils import construct_problem_set
    out, t_start = [], time.perf_counter()

    def entry(name, B, dt, bytes_per_step, extra=None):
        e = {'config': name, 'instances': B, 'ms_per_step': dt * 1e3, 'env_steps_per_s': B / dt,
             'algorithmic_bytes_per_env_step': bytes_per_step, 'roofline_frac': B * bytes_per_step / dt / 1e9 / HBM_PEAK_GBS}
        e.update(extra or {})
        out.append(e)

    with torch.no_grad():
        # ---- config 3: LDE on bbob-noisy d=30, 16384 instances; NP = 50 (the reference's population, lde_optimizer.py:10, shipped weights)
        #      and NP = 100 (BASELINE.json as written; no shipped policy fits 2 NP = 200 outputs: seeded fresh PolicyNet)
        for np_lde in (50, 100):
            if time.perf_counter() - t_start > budget_s:
                break
            cfg = get_config(['--problem', 'bbob-noisy', '--dim', '30', '--device', 'cuda'])
            cfg.agent_save_dir = None
            if np_lde != 50:
                cfg.NP_override = np_lde
            torch.manual_seed(0)
            agent = LDE_Agent(cfg)
            if np_lde == 50:
                agent.load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'lde_bbob_easy.npz')))
            agent.to('cuda')
            opt = LDE_Optimizer(cfg)
            tr, te = construct_problem_set(cfg)
            ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
            B = 16384
            env = BatchedPBO_Env(ps, opt, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
            env.reset()
            net, hh, cc = agent.net, torch.zeros(B, 50, device='cuda'), torch.zeros(B, 50, device='cuda')
            resident3 = env.batch.lde_rollout_is_resident()

            def run(n, env=env, net=net, hh=hh, cc=cc):
                env.batch.lde_rollout(net.packed_weights(), net.lstm.hidden_size, hh, cc, n)     # n generations in ONE launch (k_lde_run), PolicyNet inside the kernel
            run(3)
            dt = sorted(_bracket(run, 50) for _ in range(3))[1]      # launches of 50 generations (the default of rollout_batch; an episode is 1199 / 599 generations); median of 3
            D = 30
            entry(f'config 3: LDE bbob-noisy d=30 pop={np_lde}, 16384 instances (one GPU), LSTM policy included, mbx_lde_rollout (50 generations per launch, median of 3 launches)', B, dt,
                  (2 * np_lde * D + 2 * np_lde) * 8 + 4 * 2 * np_lde + 8 * (np_lde + 10) + (D * D + D + 2) * 8,
                  {'launch_info': {'kernel': f'k_lde_run<{np_lde}, 30>', 'threads': 64 * ((np_lde + 15) // 16), 'resident': bool(resident3),
                                   'step_kernel': env.batch.launch_info()}, 'policy': agent.policy_route('resident' if resident3 else 'hip')})
            env.close()
        # ---- config 4: DE-DDQN on protein docking, one GPU's share = 35 problems x 64 runs (seeded fresh Q-net: no checkpoint ships)
        if time.perf_counter() - t_start <= budget_s:
            cfg = get_config(['--problem', 'protein', '--device', 'cuda'])
            cfg.agent_save_dir = None
            torch.manual_seed(0)
            agent = DE_DDQN_Agent(cfg).to('cuda')
            opt = DE_DDQN_Optimizer(cfg)
            tr, te = construct_problem_set(cfg)
            ps = (tr + te).data[:35]
            B = 35 * 64
            env = BatchedPBO_Env(ps, opt, np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
            st = {'s': env.reset()}

            packed = agent.packed_weights()                       # mbx_ddqn_qnet: Q-network + argmax as one launch on the float32 matrix cores

            def run4(n, env=env, packed=packed):
                for _ in range(n):
                    env.step(env.batch.ddqn_qnet(packed))
            run4(5)
            dt = _bracket(run4, 100)
            entry('config 4: DE-DDQN protein-docking d=12 pop=100, 2240 instances = 35 problems x 64 runs (one GPU of eight), Q-net included', B, dt,
                  15 * 1024, {'note': 'compute-bound (10^4 atom pairs per evaluation), the HBM fraction is nominal', 'launch_info': env.batch.launch_info(),
                              # float64 VALU roofline of the protein energy (one evaluation per env-step): per atom pair ~35 flop (dot product 5, distance 4 + sqrt,
                              # two divisions, Lennard-Jones / Coulomb / switching terms ~24; eval_rows_protein in mbx_device.hpp) x 10^4 pairs + 300 x 12 x 3 for the
                              # displaced coordinates; peak = 1024 SIMDs x 16 lanes/clk x 2 flop x 2.4 GHz (f64 FMA issues in 4 cycles, profiles/r02_valu_issue_rates.txt)
                              'compute_roofline': {'bound': 'valu_f64', 'flops_per_env_step': 3.6e5, 'achieved': B * 3.6e5 / dt / 1e12, 'peak': 78.6, 'unit': 'TFLOP/s',
                                                   'frac': B * 3.6e5 / dt / 1e12 / 78.6,
                                                   # what the kernel executes: the 4950 pairs i < j (symmetric tables) x 35 flop + 10 800 for the coordinates; of those pairs the waves whose
                                                   # pairs all lie beyond the 9 A cut-off skip the arithmetic (~60 % of the list: eval_rows_protein), so this is an upper bound on executed work
                                                   'executed_flops_per_env_step': 4950 * 35 + 10800,
                                                   'achieved_executed': B * (4950 * 35 + 10800) / dt / 1e12, 'frac_executed': B * (4950 * 35 + 10800) / dt / 1e12 / 78.6,
                                                   'note': 'whole step, Q-network launch (mbx_ddqn_qnet) included; `frac` prices the NOMINAL work of all 10^4 pairs (throughput of useful work), `frac_executed` the pairs the kernel visits (i < j; an upper bound: waves beyond the cut-off skip)'}})
            env.close()
        # ---- config 5: RLEPSO on the mixed suite (24 bbob + 30 noisy) d=40 pop=128, 8192 instances per GPU, act + step fused
        if time.perf_counter() - t_start <= budget_s:
            ps = []
            for suite in ('bbob', 'bbob-noisy'):
                tr, te = BBOB_Dataset.get_datasets(suite, 40, 5.0)
                ps += sorted(tr.data + te.data, key=lambda p: p.func_id)
            s5 = Suite(ps)
            cfg = get_config(['--problem', 'bbob', '--dim', '40', '--device', 'cuda'])
            cfg.agent_save_dir = None
            agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
            actor = agent.actor
            h1, h2 = actor.hidden_sizes()
            B, NP5, D5 = 8192, 128, 40
            b = Batch(s5, ALGO_RLEPSO, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 3, NP5, 80000, 1600, 50, early_stop=False)
            table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
            b.reset()

            def run5(n, b=b, table=table):
                b.rlepso_rollout(table, n)                 # n generations in ONE resident launch (k_rlepso_run<1024, 128, 40, 5>)
            run5(4)
            dt = sorted(_bracket(run5, 20) for _ in range(3))[1]      # launches of 20 generations like the headline window (an episode is 624); median of 3
            S5 = (3 * NP5 * D5 + 3 * NP5 + D5 + 1) * 8 + 16
            entry('config 5: RLEPSO mixed suite (24 bbob + 30 noisy) d=40 pop=128, 8192 instances (one GPU of eight), mbx_rlepso_rollout (20 generations per launch, median of 3 launches)', B, dt,
                  2 * S5 + 4 * 35 + (D5 * D5 + D5 + 2) * 8 + 13, {'launch_info': b.launch_info()})
            b.close()
        # ---- the reference's OWN default settings, resident (built for route coverage in round 5, timed since round 6)
        # LDE as shipped: NP = 50 (lde_optimizer.py:10) on bbob --dim 10 (config.py:74), k_lde_run<50, 10>, all 24 objective kinds
        if time.perf_counter() - t_start <= budget_s + 15:
            cfg = get_config(['--problem', 'bbob', '--dim', '10', '--device', 'cuda'])
            cfg.agent_save_dir = None
            agent = LDE_Agent(cfg)
            agent.load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'lde_bbob_easy.npz')))
            agent.to('cuda')
            tr, te = construct_problem_set(cfg)
            ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
            B = 16384
            env = BatchedPBO_Env(ps, LDE_Optimizer(cfg), np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1, early_stop=False)
            env.reset()
            net, hh, cc = agent.net, torch.zeros(B, 50, device='cuda'), torch.zeros(B, 50, device='cuda')
            res10 = env.batch.lde_rollout_is_resident()

            def run10(n, env=env, net=net, hh=hh, cc=cc):
                env.batch.lde_rollout(net.packed_weights(), net.lstm.hidden_size, hh, cc, n)
            run10(3)
            dt = sorted(_bracket(run10, 50) for _ in range(3))[1]
            entry('reference default: LDE bbob d=10 pop=50 (lde_optimizer.py:10), 16384 instances, LSTM policy included, mbx_lde_rollout (50 generations per launch, median of 3 launches)', B, dt,
                  (2 * 50 * 10 + 2 * 50) * 8 + 4 * 2 * 50 + 8 * (50 + 10) + (10 * 10 + 10 + 2) * 8,
                  {'launch_info': {'kernel': 'k_lde_run<50, 10>', 'threads': 64 * ((50 + 15) // 16), 'resident': bool(res10), 'step_kernel': env.batch.launch_info()}})
            env.close()
        # RLEPSO on protein docking (config.py:86-90: dim 12, maxFEs 1000 -> 9 generations per episode), k_rlepso_run<256, 100, 12, 5>: reset + ONE launch per episode
        if time.perf_counter() - t_start <= budget_s + 25:
            cfg = get_config(['--problem', 'protein', '--device', 'cuda'])
            cfg.agent_save_dir = None
            agent = RLEPSO_Agent(cfg).load_exported_weights(np.load(os.path.join(ROOT, 'metabox_amd', 'agent_model', 'rlepso_bbob_easy.npz'))).to('cuda')
            actor = agent.actor
            h1, h2 = actor.hidden_sizes()
            tr, te = construct_problem_set(cfg)
            ps = (tr + te).data[:35]
            B = 35 * 64
            b = Batch(Suite(ps), ALGO_RLEPSO, np.repeat(np.arange(35), 64), np.arange(B, dtype=np.uint64) + 1, 100, 1000, 200, 5)
            table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
            resp = b.rollout_is_resident()

            def runp(n, b=b, table=table):
                for _ in range(n // 9):
                    b.reset()
                    b.rlepso_rollout(table, 9)
            runp(18)
            dt = sorted(_bracket(runp, 90) for _ in range(3))[1]      # ten whole episodes, the initial evaluation (mbx_reset) included; per GENERATION
            Sp = (3 * 100 * 12 + 3 * 100 + 12 + 1) * 8 + 16
            entry('reference default: RLEPSO protein-docking d=12 pop=100 (config.py:86-90), 2240 instances = 35 problems x 64 runs, mbx_reset + mbx_rlepso_rollout (whole 9-generation '
                  'episodes, one launch each; median of 3 x 10 episodes)', B, dt, 2 * Sp + 4 * 35 + 13,
                  {'note': 'compute-bound: 100 protein energies per instance-generation (4950 atom pairs each) against 60 k FDR candidate pairs; the HBM fraction is nominal',
                   'launch_info': {'kernel': 'k_rlepso_run<256, 100, 12, 5>', 'resident': bool(resp), **b.launch_info()}})
            b.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2 * EPISODE_GENS)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--instances', type=int, default=INSTANCES_PER_GPU, help='instances per GPU')
    ap.add_argument('--repeats', type=int, default=0,
                    help='how often the (warm-up + K timed generations) window is measured; 0 (default): once when it takes >= 0.5 s, else 30-200 times '
                         '(~2 s of timed windows), the line reports the median repeat')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the configs 3 / 4 / 5 legs (other_configs)')
    ap.add_argument('--no-fdr-fast', action='store_true', help='skip the side leg that times the MBX_F_FDR_FAST kernels on the same window (roofline.fdr_fast_ms_per_step); kernel traces of the headline alone')
    ap.add_argument('--no-pmc', action='store_true', help='do not collect roofline.traffic in this run (two rocprofv3 --pmc child passes, ~40 s); use the newest committed profile')
    ap.add_argument('--functions', default='all24',
                    help="all24 (default: every bbob function round-robin), train18 (the bbob-easy train split), or a comma list of "
                         "function ids (SURVEY §8(d) C2 asks for the split and per-function figures next to the headline)")
    ap.add_argument('--fixed-horizon', action='store_true',
                    help='disable the reference stop rule gbest <= 1e-8: every instance runs all 199 generations')
    ap.add_argument('--policy', choices=['resident', 'fused', 'hip', 'torch', 'table'], default='resident',
                    help='resident (default): mbx_rlepso_rollout, up to --gens-per-launch generations of act + step per launch with the state '
                         'on chip in between; fused: one launch per generation, the kernel draws its own action from the actor table '
                         '(mbx_rlepso_act_step); '
                         'hip: mbx_gauss_policy + mbx_step; torch: the two MLPs as batched PyTorch ops; table: (mu, sigma) gathered '
                         'from the per-fes table with PyTorch ops')
    ap.add_argument('--gens-per-launch', type=int, default=EPISODE_GENS,
                    help='--policy resident: generations per mbx_rlepso_rollout launch (a launch also ends at the episode / window end).  Default: the whole episode, what '
                         'RLEPSO_Agent.rollout_batch launches (one tail per episode instead of one per 50 generations: whole episodes 3.45e7 -> 3.53e7 env-steps/s)')
    ap.add_argument('--graph-policy', action='store_true', help='with --policy torch / table: replay the policy as one hipGraph')
    ap.add_argument('--event-stride', type=int, default=0,
                    help='bracket every n-th generation kernel with HIP events (default: every kernel when steps <= 64, else every 8th)')
    ap.add_argument('--dist-backend', default='nccl', help='process-group backend (nccl = RCCL; gloo only for single-GPU plumbing tests)')
    ap.add_argument('--same-device', action='store_true', help='plumbing test: every rank uses cuda:0')
    ap.add_argument('--qnet-child', action='store_true', help='(internal) config 4\'s Q-network launch alone, for the MFMA counter pass of roofline.policy_mfma')
    args = ap.parse_args()
    if args.qnet_child:
        return qnet_child()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    # a process group whenever the line is launched through torch.distributed.run -- also at --nproc-per-node 1, where the same collectives then run
    # through a one-rank RCCL communicator (tests/test_gpu_shards.py: the first RCCL call this code makes is not the 8-GPU run's)
    if world > 1 or ('TORCHELASTIC_RUN_ID' in os.environ and 'MASTER_PORT' in os.environ):
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
    from metabox_amd.distributed import philox_seed
    seeds = philox_seed(gid // len(ps), gid, epoch_salt=12345)          # key = f(run, global id): results independent of the GPU count
    env = BatchedPBO_Env(ps, optimizer, pidx, seeds, early_stop=not args.fixed_horizon)
    actor = agent.actor
    table = agent.actor_table(MAXFES, NP_, dev)

    h1, h2 = actor.hidden_sizes()
    if args.policy in ('resident', 'fused', 'hip') and args.graph_policy:
        raise SystemExit('--graph-policy applies to --policy torch / table')
    # fused: the actor evaluated at every reachable state, once (rebuilt whenever the weights change; they do not during a rollout)
    fused_table, table_build_us = None, None
    if args.policy in ('fused', 'resident'):
        fused_table = env.batch.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)     # the one launch that is redone when the weights change, outside every timed window
        e0.record()
        fused_table = env.batch.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
        e1.record()
        torch.cuda.synchronize()
        table_build_us = e0.elapsed_time(e1) * 1e3
    resident = args.policy == 'resident'

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

    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} but WORLD_SIZE is {world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
    K, W = args.steps, args.warmup
    # Kernel timing: ONE event per `stride` generations on the launch stream; consecutive events bracket `stride` back-to-back generation
    # kernels (act + step is a single launch, nothing else runs on the stream), so every kernel of the timed region is covered and an event
    # costs stream time only once per span.  Spans that contain an episode restart (mbx_reset) are left out.
    stride = args.event_stride if args.event_stride > 0 else (2 if K <= 64 else 8)
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
    slot_pool = []

    def clock_slots(detach=False):
        """A fresh zeroed slot pair {sum of shader cycles, sum of 100 MHz ticks over the workgroups' lifetimes}, attached to the batch for its next resident launch."""
        import ctypes as C
        if detach:
            env.batch.lib.mbx_debug_clock_slots(env.batch._h, None)
            return None
        if not slot_pool or slot_pool[-1][1] == slot_pool[-1][0].shape[0]:
            slot_pool.append([torch.zeros(256, 2, dtype=torch.int64, device=dev), 0])
        blk, used = slot_pool[-1]
        slot_pool[-1][1] = used + 1
        env.batch.lib.mbx_debug_clock_slots(env.batch._h, C.c_void_p(blk[used].data_ptr()))
        return blk[used]

    def timed_window():
        """One measurement: mbx_reset, W untimed warm-up generations, barrier + synchronize, EXACTLY K timed generations, barrier +
        synchronize.  Returns (elapsed seconds on this rank, live env-steps of the K generations, resident launches, marks, mark_step, reset_steps)."""
        nonlocal state
        marks, mark_step, reset_steps, launches = [], [], [], []     # launches: (event before, event after, generations, clock slots) of every timed launch
        gen_in_ep, live, base = 0, 0, 0
        t0 = None
        state = env.reset()
        with torch.no_grad():
            it = 0
            while it < W + K:
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
                    if it >= W:
                        reset_steps.append(it - W)
                    if it <= W:
                        base = 0
                if resident:
                    # one launch = up to --gens-per-launch generations; it ends where the warm-up, the timed window or the episode ends
                    n = min(max(1, args.gens_per_launch), EPISODE_GENS - gen_in_ep, (W if it < W else W + K) - it)
                    if it >= W:
                        slots = clock_slots()                          # the launch stamps its own first start / last end per XCD (mbx_debug_clock_slots): host-side pointer set, no device work
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record()
                    state, _, _ = env.batch.rlepso_rollout(fused_table, n)
                    if it >= W:
                        e1 = torch.cuda.Event(enable_timing=True)
                        e1.record()
                        launches.append((e0, e1, n, slots))
                        clock_slots(detach=True)
                    it += n
                    gen_in_ep += n
                    continue
                if fused_table is not None:
                    actions = None
                elif policy_graph is not None:
                    policy_graph.replay()
                    actions = static_actions
                else:
                    actions = policy(state)
                if it >= W and (it - W) % stride == 0:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    marks.append(e); mark_step.append(it - W)
                if fused_table is not None:
                    state, _, _ = env.batch.act_step(fused_table)
                else:
                    state, _, _ = env.step(actions)
                gen_in_ep += 1
                it += 1
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e); mark_step.append(K)
            barrier()
            elapsed = time.perf_counter() - t0
            live += steps_sum() - base
        return elapsed, live, launches, marks, mark_step, reset_steps

    red_dev = dev if args.dist_backend == 'nccl' else torch.device('cpu')

    def over_ranks(elapsed, live):
        """(max over ranks of the window's wall time, sum over ranks of its live env-steps)"""
        if dist is None:
            return elapsed, float(live)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        tsum = torch.tensor([float(live)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        return float(tmax[0]), float(tsum[0])

    # A short window (the driver's --steps 20 is ONE 3 ms launch) is a single sample with the pool's +-3 % box spread and launch jitter on top, so it
    # is repeated: every repeat is a complete measurement (mbx_reset -> a new Philox episode of the same instances, W warm-up generations, barrier, the
    # same K generations timed between barrier + synchronize pairs); nothing between two repeats is inside a timed bracket.  The line reports the
    # MEDIAN repeat.  The number of repeats is derived from the first window's max-over-ranks time, so every rank runs the same number.
    windows = [timed_window()]
    first_elapsed, _ = over_ranks(windows[0][0], windows[0][1])
    repeats = 1
    if args.repeats > 0:
        repeats = args.repeats
    elif first_elapsed < 0.5:
        repeats = int(min(200, max(30, np.ceil(2.0 / max(first_elapsed, 1e-4)))))
    for _ in range(repeats - 1):
        windows.append(timed_window())
    per_repeat = [over_ranks(w[0], w[1]) for w in windows]            # (wall time max over ranks, live env-steps over all ranks) of every repeat
    order = sorted(range(repeats), key=lambda r: per_repeat[r][0])
    med = order[(repeats - 1) // 2]
    elapsed_max, live_all = per_repeat[med]
    times = np.array([t for t, _ in per_repeat])
    launches = [l for w in windows for l in w[2]]
    # cost of an empty event pair on this stream (an event costs stream time too): measured, then subtracted from every bracket
    ea = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
    eb = [torch.cuda.Event(enable_timing=True) for _ in range(64)]
    for a_, b_ in zip(ea, eb):
        a_.record(); b_.record()
    torch.cuda.synchronize()
    pair_ms = sorted(a_.elapsed_time(b_) for a_, b_ in zip(ea, eb))[len(ea) // 2]
    span_ms, span_kernels = 0., 0
    for _, _, _, marks, mark_step, reset_steps in windows:
        for j in range(len(marks) - 1):
            a_, b_ = mark_step[j], mark_step[j + 1]
            if any(a_ < r <= b_ for r in reset_steps) or args.policy != 'fused':      # (the reset of step r is enqueued before the mark of step r)
                continue                                    # a reset kernel (or policy kernels) inside the span: not a pure generation-kernel span
            span_ms += max(marks[j].elapsed_time(marks[j + 1]) - pair_ms, 0.)
            span_kernels += b_ - a_
    if resident:                                        # every launch of the window is bracketed by its own event pair
        span_ms = sum(max(l[0].elapsed_time(l[1]) - pair_ms, 0.) for l in launches)
        span_kernels = sum(l[2] for l in launches)
    if span_kernels == 0:                               # policies that launch their own kernels between generations: fall back to the step time
        span_ms, span_kernels = float(np.median([w[0] for w in windows])) * 1e3, K
    kern_ms = span_ms / span_kernels * K                # generation-kernel time of K timed generations, averaged over every bracketed kernel of every repeat
    n_launch_all = len(launches)

    # per-rank diagnostics for the multi-GPU line: who ran where, how long each rank's median window took, how many env-steps it carried
    diag = torch.tensor([float(rank), float(torch.cuda.current_device()), windows[med][0] / K * 1e3, float(windows[med][1]), kern_ms / K * 1e3], dtype=torch.float64, device=red_dev)
    if dist is not None:
        gathered = [torch.zeros_like(diag) for _ in range(world)]
        dist.all_gather(gathered, diag)
        diag_rows = [g.cpu().tolist() for g in gathered]
    else:
        diag_rows = [diag.cpu().tolist()]
    tot = torch.tensor([kern_ms, float(sum(w[1] for w in windows))], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    kern_ms_all, live_sum_all = float(tot[0]), float(tot[1])

    if rank == 0:
        fn_desc = {'all24': '24 bbob functions', 'train18': 'the 18 bbob-easy train functions'}.get(
            args.functions, 'bbob function(s) ' + args.functions)
        stop_desc = 'fixed horizon (stop rule disabled)' if args.fixed_horizon else 'reference stop rule'
        value = live_all / elapsed_max
        # one "launch" of the dominant kernel: a generation (k_rlepso_step) or, resident, the generations of one mbx_rlepso_rollout call
        n_launch = n_launch_all / repeats if resident else K
        gens_per_launch = K / n_launch
        avg_gen_s = (kern_ms_all / world) / K / 1e3
        avg_kernel_s = avg_gen_s * gens_per_launch
        live_per_gen = live_sum_all / repeats / world / K          # the kernel time is the mean over all repeats, so are the env-steps it is priced with
        live_per_launch = live_per_gen * gens_per_launch           # env-steps (live instance-generations) one launch processes
        bytes_per_launch = ALGO_BYTES_PER_STEP * live_per_launch
        achieved = bytes_per_launch / avg_kernel_s / 1e9
        # shader cycles of the timed launches, stamped by the launches themselves (mbx_debug_clock_slots): no probe wave, no profiler, the timed repeats' own regime
        wc = None
        if resident:
            st = np.array([l[3].cpu().numpy().astype(np.float64) for l in launches if l[3] is not None]).reshape(-1, 2)      # [launch, {cycles, 10 ns ticks}] summed over workgroups
            ev_ns = np.array([max(l[0].elapsed_time(l[1]) - pair_ms, 0.) * 1e6 for l in launches if l[3] is not None])
            gens = np.array([l[2] for l in launches if l[3] is not None], dtype=np.float64)
            ok = (st[:, 1] > 0) if len(st) else np.zeros(0, bool)
            if ok.any():
                ghz = st[ok, 0] / (st[ok, 1] * 10.)
                clock = float(st[ok, 0].sum() / (st[ok, 1].sum() * 10.))
                wc = {'clock_ghz': clock, 'shader_cycles_per_generation': clock * float(ev_ns[ok].sum() / gens[ok].sum()), 'stamped_launches': int(ok.sum()),
                      'clock_ghz_p10_p50_p90_over_launches': [float(np.percentile(ghz, q)) for q in (10, 50, 90)],
                      'method': ('thread 0 of every workgroup of k_rlepso_run reads s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end and adds both '
                                 'differences to the launch\'s sums (mbx_debug_clock_slots): clock = sum of cycles / sum of time over the workgroups of all timed launches; '
                                 'shader cycles per generation = that clock x the launches\' generation time by HIP events.  No probe wave, no profiler, the timed repeats themselves.')}
        # side field: the same window on a batch created with MBX_F_FDR_FAST (the cross-multiplied FDR scan without the near-tie flag / settle stage; the headline runs the
        # exact default).  Same instances, same launches, wall time between synchronize pairs like the headline's repeats; right behind them (same clock regime).
        fdr_fast = None
        if resident and world == 1 and not args.no_fdr_fast and not os.environ.get('MBX_BENCH_CHILD'):
            try:
                from metabox_amd._abi import ALGO_RLEPSO, F_FDR_FAST
                from metabox_amd.suite import Batch
                fb = Batch(env.suite, ALGO_RLEPSO, pidx, seeds, NP_, MAXFES, MAXFES // 50, 50, early_stop=not args.fixed_horizon, flags=F_FDR_FAST)
                assert fb.flags & F_FDR_FAST and fb.rollout_is_resident()
                ts = []
                for _ in range(min(repeats, 60)):
                    fb.reset()
                    it = 0
                    while it < W + K:
                        if it == W:
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                        n = min(max(1, args.gens_per_launch), EPISODE_GENS - it % EPISODE_GENS, (W if it < W else W + K) - it)
                        if it % EPISODE_GENS == 0 and it > 0:
                            fb.reset()
                        fb.rlepso_rollout(fused_table, n)
                        it += n
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t0)
                fb.close()
                fdr_fast = float(np.median(ts) / K * 1e3)
            except Exception as exc:
                fdr_fast = {'error': repr(exc)}
        # the side legs are timed legs: they run right behind the headline's repeats, before the (slow, idle-heavy) profiler children
        side_legs = None
        if world == 1 and not args.no_other_configs:
            try:
                # (the first call of a process also pays the one-instance suite's set-up and a cold clock: 0.13 against 0.058 ms per env.step standalone; the second counts)
                plugin_view_cost(episodes=1)
                side_legs = {'plugin_view': plugin_view_cost(), 'other_configs': other_configs()}
            except Exception as exc:                       # the headline line must survive a failure of the side legs
                side_legs = {'other_configs': {'error': repr(exc)}}
        traffic, traffic_src = pmc_traffic_per_launch(live_per_launch, resident)
        traffic_in_run, traffic_note, valu_in_run = False, None, None
        if resident and world == 1 and not args.no_pmc and not os.environ.get('MBX_BENCH_CHILD'):
            per_step, traffic_note, valu_in_run = pmc_traffic_in_run(B)
            if per_step is not None:
                traffic, traffic_in_run = per_step * live_per_launch, True
        first_gen = W % EPISODE_GENS + 1
        policy_mfma = policy_mfma_profile()
        if world == 1 and not args.no_pmc and not args.no_other_configs and not os.environ.get('MBX_BENCH_CHILD'):
            live = policy_mfma_in_run()
            if live is not None:
                policy_mfma = {'measured_in_run': True, 'peak': 'float32 MFMA 157.3 TFLOP/s (one v_mfma_f32_16x16x4_f32 per 32 cycles per SIMD)', **live,
                               'note': 'config 4\'s Q-network + argmax (mbx_ddqn_qnet), the one policy GEMM on a timed route: 2240 instances x 40 k MAC is ~1 us of matrix '
                                       'work at peak; the launch is five dependent layers of 25-step MFMA chains, i.e. latency.  The headline policy is the memoised actor table '
                                       '(no matrix work on the step path); LDE\'s PolicyNet runs inside k_lde_run as float32 fma chains.',
                               'committed_profiles': (policy_mfma or {}).get('utilisation')}
        if valu_in_run is not None:
            # the PMC child is a process's FIRST short window: the chip's low-clock regime (window_clock), a longer launch: its ratio is that launch's.  The
            # wave-instruction COUNT is the same work in the timed windows; price it against the SIMD cycles of a window in the repeats' regime at the clock sampled there.
            valu_in_run['frac_profiled_launch'] = valu_in_run.pop('frac')
            valu_in_run['clock_ghz_profiled_launch'] = valu_in_run.pop('clock_ghz')
            valu_in_run['timed_window_clock'] = wc
            if wc and 1.0 < wc['clock_ghz'] < 3.0:
                # frac prices the vector wave-instructions of a generation (count per env-step from the profiled child: same work) at 4 issue cycles against the shader
                # cycles the TIMED launches themselves took (stamped in the kernel); clock_ghz = those cycles / the same launches' HIP-event time.
                cyc = wc['shader_cycles_per_generation']
                wi = valu_in_run['wave_instructions_per_env_step'] * live_per_gen
                valu_in_run['shader_cycles_per_generation'] = cyc
                valu_in_run['frac'] = wi * 4. / (1024. * cyc)
                valu_in_run['clock_ghz'] = wc['clock_ghz']
                valu_in_run['frac_is'] = ('vector wave-instructions per generation (per-env-step count of the profiled child x live instances per generation of the timed windows) x 4 '
                                          'issue cycles / (1024 SIMDs x shader cycles per generation stamped by the timed launches themselves); clock_ghz = stamped cycles / HIP-event time')
            else:
                valu_in_run['frac'] = valu_in_run['frac_profiled_launch']
                valu_in_run['clock_ghz'] = valu_in_run['clock_ghz_profiled_launch']
                valu_in_run['frac_is'] = 'the profiled launch\'s own ratio (no stamped cycles of the timed windows)'
            if valu_in_run.get('f64_share') and valu_in_run.get('active_lanes_per_instruction'):
                # issue slots are not useful work: the share of float64 arithmetic among the vector instructions x the lanes that are switched on
                valu_in_run['useful_f64_frac'] = valu_in_run['frac'] * valu_in_run['f64_share'] * valu_in_run['active_lanes_per_instruction'] / 64.
        out = {
            'metric': 'env-steps/sec (instances x gens/s), RLEPSO bbob-easy d=10', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': elapsed_max / K * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            # the K-generation window measured `repeats` times (see timed_window); value / ms_per_step are the median repeat's, spread = (max - min) / median
            # of the repeats' wall times, timed_region_s = the sum of all timed brackets
            'backend': (dist.get_backend() if dist is not None else None),
            'ranks_seen': [{'rank': int(r[0]), 'device': int(r[1])} for r in diag_rows],
            'per_rank': {'ms_per_step': [r[2] for r in diag_rows], 'live_env_steps': [r[3] for r in diag_rows], 'kernel_us_per_generation': [r[4] for r in diag_rows]},
            # spread: (max - min) / median of the repeats' wall times WITHOUT the first repeat (a process's first window runs in the chip's low-clock regime and pays
            # first-touch costs; it is a timed repeat like the others -- the median does not care -- but it says nothing about the measurement's stability)
            'repeats': repeats, 'spread': float((times[1:].max() - times[1:].min()) / np.median(times[1:])) if repeats > 2 else 0.0,
            'spread_with_first_repeat': float((times.max() - times.min()) / np.median(times)) if repeats > 1 else 0.0,
            'timed_region_s': float(times.sum()),
            'repeat_ms_per_step': {'min': float(times.min() / K * 1e3), 'median': float(np.median(times) / K * 1e3), 'max': float(times.max() / K * 1e3),
                                   'p10': float(np.percentile(times, 10) / K * 1e3), 'p50': float(np.percentile(times, 50) / K * 1e3), 'p90': float(np.percentile(times, 90) / K * 1e3),
                                   'slowest_repeat': int(times.argmax()), 'max_without_first': float(times[1:].max() / K * 1e3) if repeats > 1 else None},
            'config': {'workload': f'RLEPSO_Agent + RLEPSO_Optimizer, bbob dim=10 pop=100, {B} lock-step instances per GPU '
                                   f'({fn_desc} round-robin x seeds), maxFEs=20000 (199 generations/episode), '
                                   f'{stop_desc}, policy = exported bbob_easy RLEPSO weights sampled on device',
                       'instances_per_gpu': B, 'live_env_steps': live_all, 'parallelism': f'instances sharded x{world}',
                       'policy': {'resident': f'mbx_rlepso_rollout: up to {args.gens_per_launch} generations of act + step per launch, the instance state stays in '
                                              'LDS / registers between generations (read once, written once per launch); actions drawn in the kernel from the '
                                              'actor (mu, sigma) table built by mbx_rlepso_policy_table; bit-identical to one mbx_rlepso_act_step launch per generation',
                                  'fused': 'act + step in one launch (mbx_rlepso_act_step): the generation kernel draws its action from the '
                                           'actor (mu, sigma) table built by mbx_rlepso_policy_table (actor evaluated at every reachable '
                                           'state fes/maxFEs)',
                                  'hip': 'mbx_gauss_policy (both MLPs over the whole batch, one launch) + mbx_step per generation',
                                  'torch': 'both actor MLPs as batched PyTorch ops every generation' + (', hipGraph replay' if args.graph_policy else ''),
                                  'table': '(mu, sigma) gathered from the per-fes table with PyTorch ops' + (', hipGraph replay' if args.graph_policy else '')}[args.policy],
                       'kernel_timing': (f'every launch of every timed window bracketed by its own HIP event pair on the launch stream ({len(launches)} launches in {repeats} repeats), '
                                         f'minus the cost of an empty event pair ({pair_ms * 1e3:.1f} us)') if resident else
                                        (f'one HIP event every {stride} generations on the launch stream; consecutive events bracket {stride} back-to-back '
                                         f'generation kernels (all of them are covered), minus the cost of an empty event pair ({pair_ms * 1e3:.1f} us)'),
                       'policy_table_build_us': table_build_us,
                       # the policy work north_star puts on the step path, memoised: the actor is a function of ONE scalar state (fes / maxFEs), so it is evaluated once
                       # per weight set at every reachable fes (one mbx_rlepso_policy_table launch) and every generation of every instance looks its row up
                       'policy_table_amortisation': (f'one build per weight set = per {EPISODE_GENS} generations x {B} instances ({EPISODE_GENS * B} env-steps) in a rollout: '
                                                     f'{(table_build_us or 0.) / EPISODE_GENS:.2f} us per generation if charged to the step path, outside the timed window; '
                                                     f'training rebuilds it after every optimizer step (n_step = 10 generations)'),
                       'timed_window': f'{K} consecutive lock-step generations starting at generation {first_gen} of an episode of {EPISODE_GENS} '
                                       f'(episodes restart with mbx_reset inside the window when it is longer)'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_measured_in_run': traffic_in_run,
                         'traffic_source': (f'collected during this run: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of a child bench.py --steps 20 --warmup 2 '
                                            f'(every instance live): {traffic_note}; scaled to the env-steps of an average launch of this run') if traffic_in_run else
                                           ((f'profiles/{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, scaled to this '
                                             f"run's live instances (not collected during this run" + (f': {traffic_note}' if traffic_note else '') + ')') if traffic_src else None),
                         'kernel': 'k_rlepso_run<256, 100, 10, 5, true>' if resident else 'k_rlepso_step<256, 100, 10, 5, true>',
                         # FDR exemplar: exact by default (index = the reference's np.argmin of rounded quotients on any input); the MBX_F_FDR_FAST kernels on the same window
                         'fdr': 'exact (default flags)', 'fdr_fast_ms_per_step': fdr_fast,
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'avg_kernel_us': avg_kernel_s * 1e6, 'algorithmic_bytes_per_env_step': ALGO_BYTES_PER_STEP,
                         'env_steps_per_launch': live_per_launch, 'generations_per_launch': gens_per_launch,
                         'avg_generation_us': avg_gen_s * 1e6, 'live_instances_per_generation': live_per_gen,
                         # the kernel is bound by vector-instruction issue, not by HBM: `valu` is the fraction of SIMD cycles that issue a VALU
                         # instruction at the clock the chip really sustained (measured in this run when rocprofv3 is there; else priced from the committed profile)
                         'binding': 'valu',
                         'valu': valu_in_run or valu_roofline(live_per_gen, avg_gen_s, resident),
                         'policy_mfma': policy_mfma},
        }
        if side_legs is not None:
            out.update(side_legs)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
