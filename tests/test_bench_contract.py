"""bench.py's output contract: one JSON line with the driver's keys plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_worker_runs_whole_episodes(tmp_path):
    """oracle/cpu_workload.py (one worker of bench.py's cpu_baseline leg): a flat policy table, half a second of episodes."""
    table = np.concatenate([np.full((20202, 35), 0.5, np.float32), np.full((20202, 35), 0.2, np.float32)], axis=1)
    path = tmp_path / 't.npy'
    np.save(path, table)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'cpu_workload.py'), '--table', str(path), '--seconds', '0.5',
                          '--worker', '3'], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-400:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r['episodes'] >= 1 and r['steps'] >= 100 and r['seconds'] > 0
    assert r['steps'] <= r['episodes'] * 199                                    # an episode is at most 199 generations


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '40', '--warmup', '4', '--no-cpu-baseline'],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-600:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 40 and d['warmup'] == 4 and d['scaling'] == 'weak' and d['dtype'] == 'f64'
    assert d['vs_baseline'] is None and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    # value = live env-steps / wall time of the timed region
    assert abs(d['value'] - d['config']['live_env_steps'] / (d['ms_per_step'] * 1e-3 * d['steps'])) <= 1e-6 * d['value']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['avg_kernel_us'] * 1e-6) / 1e9) <= 1e-6 * r['achieved']
    assert r['algorithmic_bytes_per_env_step'] == 54057 and 0 < r['avg_kernel_us'] < 1e4
    # default route: the resident rollout kernel, 40 generations in one launch; its state never leaves the chip between generations, so the
    # measured HBM traffic (when a profile of this kernel is committed) is far BELOW the algorithmic bytes of a per-generation round trip
    assert r['kernel'].startswith('k_rlepso_run') and r['generations_per_launch'] == 40 and 'mbx_rlepso_rollout' in d['config']['policy']
    # (env_steps_per_launch is the mean over the repeats, live_env_steps the median repeat's: episodes differ between repeats)
    assert abs(r['env_steps_per_launch'] - d['config']['live_env_steps']) <= 2e-3 * r['env_steps_per_launch']
    # a window shorter than 0.5 s is measured >= 30 times; the line carries the median repeat and the spread
    assert d['repeats'] >= 30 and 0 <= d['spread'] < 1.0 and d['spread'] <= d['spread_with_first_repeat'] and d['timed_region_s'] >= d['repeats'] * d['repeat_ms_per_step']['min'] * 1e-3 * d['steps'] * 0.999
    assert d['repeat_ms_per_step']['min'] <= d['ms_per_step'] <= d['repeat_ms_per_step']['max']
    # the HBM traffic is collected during the run (two rocprofv3 --pmc child passes) when rocprofv3 is there, else taken from the committed profile
    assert isinstance(r['traffic_measured_in_run'], bool) and ('collected during this run' in (r['traffic_source'] or '')) == r['traffic_measured_in_run']
    assert r['traffic'] is None or 0 < r['traffic'] / r['algorithmic_bytes_per_launch'] < 1.0
    # event brackets are net of the empty-pair cost: the launch cannot take longer than the steps it contains
    assert r['avg_kernel_us'] <= d['ms_per_step'] * 1e3 * r['generations_per_launch'] * 1.02
    assert abs(r['avg_generation_us'] * r['generations_per_launch'] - r['avg_kernel_us']) <= 1e-6 * r['avg_kernel_us']
    assert 'generation' in d['config']['timed_window'] and 'event pair' in d['config']['kernel_timing']
    v = r['valu']
    assert r['binding'] == 'valu'
    if v is not None and v.get('measured_in_run'):      # collected during the run: a point value at the measured clock
        assert v['bound'] == 'valu' and 0.2 < v['frac'] < 1.0 and 1.2 < v['clock_ghz'] <= 2.45 and v['wave_instructions_per_generation'] > 1e6
        assert 5e3 < v['wave_instructions_per_env_step'] < 5e4 and 30 < v['active_lanes_per_instruction'] <= 64
        # the cycles are stamped by the timed launches themselves (mbx_debug_clock_slots): one stamp set per launch of every repeat
        tw = v['timed_window_clock']
        assert tw["stamped_launches"] == d["repeats"] and 2e5 < tw['shader_cycles_per_generation'] < 4e5 and abs(v['frac'] - v['frac_profiled_launch']) < 0.1
    else:
        assert v is None or (v['bound'] == 'valu' and 0 < v['frac'][0] <= v['frac'][1] and v['wave_instructions_per_generation'] > 1e6)
    pm = r['policy_mfma']
    # config 4's Q-network launch, measured in the run when rocprofv3 is there (a scalar utilisation), else the committed profile (per-kernel dict)
    assert pm is None or (pm['measured_in_run'] is True and 0 < pm['utilisation'] < 1 and 'k_qnet_argmax' in pm['kernel']) or \
        (pm['measured_in_run'] is False and all(0 < u < 1 for u in pm['utilisation'].values()))
    assert d['config']['policy_table_build_us'] > 0
    assert d['backend'] is None and d['ranks_seen'] == [{'rank': 0, 'device': 0}] and len(d['per_rank']['ms_per_step']) == 1
    oc = d['other_configs']
    assert isinstance(oc, list) and len(oc) == 6, oc
    assert [e['config'].split(':')[0] for e in oc] == ['config 3', 'config 3', 'config 4', 'config 5', 'reference default', 'reference default']
    assert oc[4]['launch_info']['kernel'] == 'k_lde_run<50, 10>' and oc[4]['launch_info']['resident'] and oc[5]['launch_info']['kernel'].startswith('k_rlepso_run<256, 100, 12') and oc[5]['launch_info']['resident']
    c4 = oc[2]['compute_roofline']
    assert 0 < c4['frac_executed'] < c4['frac'] < 1 and c4['executed_flops_per_env_step'] == 4950 * 35 + 10800
    assert all(e['launch_info']['kernel'].startswith('k_lde_run') and e['launch_info']['resident'] for e in oc[:2])
    assert 'fdr_fast_ms_per_step' in d['roofline'] and 0.8 < d['roofline']['fdr_fast_ms_per_step'] / d['ms_per_step'] < 1.02      # the MBX_F_FDR_FAST kernels on the same window: a side field
    for e in oc:
        assert e['ms_per_step'] > 0 and abs(e['env_steps_per_s'] - e['instances'] / (e['ms_per_step'] * 1e-3)) <= 1e-6 * e['env_steps_per_s']
        assert 0 < e['roofline_frac'] < 1


@pytest.mark.gpu
def test_bench_one_launch_per_generation_route():
    """--policy fused: one mbx_rlepso_act_step launch per generation (the route the resident kernel is bit-identical to); a launch is a
    generation, so the per-launch and per-generation figures coincide and the traffic is the state block's round trip."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '24', '--warmup', '4', '--no-cpu-baseline',
                          '--no-other-configs', '--policy', 'fused'], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert out.returncode == 0, out.stderr[-600:]
    d = json.loads([l for l in out.stdout.splitlines() if l.strip()][0])
    r = d['roofline']
    assert r['kernel'].startswith('k_rlepso_step') and r['generations_per_launch'] == 1 and 'mbx_rlepso_act_step' in d['config']['policy']
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['avg_kernel_us'] * 1e-6) / 1e9) <= 1e-6 * r['achieved']
    assert r['avg_kernel_us'] <= d['ms_per_step'] * 1e3 * 1.02 and 'every 2 generations' in d['config']['kernel_timing']
    assert r['traffic'] is None or 0.5 < r['traffic'] / r['algorithmic_bytes_per_launch'] < 2.0


@pytest.mark.gpu
def test_bench_under_torchrun_two_ranks_one_device():
    """The driver's multi-GPU launch line (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`) with N = 2
    ranks sharing the one GPU of the test box (--same-device; gloo, because RCCL rejects two ranks on one device): rank 0 prints ONE line,
    the aggregate counts both ranks' instances, weak scaling."""
    port = 36500 + os.getpid() % 2000
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '12', '--warmup', '3',
                          '--same-device', '--dist-backend', 'gloo', '--no-cpu-baseline'],
                         capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-400:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['scaling'] == 'weak' and d['steps'] == 12
    assert d['config']['parallelism'] == 'instances sharded x2' and d['config']['instances_per_gpu'] == 4096
    assert 12 * 4096 < d['config']['live_env_steps'] <= 2 * 12 * 4096          # both ranks' instances are in the aggregate
    assert 'other_configs' not in d and 'cpu_baseline' not in d                 # side legs only at N = 1
    # the multi-rank line diagnoses itself: backend, which rank ran on which device, every rank's own window time and env-steps
    assert d['backend'] == 'gloo' and sorted(r['rank'] for r in d['ranks_seen']) == [0, 1] and all(r['device'] == 0 for r in d['ranks_seen'])
    pr = d['per_rank']
    assert len(pr['ms_per_step']) == len(pr['live_env_steps']) == len(pr['kernel_us_per_generation']) == 2
    assert abs(sum(pr['live_env_steps']) - d['config']['live_env_steps']) <= 1e-6 * d['config']['live_env_steps']
    assert max(pr['ms_per_step']) <= d['ms_per_step'] * 1.0001 and min(pr['ms_per_step']) > 0
    # a launch whose world size is not --gpus refuses to run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-other-configs', '--no-pmc'],
                         capture_output=True, text=True, cwd=ROOT, timeout=300, env=env)
    assert bad.returncode != 0 and 'WORLD_SIZE' in (bad.stderr + bad.stdout)
