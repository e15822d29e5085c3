#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSVs (tools/profile_round.sh) into the per-kernel JSON kept under profiles/.

HBM bytes per launch follow MI355X_MICROARCH.md's gfx950 recipe: FETCH_SIZE and WRITE_SIZE are reported in KB; FETCH_SIZE
under-reports these 8-byte coalesced loads by x2 on gfx950 (calibrated in profiles/README.md), WRITE_SIZE is 1:1:
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import csv
import re
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r'<[^>]*>', '', row['Kernel_Name'].split('(')[0]).replace('void ', '').strip()     # templated kernels: 'void mbx::k<256>'
            if 'mbx::' in name:
                vals[name][row['Counter_Name']].append(float(row['Counter_Value']))
out = {'command': 'python bench.py --steps 20 --warmup 2 --no-cpu-baseline under rocprofv3 --pmc <counters> '
                  '(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*)',
       'unit': 'FETCH_SIZE / WRITE_SIZE in KB per dispatch as reported by rocprofv3', 'kernels': {}}
for k, cs in sorted(vals.items()):
    out['kernels'][k] = {c: {'dispatches': len(v), 'mean': sum(v) / len(v), 'min': min(v), 'max': max(v)} for c, v in sorted(cs.items())}
# The dominant kernel: the resident rollout kernel (bench.py's default route, one dispatch = the --steps generations of the timed window;
# the shorter warm-up dispatch is left out by taking each counter's maximum over the dispatches) or, with --policy fused, k_rlepso_step.
GENS = int(os.environ.get('PMC_GENS', '20'))
INST = int(os.environ.get('PMC_INSTANCES', '4096'))
kernel = 'mbx::k_rlepso_run' if 'mbx::k_rlepso_run' in out['kernels'] else 'mbx::k_rlepso_step'
step = out['kernels'].get(kernel, {})
if kernel == 'mbx::k_rlepso_run':
    step = {c: dict(v, mean=v['max']) for c, v in step.items()}
if 'FETCH_SIZE' in step and 'WRITE_SIZE' in step:
    out['calibration'] = {
        'kernel': kernel, 'env_steps_per_launch': INST * (GENS if kernel == 'mbx::k_rlepso_run' else 1),
        'note': 'PMC passes ran generations 3..22 of an episode, i.e. every instance live.  Calibration (profiles/r01a_*): '
                'k_rlepso_reset writes 26.5 KB x 4096 = 108.6 MB and reads almost nothing: WRITE_SIZE reported 106.7 MB => 1:1 for '
                'these 8-byte coalesced stores; k_rlepso_step reads the same 108.6 MB: FETCH_SIZE reported 55.0 MB => the gfx950 x2 '
                'under-report of MI355X_MICROARCH.md applies.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 for k_rlepso_step.',
        'hbm_bytes_per_launch': (2 * step['FETCH_SIZE']['mean'] + step['WRITE_SIZE']['mean']) * 1024}
    if 'SQ_ACTIVE_INST_VALU' in step and 'SQ_BUSY_CYCLES' in step:
        out['calibration']['valu_wave_instructions_per_launch'] = step['SQ_INSTS_VALU']['mean']
    if 'SQ_INSTS_VALU_ADD_F64' in step and 'SQ_INSTS_VALU_INT32' in step and 'SQ_INSTS_VALU' in step:
        # VALU issue bound of the generation kernel: wave-instructions by class x the issue cost of the class measured with
        # tools/ubench/valu_rates.hip on this GPU (profiles/r02_valu_issue_rates.txt; cycles per wave-instruction per SIMD at saturation):
        #   float64 add / mul / fma / min / cmp, v_mov_b64, cvt, v_mad_u64_u32, every compare / select / min / max / shift / 3-operand
        #   integer op: 4;  float64 rcp / sqrt: 16;  float32 add / mul / fma, v_mov_b32, 32-bit add / sub / and / or / xor: 2;  float32 rcp etc.: 8.
        # SQ_INSTS_VALU_INT32 mixes 2-cycle (add, logic) and 4-cycle (shift, mul, compare) instructions and the remainder of SQ_INSTS_VALU
        # (moves, selects, compares, lane ops) is not broken down by the counters, so the bound is an interval: both at 2 / both at 4.
        m = lambda c: step[c]['mean'] if c in step else 0.
        f64 = m('SQ_INSTS_VALU_ADD_F64') + m('SQ_INSTS_VALU_MUL_F64') + m('SQ_INSTS_VALU_FMA_F64')
        f32 = m('SQ_INSTS_VALU_ADD_F32') + m('SQ_INSTS_VALU_MUL_F32') + m('SQ_INSTS_VALU_FMA_F32')
        fixed = 4 * (f64 + m('SQ_INSTS_VALU_INT64') + m('SQ_INSTS_VALU_CVT')) + 16 * m('SQ_INSTS_VALU_TRANS_F64') + 2 * f32 + 8 * m('SQ_INSTS_VALU_TRANS_F32')
        i32 = m('SQ_INSTS_VALU_INT32')
        other = m('SQ_INSTS_VALU') - (f64 + f32 + m('SQ_INSTS_VALU_INT64') + m('SQ_INSTS_VALU_CVT') + m('SQ_INSTS_VALU_TRANS_F64') + m('SQ_INSTS_VALU_TRANS_F32') + i32)
        simds, clock = 256 * 4, 2.4e9
        lo, hi = fixed + 2 * (i32 + other), fixed + 4 * (i32 + other)
        out['valu_issue_bound'] = {
            'wave_instructions_per_launch': {'total': m('SQ_INSTS_VALU'), 'f64_add_mul_fma': f64, 'f64_trans': m('SQ_INSTS_VALU_TRANS_F64'), 'f32_add_mul_fma': f32,
                                             'f32_trans': m('SQ_INSTS_VALU_TRANS_F32'), 'int64': m('SQ_INSTS_VALU_INT64'), 'cvt': m('SQ_INSTS_VALU_CVT'), 'int32': i32,
                                             'other_mov_cmp_select_lane': other},
            'simd_cycles_per_launch': [lo, hi],
            'issue_bound_us_at_2.4GHz': [lo / simds / clock * 1e6, hi / simds / clock * 1e6],
            'active_lanes_per_valu_instruction': m('SQ_THREAD_CYCLES_VALU') / m('SQ_INSTS_VALU') if m('SQ_THREAD_CYCLES_VALU') else None,
            'note': 'issue cost per class from tools/ubench/valu_rates.hip (profiles/r02_valu_issue_rates.txt); the interval prices SQ_INSTS_VALU_INT32 and the '
                    'uncategorised remainder at 2 (low) or 4 (high) cycles; 1024 SIMDs at the nominal 2.4 GHz (the sustained clock under this load is lower, MI355X_MICROARCH.md DVFS note, so the wall-clock bound is higher)'}
print(json.dumps(out, indent=1))
