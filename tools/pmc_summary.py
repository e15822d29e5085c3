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
step = out['kernels'].get('mbx::k_rlepso_step', {})
if 'FETCH_SIZE' in step and 'WRITE_SIZE' in step:
    out['calibration'] = {
        'note': 'PMC passes ran generations 3..22 of an episode, i.e. every instance live.  Calibration (profiles/r01a_*): '
                'k_rlepso_reset writes 26.5 KB x 4096 = 108.6 MB and reads almost nothing: WRITE_SIZE reported 106.7 MB => 1:1 for '
                'these 8-byte coalesced stores; k_rlepso_step reads the same 108.6 MB: FETCH_SIZE reported 55.0 MB => the gfx950 x2 '
                'under-report of MI355X_MICROARCH.md applies.  hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 for k_rlepso_step.',
        'hbm_bytes_per_launch': (2 * step['FETCH_SIZE']['mean'] + step['WRITE_SIZE']['mean']) * 1024}
    if 'SQ_ACTIVE_INST_VALU' in step and 'SQ_BUSY_CYCLES' in step:
        out['calibration']['valu_wave_instructions_per_launch'] = step['SQ_INSTS_VALU']['mean']
print(json.dumps(out, indent=1))
