#!/bin/bash
# MFMA-pipe utilisation of the kernels that use the matrix cores (k_qnet_argmax, k_lstm_policy: float32; k_lde_step: float64 matvec), from one rocprofv3 --pmc pass over
# bench.py's other-config legs:   bash tools/exp/mfma_util.sh <tag>   -> gpurun_out/<tag>/mfma_utilisation.json
# utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles); the kernel's duration comes from GRBM_GUI_ACTIVE.
OUT=gpurun_out/${1:-mfma}; mkdir -p $OUT
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/mfma_pmc -o p -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > /tmp/mfma_pmc.log 2>&1
python - "$ROOT/$OUT/mfma_utilisation.json" <<'PY'
import csv, glob, json, re, sys, collections
v = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('/tmp/mfma_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        if any(k in n for k in ('k_qnet_argmax', 'k_lstm_policy', 'k_lde_step', 'k_dq_step', 'k_rlepso_run')):
            v[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in sorted(v.items()):
    m = {c: sum(x) / len(x) for c, x in cs.items()}
    m['dispatches'] = len(next(iter(cs.values())))
    if m.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        m['mfma_pipe_utilisation'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * m['GRBM_GUI_ACTIVE'])      # 1024 SIMDs; BUSY_CYCLES summed over them
    out[k] = m
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps({k: {a: round(b, 4) if isinstance(b, float) else b for a, b in m.items()} for k, m in out.items()}, indent=1))
PY
