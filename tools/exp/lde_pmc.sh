#!/bin/bash
# PMC passes over config 3's kernels (k_lde_step / k_lde_run / k_lstm_policy):  bash tools/exp/lde_pmc.sh <tag> [lde_run.py arguments]
# -> gpurun_out/<tag>/lde_pmc.json.  One rocprofv3 --pmc pass per counter group (8 SQ counters per pass; the TCC counters in passes of their own;
# no trace domains next to --pmc).  Per kernel: mean counter values per dispatch and the derived VALU / LDS busy fractions.
TAG=${1:-lde_pmc}; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
            "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F32" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rm -rf /tmp/lde_pmc_$i
    rocprofv3 --pmc $pass --output-format csv -d /tmp/lde_pmc_$i -o p -- python $ROOT/tools/exp/lde_run.py "$@" > $OUT/pass_$i.log 2>&1
done
python - "$OUT/lde_pmc.json" <<'PY'
import csv, glob, json, sys, collections
v = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('/tmp/lde_pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        if any(k in n for k in ('k_lde_step', 'k_lde_run', 'k_lstm_policy')):
            v[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in sorted(v.items()):
    m = {c: sum(x) / len(x) for c, x in cs.items()}
    m['dispatches'] = max(len(x) for x in cs.values())
    g = m.get('GRBM_GUI_ACTIVE')
    if g:
        g = g / 8                                                 # GRBM_GUI_ACTIVE comes summed over the 8 XCDs
        m['kernel_cycles'] = g
        simd_cycles = 1024 * g                                   # 1024 SIMDs
        if 'SQ_ACTIVE_INST_VALU' in m: m['valu_busy_frac'] = 4 * m['SQ_ACTIVE_INST_VALU'] / simd_cycles      # quad-cycles
        if 'SQ_ACTIVE_INST_LDS' in m: m['lds_inst_busy_frac'] = 4 * m['SQ_ACTIVE_INST_LDS'] / simd_cycles
        if 'SQ_LDS_IDX_ACTIVE' in m: m['lds_array_busy_frac'] = m['SQ_LDS_IDX_ACTIVE'] / (256 * g)
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in m: m['mfma_busy_frac'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
        if 'SQ_WAVE_CYCLES' in m: m['waves_per_simd_avg'] = 4 * m['SQ_WAVE_CYCLES'] / simd_cycles
        if 'SQ_WAIT_ANY' in m and 'SQ_WAVE_CYCLES' in m: m['wave_wait_frac'] = m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']
    if 'SQ_THREAD_CYCLES_VALU' in m and m.get('SQ_ACTIVE_INST_VALU'): m['active_lanes_per_valu_inst'] = m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU']
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m: m['hbm_bytes_per_dispatch'] = (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
    out[k] = m
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps({k: {a: (round(b, 4) if b < 100 else round(b)) for a, b in m.items()} for k, m in out.items()}, indent=1))
PY
