#!/bin/bash
# PMC passes over any command:  bash tools/exp/pmc.sh <tag> <kernel-name filter, comma separated> <command ...>
# -> gpurun_out/<tag>/pmc.json.  One rocprofv3 --pmc pass per counter group (8 SQ counters per pass; the TCC counters in passes of their own; no
# trace domains next to --pmc).  Per kernel whose name contains one of the filters: mean counter values per dispatch and the derived fractions
# (same counter set and derivations as tools/exp/lde_pmc.sh, which profiles/r04e_lde_run_pmc.json came from).
TAG=$1; FILTER=$2; shift 2
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
    rm -rf /tmp/pmc_${TAG}_$i
    (cd $ROOT && rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- "$@") > $OUT/pass_$i.log 2>&1
done
python - "$OUT/pmc.json" "$FILTER" "$TAG" "$*" <<'PY'
import csv, glob, json, sys, collections
filters = sys.argv[2].split(',')
v = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(f'/tmp/pmc_{sys.argv[3]}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        if any(k in n for k in filters):
            v[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {'_command': sys.argv[4]}
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
    if m.get('SQ_LDS_IDX_ACTIVE'): m['lds_bank_conflict_frac'] = m.get('SQ_LDS_BANK_CONFLICT', 0.) / m['SQ_LDS_IDX_ACTIVE']
    if 'SQ_THREAD_CYCLES_VALU' in m and m.get('SQ_ACTIVE_INST_VALU'): m['active_lanes_per_valu_inst'] = m['SQ_THREAD_CYCLES_VALU'] / m['SQ_ACTIVE_INST_VALU']
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m: m['hbm_bytes_per_dispatch'] = (2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024
    out[k] = m
json.dump(out, open(sys.argv[1], 'w'), indent=1)
print(json.dumps({k: ({a: (round(b, 4) if b < 100 else round(b)) for a, b in m.items()} if isinstance(m, dict) else m) for k, m in out.items()}, indent=1))
PY
