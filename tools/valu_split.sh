#!/bin/bash
# VALU instruction budget of k_rlepso_step by ablation:  bash tools/valu_split.sh   (needs metabox_amd/csrc/variants/libmbx_*.so)
# One rocprofv3 --pmc pass per build (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES only; no trace domains).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for l in libmbx.so variants/libmbx_noFDR.so variants/libmbx_noEVAL.so variants/libmbx_noRANK.so variants/libmbx_skel.so; do
    [ -f "$ROOT/metabox_amd/csrc/$l" ] || continue
    rm -rf /tmp/vs_out
    MBX_LIB=$ROOT/metabox_amd/csrc/$l rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/vs_out -o p -- python $ROOT/tools/kbench.py --steps 10 > /dev/null 2>&1
    python - "$l" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/vs_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_rlepso_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[1], {k: round(sum(v) / len(v) / 4096) for k, v in acc.items()}, 'per instance-generation')
PY
done
