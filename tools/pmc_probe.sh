cd /tmp && export TMPDIR=/tmp
for c in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
rm -rf /tmp/pm_out
rocprofv3 --pmc $c --output-format csv -d /tmp/pm_out -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py --steps 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pm_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_rlepso_step' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v) / len(v)) for k, v in acc.items()})
PY
done
