#!/bin/bash
# Ad-hoc SQ counter probe of k_rlepso_step on the kernel micro-benchmark:  bash tools/pmc_probe.sh "CTR_A CTR_B" "CTR_C ..."
# (one rocprofv3 --pmc pass per argument; counters only, no trace domains).  Prints per-launch means.
# PROBE_CMD / PROBE_KERNEL select another micro-benchmark / kernel, e.g. PROBE_CMD="tools/kbench_lde_funcs.py" PROBE_KERNEL=k_lde_step.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
[ $# -eq 0 ] && set -- "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS"
for c in "$@"; do
rm -rf /tmp/pm_out
rocprofv3 --pmc $c --output-format csv -d /tmp/pm_out -o p -- python $ROOT/${PROBE_CMD:-tools/kbench.py --steps 10} > /tmp/pm_log 2>&1
PROBE_KERNEL=${PROBE_KERNEL:-k_rlepso_step} python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(list)
for f in glob.glob('/tmp/pm_out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if os.environ['PROBE_KERNEL'] in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print({k: round(sum(v) / len(v)) for k, v in acc.items()} or open('/tmp/pm_log').read()[-400:])
PY
done
