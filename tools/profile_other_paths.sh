#!/bin/bash
# rocprofv3 --kernel-trace --stats of the non-headline paths:  bash tools/profile_other_paths.sh r01h
# One run of tools/kbench_algos.py per algorithm (LDE config 3, DE-DDQN config 4's share, GLEET, RL-PSO, QLPSO) and of
# tools/kbench_config5.py; the mbx:: kernels' rows of every kernel_stats.csv go to gpurun_out/<tag>_other_paths_kernel_stats.csv.
TAG=${1:-r01x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_other_paths_kernel_stats.csv
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
echo '"Path","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"' > "$OUT"
for what in lde ddqn gleet rlpso qlpso config5; do
    rm -rf /tmp/op_tr
    if [ $what = config5 ]; then cmd="$ROOT/tools/kbench_config5.py"; else cmd="$ROOT/tools/kbench_algos.py $what"; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/op_tr -o t -- python $cmd > /tmp/op_log 2>&1
    f=$(find /tmp/op_tr -name '*kernel_stats.csv' | head -n 1)
    [ -n "$f" ] && grep 'mbx::' "$f" | sed "s/^/\"$what\",/" >> "$OUT"
    grep -h '^{' /tmp/op_log | cut -c1-300
done
