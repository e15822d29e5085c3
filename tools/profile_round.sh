#!/bin/bash
# Round profile on the GPU box:  bash tools/profile_round.sh r01c
# Writes gpurun_out/<tag>/: bench.json, kernel_stats.csv (rocprofv3 --kernel-trace --stats), pmc_hbm_traffic.json (three separate
# --pmc passes, no trace domains, summarised by tools/pmc_summary.py).  Copy what should be judged into profiles/.
set -u
TAG=${1:-r01x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/bench_steps20_warmup5.json" 2>> "$OUT/bench.err"     # the driver's window
# headline command alone (no plugin_view / other_configs legs: they launch the same kernels on other batch sizes and would pollute the averages)
# (--warmup 0 --repeats 5: every k_rlepso_run launch in the trace is a timed one, 5 x 199 generations in 5 whole-episode launches -- tools/profile_recompute.py divides)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python "$ROOT/bench.py" --steps 199 --warmup 0 --repeats 5 --no-cpu-baseline --no-other-configs --no-pmc --no-fdr-fast > "$OUT/trace.log" 2>&1
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
# the side legs (B = 1 plugin view, configs 3 / 4 / 5) in a trace of their own
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace2" -o t -- python "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > "$OUT/trace2.log" 2>&1
find "$OUT/trace2" -name '*kernel_stats.csv' -exec cp {} "$OUT/other_configs_kernel_stats.csv" \;
# (one pass per quoted group: TCC counters cannot share a pass; 8 SQ counters per pass; no trace domains next to --pmc)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
            "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -o p -- python "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-other-configs --no-pmc --no-fdr-fast > "$OUT/pmc_$name.log" 2>&1
done
python "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/pmc_hbm_traffic.json"
python "$ROOT/tools/profile_recompute.py" "$OUT" > "$OUT/recompute.md" 2> "$OUT/recompute.err"      # paste into profiles/README.md
rm -rf "$OUT"/trace "$OUT"/trace2 "$OUT"/pmc_*/ 2>/dev/null
ls -la "$OUT"
