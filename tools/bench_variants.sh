#!/bin/bash
# SURVEY §8(d) C2 variants of the headline bench, same timed loop:  bash tools/bench_variants.sh r01f
# all 24 functions / the 18-function bbob-easy train split, each with the reference stop rule and with a fixed 199-generation
# horizon, then every function on its own (fixed horizon).  One JSON line per run in gpurun_out/<tag>_variants.jsonl.
set -u
TAG=${1:-r01x}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/${TAG}_variants.jsonl
mkdir -p "$ROOT/gpurun_out"; : > "$OUT"
cd "$ROOT"
for fn in all24 train18; do
    python bench.py --no-cpu-baseline --no-pmc --no-other-configs --functions $fn >> "$OUT" 2>> "$OUT.err"
    python bench.py --no-cpu-baseline --no-pmc --no-other-configs --functions $fn --fixed-horizon >> "$OUT" 2>> "$OUT.err"
done
for f in $(seq 1 24); do
    python bench.py --no-cpu-baseline --no-pmc --no-other-configs --functions $f --fixed-horizon --steps 199 --warmup 10 >> "$OUT" 2>> "$OUT.err"
done
python - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    d = json.loads(line)
    w = d['config']['workload']
    print(f"{w[w.index('(') + 1:w.index(' round-robin')]:45s} {'fixed' if 'fixed horizon' in w else 'stop-rule':9s} "
          f"value {d['value']:.3e}  {d['roofline'].get('avg_generation_us', d['roofline']['avg_kernel_us']):.1f} us per generation  live instances per generation {d['roofline'].get('live_instances_per_generation', float('nan')):.0f}")
PY
