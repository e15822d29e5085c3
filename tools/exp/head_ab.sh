# same-box A/B of the headline (config 2, driver window): build/libmbx_head.so (the previous commit) against the working tree; parity tests of the resident route first
set -e
timeout 900 python -m pytest tests/test_gpu_rlepso.py -x -q -k "resident or rollout" 2>&1 | tail -3
for rep in 1 2; do
for lib in build/libmbx_head.so metabox_amd/csrc/libmbx.so; do
  echo "== $lib $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
  echo "   whole episodes: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
done; done
