# same-box A/B of config 3 (resident route): build/libmbx_head.so (the previous commit) against the working tree; parity tests first
set -e
timeout 600 python -m pytest tests/test_gpu_lde.py -x -q -k "resident or host_loop" 2>&1 | tail -3
for rep in 1 2; do
for lib in build/libmbx_head.so metabox_amd/csrc/libmbx.so; do
  echo "== $lib"
  MBX_LIB=$PWD/$lib timeout 300 python tools/exp/lde_run.py --route resident --gens-per-launch 50 --steps 100 2>&1 | grep -o '"pop": [0-9]*\|"ms_per_generation": [0-9.]*' | paste - -
done; done
