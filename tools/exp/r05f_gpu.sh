set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05f
# the tie-flag build: adversarial + natural FDR tests, tape replay and resident parity
MBX_LIB=$PWD/build/libmbx_tieflag.so python -m pytest tests/test_fdr_ties.py tests/test_gpu_rlepso.py -m gpu -x -q -s -k "fdr or tape_replay_matches_reference_episodes or resident_rollout_equals_one_launch_per_generation or philox_parity or oracle" > gpurun_out/r05f/pytest_tieflag.log 2>&1; echo "rc=$?"
grep -n "ulp apart\|natural\|passed\|failed\|tape replay" gpurun_out/r05f/pytest_tieflag.log | cut -c1-220
for rep in 1 2 3; do
for lib in metabox_amd/csrc/libmbx.so build/libmbx_tieflag.so; do
  echo "== $lib $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["avg_generation_us"])')"
done; done
for lib in metabox_amd/csrc/libmbx.so build/libmbx_tieflag.so; do
  echo "   whole episodes $lib: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  echo "   config 5 $lib: $(MBX_LIB=$PWD/$lib python tools/kbench_config5.py --steps 20 2>/dev/null | grep resident | cut -c1-200)"
done
python tools/exp/clock_windows.py > gpurun_out/r05f/clock_windows.jsonl 2>&1
