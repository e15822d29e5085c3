# same-box A/B of the headline (driver window + whole episodes) and config 5:  bash tools/exp/ab_lib.sh build/libmbx_<variant>.so   (run through gpurun)
cd $GRAFT_REPO_ROOT
VAR=$1
MBX_LIB=$PWD/$VAR python -m pytest tests/test_gpu_rlepso.py -m gpu -x -q -k "tape_replay_matches_reference_episodes or resident_rollout_equals_one_launch_per_generation or philox_parity" 2>&1 | tail -n 2
for rep in 1 2 3; do
for lib in metabox_amd/csrc/libmbx.so $VAR; do
  echo "== $lib $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["avg_generation_us"])')"
done; done
for lib in metabox_amd/csrc/libmbx.so $VAR; do
  echo "   whole episodes $lib: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])')"
  echo "   config 5 $lib: $(MBX_LIB=$PWD/$lib python tools/kbench_config5.py --steps 20 2>/dev/null | grep resident | cut -c1-200)"
done
