# same-box A/B of whole libraries on the headline driver window (+ config 5):  MBX_LIBS="a.so b.so" bash tools/exp/libs_ab.sh     (run through gpurun)
cd $GRAFT_REPO_ROOT
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"].get("avg_generation_us"))'; }
for rep in 1 2 3; do
for lib in $MBX_LIBS; do
  echo "== $lib window: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | line)"
done; done
for lib in $MBX_LIBS; do
  echo "   config 5 $lib: $(MBX_LIB=$PWD/$lib python tools/kbench_config5.py --steps 20 2>/dev/null | grep resident | cut -c1-200)"
done
