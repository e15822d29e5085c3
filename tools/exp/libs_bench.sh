# headline driver window for a list of library builds on one box:  MBX_LIBS="a.so b.so" bash tools/exp/libs_bench.sh   (us per generation, env-steps/s)
for rep in 1 2; do for lib in $MBX_LIBS; do
  echo "$lib $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"]*1e3,2), round(d["value"]))')"
done; done
