# same-box A/B of the exact (default) FDR scan against the MBX_F_FDR_FAST form, one library: headline driver window, whole episodes, config 5.  Run through gpurun.
cd $GRAFT_REPO_ROOT
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"].get("avg_generation_us"))'; }
for rep in 1 2 3; do
for fast in 0 1; do
  echo "== MBX_FDR_FAST=$fast window: $(MBX_FDR_FAST=$fast timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | line)"
done; done
for fast in 0 1; do
  echo "   whole episodes MBX_FDR_FAST=$fast: $(MBX_FDR_FAST=$fast timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc 2>/dev/null | line)"
  echo "   config 5 MBX_FDR_FAST=$fast: $(MBX_FDR_FAST=$fast python tools/kbench_config5.py --steps 20 2>/dev/null | grep resident | cut -c1-200)"
done
