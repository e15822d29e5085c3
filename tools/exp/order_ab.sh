cd $GRAFT_REPO_ROOT
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do for lib in build/libmbx_w0.so metabox_amd/csrc/libmbx.so; do
echo "$lib whole: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-fdr-fast 2>/dev/null | line)   window: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-pmc --no-fdr-fast 2>/dev/null | line)"
done; done
