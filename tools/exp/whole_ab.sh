# whole-episode A/B of libraries (headline workload, 2 episodes):  MBX_LIBS="a.so b.so" bash tools/exp/whole_ab.sh   (run through gpurun)
cd $GRAFT_REPO_ROOT
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'; }
for rep in 1 2 3; do for lib in $MBX_LIBS; do
echo "$lib whole: $(MBX_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-pmc --no-fdr-fast 2>/dev/null | line)"
done; done
