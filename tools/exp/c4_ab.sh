# config 4 (DE-DDQN on protein docking, one GPU's share) same-box A/B:  MBX_LIBS="a.so b.so" bash tools/exp/c4_ab.sh   (run through gpurun; parity tests of the working tree first)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ddqn.py tests/test_protein.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do for lib in $MBX_LIBS; do echo "$lib $(MBX_LIB=$PWD/$lib timeout 300 python tools/exp/dq_phases.py 2>/dev/null | tail -1 | cut -c1-120)"; done; done
