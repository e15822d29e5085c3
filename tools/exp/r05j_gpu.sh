cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05j
python -m pytest tests/test_protein.py tests/test_ddqn.py tests/test_harness.py -m gpu -x -q > gpurun_out/r05j/pytest.log 2>&1; echo "rc=$?"; tail -n 3 gpurun_out/r05j/pytest.log
python tools/exp/dq_size_sweep.py 2>&1 | grep instances
python tools/kbench_algos.py ddqn 2>&1 | grep -v launch_info | cut -c1-200
