set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
(time python -m pytest tests -m gpu -x -q -s --durations=15 > gpurun_out/r05a/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/r05a/pytest_gpu.log
tail -5 gpurun_out/r05a/pytest_gpu.log
# PMC evidence for configs 4 and 5
bash tools/exp/pmc.sh r05a_dq k_dq_step,k_qnet_argmax python tools/kbench_algos.py ddqn > gpurun_out/r05a/pmc_dq.log 2>&1
bash tools/exp/pmc.sh r05a_c5 k_rlepso_run,k_rlepso_step python tools/kbench_config5.py --steps 10 > gpurun_out/r05a/pmc_c5.log 2>&1
tail -3 gpurun_out/r05a/pmc_dq.log gpurun_out/r05a/pmc_c5.log
