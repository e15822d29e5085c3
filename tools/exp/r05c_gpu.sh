set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
(time python -m pytest tests/test_gpu_lde.py tests/test_protein.py tests/test_training_parity.py tests/test_bench_contract.py -m gpu -q -s --durations=12 > gpurun_out/r05c/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/r05c/pytest_gpu.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r05c/pytest_gpu.log | tail -n 20
python bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline > gpurun_out/r05c/bench_steps20_warmup5.json 2> gpurun_out/r05c/bench_steps20.err
python -c "
import json; d=json.load(open('gpurun_out/r05c/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['valu'], indent=1))"
