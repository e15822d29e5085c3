set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05k
(time python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/r05k/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/r05k/pytest_gpu.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r05k/pytest_gpu.log | tail -n 10
bash tools/profile_round.sh r05k > gpurun_out/r05k/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r05k/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['valu'], indent=1))
d=json.load(open('gpurun_out/r05k/bench.json')); print(d['value'], d['ms_per_step']); print([ (o['config'][:40], o['ms_per_step'], o['roofline_frac']) for o in d['other_configs']])"
cat gpurun_out/r05k/recompute.md | head -n 30
