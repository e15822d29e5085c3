set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05g
(time python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/r05g/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/r05g/pytest_gpu.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r05g/pytest_gpu.log | tail -n 10
bash tools/profile_round.sh r05g > gpurun_out/r05g/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r05g/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['valu'], indent=1))
d=json.load(open('gpurun_out/r05g/bench.json')); print(d['value'], d['ms_per_step']); print([ (o['config'][:40], o['ms_per_step'], o['roofline_frac']) for o in d['other_configs']])"
cat gpurun_out/r05g/recompute.md | head -n 30
