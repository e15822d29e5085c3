# end-of-round GPU job:  TAG=r05k bash tools/exp/round_gpu.sh   (pytest -m gpu, then tools/profile_round.sh $TAG; run through gpurun)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${TAG:-r05x}
(time python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/${TAG:-r05x}/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/${TAG:-r05x}/pytest_gpu.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/${TAG:-r05x}/pytest_gpu.log | tail -n 10
bash tools/profile_round.sh ${TAG:-r05x} > gpurun_out/${TAG:-r05x}/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/${TAG:-r05x}/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['valu'], indent=1))
d=json.load(open('gpurun_out/${TAG:-r05x}/bench.json')); print(d['value'], d['ms_per_step']); print([ (o['config'][:40], o['ms_per_step'], o['roofline_frac']) for o in d['other_configs']])"
cat gpurun_out/${TAG:-r05x}/recompute.md | head -n 30
