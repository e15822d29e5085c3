set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
(time python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/r05b/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/r05b/pytest_gpu.log
tail -n 4 gpurun_out/r05b/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r05b/bench_steps20_warmup5.json 2> gpurun_out/r05b/bench_steps20.err
python -c "
import json; d=json.load(open('gpurun_out/r05b/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['valu'], indent=1))"
bash tools/bench_variants.sh r05b > gpurun_out/r05b/variants_table.txt 2>&1
tail -n 30 gpurun_out/r05b/variants_table.txt
