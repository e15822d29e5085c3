cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05i
python -m pytest tests/test_bench_contract.py tests/test_gpu_rlepso.py tests/test_fdr_ties.py -m gpu -x -q -s > gpurun_out/r05i/pytest.log 2>&1; echo "rc=$?"
grep -n "passed\|failed\|FAILED\|Error\|exact-FDR\|ulp apart" gpurun_out/r05i/pytest.log | cut -c1-200 | tail -n 20
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05i/bench_steps20_warmup5.json 2> gpurun_out/r05i/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r05i/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline']['policy_mfma'], indent=1)); print(d['plugin_view']['ms_per_env_step']); v=d['roofline']['valu']; print(v['frac'], v['clock_ghz'], v['shader_cycles_per_generation'])"
tail -n 5 gpurun_out/r05i/bench.err
