set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
(time python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/r05e/pytest_gpu.log 2>&1); echo "pytest rc=$?" | tee -a gpurun_out/r05e/pytest_gpu.log
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r05e/pytest_gpu.log | tail -n 10
bash tools/profile_round.sh r05e > gpurun_out/r05e/profile_round.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r05e/bench_steps20_warmup5.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline']['valu'], indent=1))
d=json.load(open('gpurun_out/r05e/bench.json')); print(d['value'], d['ms_per_step']); print([ (o['config'][:40], o['ms_per_step'], o['roofline_frac']) for o in d['other_configs']])"
cat gpurun_out/r05e/recompute.md | head -n 30
# LDE: what the random parent gathers cost (bank conflicts): PMC + timing, shipped build against the gather-free ablation build
bash tools/exp/lde_pmc.sh r05e_lde --route resident --gens-per-launch 50 --steps 50 > gpurun_out/r05e/lde_pmc.log 2>&1
MBX_LIB=$PWD/build/libmbx_lde_abl64.so bash tools/exp/lde_pmc.sh r05e_lde_abl64 --route resident --gens-per-launch 50 --steps 50 > gpurun_out/r05e/lde_pmc_abl64.log 2>&1
for lib in metabox_amd/csrc/libmbx.so build/libmbx_lde_abl64.so metabox_amd/csrc/libmbx.so build/libmbx_lde_abl64.so; do MBX_LIB=$PWD/$lib python tools/exp/lde_run.py --route resident --gens-per-launch 50 --steps 100 2>&1 | grep -o '"pop": [0-9]*\|"ms_per_generation": [0-9.]*' | paste - - | sed "s|^|$lib |"; done
