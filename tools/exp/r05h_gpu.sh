cd $GRAFT_REPO_ROOT
python - <<'PY'
import bench, json, torch
torch.cuda.set_device(0)
for i in range(3):
    r = bench.plugin_view_cost()
    print('plugin_view standalone', round(r['ms_per_env_step'], 4), round(r['ms_per_reset'], 3))
PY
for lib in metabox_amd/csrc/libmbx.so build/libmbx_lde_w8.so build/libmbx_lde_w5.so build/libmbx_lde_w4.so metabox_amd/csrc/libmbx.so; do
  for suite in bbob bbob-noisy; do
    MBX_LIB=$PWD/$lib python tools/exp/lde_run.py --suite $suite --dim 10 --pop 50 --route resident --gens-per-launch 50 --steps 150 2>&1 | grep '^{' | cut -c1-160 | sed "s|^|$lib |"
  done
done
MBX_ROLLOUT_PER_GENERATION=1 python tools/exp/lde_run.py --suite bbob --dim 10 --pop 50 --route resident --gens-per-launch 50 --steps 150 2>&1 | grep '^{' | cut -c1-200
python tools/exp/lde_run.py --suite bbob --dim 10 --pop 50 --route step --steps 150 2>&1 | grep '^{' | cut -c1-200
