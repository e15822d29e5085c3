# headline kernel on one-function batches (D = 10, resident rollout): build/libmbx_head.so against the working tree, ns per instance-generation
for rep in 1 2; do
for lib in build/libmbx_head.so metabox_amd/csrc/libmbx.so; do
  echo "$lib $(MBX_LIB=$PWD/$lib timeout 300 python tools/kbench_costs.py --dims 10 --kinds ${KINDS:-1,10,16} --gens 40 2>&1 | tail -1)"
done; done
