# config 5 (D = 40, NP = 128) same-box A/B: MBX_LIBS="a.so b.so" bash tools/exp/c5_ab.sh
for rep in 1 2; do for lib in ${MBX_LIBS:-build/libmbx_head.so metabox_amd/csrc/libmbx.so}; do MBX_LIB=$PWD/$lib timeout 300 python tools/exp/c5_time.py 2>&1 | tail -1; done; done
