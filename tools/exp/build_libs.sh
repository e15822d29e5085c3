#!/bin/bash
# build metabox_amd/csrc/libmbx.so (make, three translation units) and the phase-instrumented variant build/libmbx_phase.so (one translation unit: the phase
# counters are a __device__ array) side by side:  bash tools/exp/build_libs.sh [extra -D flags for both]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/metabox_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMBX_NOINLINE_MATH -fPIC -fvisibility=default -shared"
(make -s clean; make -s EXTRA="$*" 2>&1 | grep -E "error" ) &
if [ -z "$NO_PHASE" ]; then (/opt/rocm/bin/hipcc $FL "$@" -DMBX_PHASE_TIMING -DMBX_SINGLE_TU -o $ROOT/build/libmbx_phase.so mbx.hip 2>&1 | grep -E "error" ) & fi
wait
ls -la libmbx.so $ROOT/build/libmbx_phase.so
