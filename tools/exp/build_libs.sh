#!/bin/bash
# build metabox_amd/csrc/libmbx.so and the phase-instrumented variant build/libmbx_phase.so side by side:  bash tools/exp/build_libs.sh [extra -D flags for both]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd $ROOT/metabox_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMBX_NOINLINE_MATH -fPIC -fvisibility=default -shared"
(/opt/rocm/bin/hipcc $FL "$@" -o libmbx.so mbx.hip 2>&1 | grep -E "error" ) &
(/opt/rocm/bin/hipcc $FL "$@" -DMBX_PHASE_TIMING -o $ROOT/build/libmbx_phase.so mbx.hip 2>&1 | grep -E "error" ) &
wait
ls -la libmbx.so $ROOT/build/libmbx_phase.so
