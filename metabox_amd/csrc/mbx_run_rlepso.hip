// mbx_run_rlepso.hip — the instantiations of k_rlepso_run (one body per function kind for D = 10 and D = 40), a translation unit of their own so that
// `make -j` compiles them beside mbx.hip (which holds `extern template` declarations of them and the launch code).
#include <hip/hip_runtime.h>
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_run_kernels.hpp"

namespace mbx {
template __global__ void k_rlepso_run<MBX_RUN10_THREADS, 100, 10, 5>(BatchParams, const float*, int, int, RunOut);
template __global__ void k_rlepso_run<1024, 128, 40, 5>(BatchParams, const float*, int, int, RunOut);
template __global__ void k_rlepso_run<512, 100, 30, 5>(BatchParams, const float*, int, int, RunOut);
template __global__ void k_rlepso_run<256, 100, 12, 5>(BatchParams, const float*, int, int, RunOut);      // protein docking (src/config.py:86-90: dim 12), any-kind body
}  // namespace mbx
