// mbx_run_lde.hip — the instantiations of k_lde_run (one generation loop per objective kind of the noisy suite), a translation unit of their own so that
// `make -j` compiles them beside mbx.hip (which holds `extern template` declarations of them and the launch code).
#include <hip/hip_runtime.h>
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_lde.hpp"
#include "mbx_lstm_policy.hpp"
#include "mbx_lde_run.hpp"
#include "mbx_run_kernels.hpp"

namespace mbx {
template __global__ void k_lde_run<100, 30>(LdeRunArgs);
template __global__ void k_lde_run<50, 30>(LdeRunArgs);
template __global__ void k_lde_run<50, 10>(LdeRunArgs);
template __global__ void k_lde_run<50, 30, 50, true>(LdeRunArgs);        // the reference's NP on plain bbob --dim 30: second tile array, all 24 kinds        // the reference's own LDE setting (NP = 50, bbob --dim 10): all 24 kinds
}  // namespace mbx
