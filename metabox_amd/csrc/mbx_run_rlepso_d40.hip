// mbx_run_rlepso_d40.hip — k_rlepso_run<1024, 100, 40, 5>: RLEPSO at the reference's own population (NP = 100, rlepso_optimizer.py:11) on bbob --dim 40 (config.py:74), one body per
// function kind, exact FDR scan.  See mbx_run_rlepso.hip.
#include <hip/hip_runtime.h>
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_run_kernels.hpp"

namespace mbx {
MBX_RUN_RLEPSO_D40()
}  // namespace mbx
