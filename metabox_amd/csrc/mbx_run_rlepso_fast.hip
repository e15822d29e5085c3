// mbx_run_rlepso_fast.hip — the MBX_F_FDR_FAST forms of the resident kernels of BASELINE configs 2 and 5 (cross-multiplied FDR scan without the near-tie
// flag and second pass).  See mbx_run_rlepso.hip.
#include <hip/hip_runtime.h>
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_run_kernels.hpp"

namespace mbx {
MBX_RUN_RLEPSO_FAST()
}  // namespace mbx
