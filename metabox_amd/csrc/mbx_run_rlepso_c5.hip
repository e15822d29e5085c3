// mbx_run_rlepso_c5.hip — k_rlepso_run<1024, 128, 40, 5> (BASELINE config 5; one body per function kind), exact FDR scan.  See mbx_run_rlepso.hip.
#include <hip/hip_runtime.h>
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_run_kernels.hpp"

namespace mbx {
MBX_RUN_RLEPSO_C5()
}  // namespace mbx
