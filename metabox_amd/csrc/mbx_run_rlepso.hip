// mbx_run_rlepso.hip — instantiations of k_rlepso_run with the exact FDR scan: NP 100 at D 10 (headline: one body per function kind), D 12 and D 30.  A translation
// unit of its own so that `make -j` compiles it beside mbx.hip (which holds `extern template` declarations and the launch code).
#include <hip/hip_runtime.h>
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_run_kernels.hpp"

namespace mbx {
MBX_RUN_RLEPSO_EXACT()
}  // namespace mbx
