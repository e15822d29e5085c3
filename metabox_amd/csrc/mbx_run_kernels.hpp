// mbx_run_kernels.hpp — the resident-rollout kernels that are compiled in translation units of their own (mbx_run_rlepso*.hip, mbx_run_lde.hip): mbx.hip sees
// explicit instantiation DECLARATIONS (no device code, the host-side launch stub is resolved at link time), the other files hold the definitions.
#pragma once
#ifndef MBX_RUN10_THREADS
#define MBX_RUN10_THREADS 256           // workgroup size of the headline resident kernel k_rlepso_run<., 100, 10, 5> (A/B knob; 128 threads at 3 waves per SIMD, no spills: 158.9 against 117.1 us per generation)
#endif
// k_rlepso_run<THREADS, NP, D, groups, TIE>: TIE = true is the exact FDR scan (default), false the MBX_F_FDR_FAST form (BASELINE configs 2 / 5 only)
#define MBX_RUN_RLEPSO_EXACT(X)                                                                          \
    X template __global__ void k_rlepso_run<MBX_RUN10_THREADS, 100, 10, 5, true>(BatchParams, const float*, int, int, RunOut); \
    X template __global__ void k_rlepso_run<512, 100, 30, 5, true>(BatchParams, const float*, int, int, RunOut);                \
    X template __global__ void k_rlepso_run<256, 100, 12, 5, true>(BatchParams, const float*, int, int, RunOut);      /* protein docking (src/config.py:86-90: dim 12), any-kind body */
#define MBX_RUN_RLEPSO_C5(X) X template __global__ void k_rlepso_run<1024, 128, 40, 5, true>(BatchParams, const float*, int, int, RunOut);
#define MBX_RUN_RLEPSO_D40(X) X template __global__ void k_rlepso_run<1024, 100, 40, 5, true>(BatchParams, const float*, int, int, RunOut);      /* the reference's NP at --dim 40 */
#define MBX_RUN_RLEPSO_FAST(X)                                                                           \
    X template __global__ void k_rlepso_run<MBX_RUN10_THREADS, 100, 10, 5, false>(BatchParams, const float*, int, int, RunOut); \
    X template __global__ void k_rlepso_run<1024, 128, 40, 5, false>(BatchParams, const float*, int, int, RunOut);
#ifdef MBX_RUN_KERNELS_EXTERN
namespace mbx {
MBX_RUN_RLEPSO_EXACT(extern)
MBX_RUN_RLEPSO_C5(extern)
MBX_RUN_RLEPSO_D40(extern)
MBX_RUN_RLEPSO_FAST(extern)
extern template __global__ void k_lde_run<100, 30>(LdeRunArgs);
extern template __global__ void k_lde_run<50, 30>(LdeRunArgs);
extern template __global__ void k_lde_run<50, 10>(LdeRunArgs);
extern template __global__ void k_lde_run<50, 30, 50, true>(LdeRunArgs);
}  // namespace mbx
#endif
