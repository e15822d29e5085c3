// mbx_rs.hpp — batched Random_search (reference: src/optimizer/random_search.py:24-58): every step draws NP uniform
// samples in [lb, ub]^D, evaluates them and keeps the best-so-far.  It is the normaliser of the AEI metric
// (src/logger.py:94-120) and a ten-line consumer of the block-cooperative evaluator.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"

namespace mbx {

__host__ __device__ inline int64_t rs_lds_doubles(int NP, int D)
{
    const int64_t NE = align2((int64_t)NP * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D);
    return NE + eval_t_doubles(NP, D) + SC + 2 * DD + 4 * align2(D) + align2(NP) + 16;
}

// `first` = 1: the initial population of run_episode's __reset (:24-29); 0: one more population (:45-46)
__global__ __launch_bounds__(kThreads) void k_rs_population(BatchParams bp, int first, double* __restrict__ state_out,
                                                            double* __restrict__ reward_out, uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    double* sc = bp.state + (int64_t)b * bp.state_stride;
    if (!first && sc[MBX_SC_DONE] != 0.) {
        if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; }
        return;
    }
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const int64_t NEa = align2((int64_t)NE), SC = align2(NEa > 2 * kThreads ? NEa : 2 * kThreads), DD = align2((int64_t)D * D), DV = align2(D);
    double* X = smem; double* T = X + NEa; double* Z = T + eval_t_doubles(NP, D); double* M1T = Z + SC; double* M2T = M1T + DD; double* VEC = M2T + DD;
    double* NC = VEC + 4 * DV; double* RED = NC + align2(NP);
    const EvalLds L{X, Z, T, M1T, M2T, VEC, VEC + DV, VEC + 2 * DV, VEC + 3 * DV, NC};
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int episode = first ? (int)sc[MBX_SC_EPISODE] + 1 : (int)sc[MBX_SC_EPISODE];
    const int gen = first ? 0 : (int)sc[MBX_SC_GEN] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)episode};
    stage_problem(P, L);
    for (int e = tid; e < NE; e += kThreads) {
        double u;
        if (tape) u = tape[MBX_RS_TAPE_POS(NP, D) + e];
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_LDE_ELEM); u = u53(w.x, w.y); }
        X[e] = P.lb + (P.ub - P.lb) * u;
    }
    __syncthreads();
    eval_rows(P, L, NP);
    for (int i = tid; i < NP; i += kThreads) {
        double f = NC[i];
        if (P.noise_kind != MBX_NOISE_NONE) {
            double a, bb, c;
            if (tape) { const double* t = tape + MBX_RS_TAPE_NOISE(NP, D); a = t[i]; bb = t[NP + i]; c = t[2 * NP + i]; }
            else philox_noise(rng, (uint32_t)i, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B, P.noise_kind, a, bb, c);
            f = apply_noise(P, f, a, bb, c);
        }
        NC[i] = isnan(P.optimum) ? f : f - P.optimum;
    }
    __syncthreads();
    double m; int mi;
    block_argmin(NC, NP, RED, m, mi);
    if (tid == 0) {
        double* cost = sc + MBX_NSCALAR;
        if (first) {
            sc[MBX_SC_GBEST] = m; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1; sc[MBX_SC_DONE] = 0;
            sc[MBX_SC_RETURN] = 0; sc[MBX_SC_GEN] = 0; sc[MBX_SC_EPISODE] = episode; cost[0] = m;
            if (state_out) state_out[b] = (double)NP / bp.max_fes;
        } else {
            double gbest = sc[MBX_SC_GBEST];
            const double fes = sc[MBX_SC_FES] + NP;
            if (gbest > m) gbest = m;
            int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
            const bool done = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, cost);
            sc[MBX_SC_GBEST] = gbest; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
            sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_GEN] = gen;
            if (state_out) state_out[b] = fes / bp.max_fes;
            if (reward_out) reward_out[b] = 0.;
            if (done_out) done_out[b] = done ? 1 : 0;
        }
    }
}

}  // namespace mbx
