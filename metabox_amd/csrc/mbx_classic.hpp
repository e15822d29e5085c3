// mbx_classic.hpp — the classic baselines of the test harness as batched kernels: differential evolution and particle
// swarm as the reference's wrappers write them out (src/optimizer/deap_de.py:8-82, deap_pso.py:8-122; deap supplies containers,
// tools.selTournament and HallOfFame only) and CMA-ES as deap.cma.Strategy runs it (src/optimizer/deap_cmaes.py:12-66; deap==1.3.3
// is not part of the reference tree, so its published update equations are restated here -- parity unpinned).
//
// None of them has an agent: mbx_reset builds the first population, every mbx_step is one sweep over the population (DE / PSO: np
// sequential single-individual updates, because both wrappers let an individual see the replacements made earlier in the same
// sweep, each billed one evaluation and checked for logging / termination) or one generation (CMA-ES).  One workgroup per
// instance; the sequential part of a step is the evaluator's single-row path, as in mbx_rlpso.hpp.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, align2, log_and_terminate

namespace mbx {
// four waves per SIMD: left alone the compiler takes 126-160 VGPRs for the multi-step / sweep kernels (three resident workgroups per CU
// although the LDS would hold five); capped at 128 they spill little or nothing (QLPSO rollout 82 -> 69 us, RL-PSO rollout 60 -> 53 us per step)
#ifndef MBX_N4_WAVES
#define MBX_N4_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#endif

struct ClLds {
    double *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *NC, *RED, *POP, *COST, *GB, *SC, *C, *B, *VEC;
    int* ORD;
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC}; }
};

// rows: evaluation rows of the launch; pop: 1 if the population is kept in LDS (DE sweeps); cma: 1 for the CMA-ES matrices
__host__ __device__ inline int64_t cl_lds_doubles(int rows, int NP, int D, int pop, int cma)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    return NE + eval_t_doubles(rows, D) + SC + 2 * DD + 5 * align2(D) + align2(rows) + 32 + MBX_NSCALAR + P + align2((P + 1) / 2) +
           (pop ? align2((int64_t)NP * D) : 0) + (cma ? 3 * DD + 8 * align2(D) + P : 0);
}

__device__ __forceinline__ ClLds cl_carve(double* base, int rows, int NP, int D, int pop, int cma)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    ClLds L;
    double* p = base;
    L.X = p; p += NE;  L.T = p; p += eval_t_doubles(rows, D);  L.Z = p; p += SC;  L.M1T = p; p += DD;  L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);  L.V0 = p; p += align2(D);  L.V1 = p; p += align2(D);  L.V2 = p; p += align2(D);  L.GB = p; p += align2(D);
    L.NC = p; p += align2(rows);  L.RED = p; p += 32;  L.SC = p; p += MBX_NSCALAR;  L.COST = p; p += P;
    L.ORD = reinterpret_cast<int*>(p); p += align2((P + 1) / 2);
    L.POP = p; p += pop ? align2((int64_t)NP * D) : 0;
    L.C = p; L.B = p + DD; L.VEC = p + 3 * DD;                        // only valid when cma (C, B, Jacobi work matrix, vectors, weights)
    return L;
}

// single-row cost of the point in L.X[0..D): objective, noise (Philox row `row`), optimum.  Thread 0 gets the value.
__device__ __forceinline__ double cl_cost1(const DevProblem& P, const ClLds& L, const Rng& rng, uint32_t row)
{
    eval_rows(P, L.eval(), 1);
    double f = L.NC[0];
    if (threadIdx.x == 0) {
        if (P.noise_kind != MBX_NOISE_NONE) {
            double a, b, c;
            philox_noise(rng, row, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B, P.noise_kind, a, b, c);
            f = apply_noise(P, f, a, b, c);
        }
        f = isnan(P.optimum) ? f : f - P.optimum;
    }
    return f;
}

// ------------------------------------------------------------------------------------------------ DE / PSO: first population
__global__ __launch_bounds__(kThreads) void k_classic_reset(BatchParams bp, int algo, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + bp.sc_off;
    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    if (algo == MBX_ALGO_CMAES) {                                   // deap_cmaes.py:32-46 + cma.Strategy.__init__: nothing is evaluated yet
        for (int k = tid; k < D * D; k += kThreads) { const double v = (k / D == k % D) ? 1. : 0.; S[MBX_CMA_ST_C(NP, D) + k] = v; S[MBX_CMA_ST_B(NP, D) + k] = v; }
        if (tid < D) {
            S[MBX_CMA_ST_CENTROID(NP, D) + tid] = P.ub; S[MBX_CMA_ST_DIAGD(NP, D) + tid] = 1.;
            S[MBX_CMA_ST_PS(NP, D) + tid] = 0.; S[MBX_CMA_ST_PC(NP, D) + tid] = 0.;
        }
        if (tid == 0) {
            for (int k = 0; k < MBX_NSCALAR; ++k) sc[k] = 0.;
            sc[MBX_SC_GBEST] = INFINITY; sc[MBX_SC_EPISODE] = episode; sc[MBX_SC_CMA_SIGMA] = 0.5;
            if (state_out) state_out[b] = 0.;
        }
        return;
    }
    const ClLds L = cl_carve(smem, NP, NP, D, 0, 0);
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub, smax = 0.5 * ub, smin = -smax;
    stage_problem(P, L.eval());
    for (int e = tid; e < NE; e += kThreads) {
        if (algo == MBX_ALGO_DE) {
            const U4 w = rng.draw((uint32_t)e, MBX_SITE_LDE_ELEM);
            const double x = lb + (ub - lb) * u53(w.x, w.y);
            L.X[e] = x; S[MBX_DE_ST_X(NP, D) + e] = x;
        } else {
            const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_R);
            const double x = lb + (ub - lb) * u53(w.x, w.y);
            L.X[e] = x; S[MBX_PSO_ST_X(NP, D) + e] = x; S[MBX_PSO_ST_PBPOS(NP, D) + e] = x;
            S[MBX_PSO_ST_SPEED(NP, D) + e] = smin + (smax - smin) * u53(w.z, w.w);
        }
    }
    __syncthreads();
    population_costs(P, L.eval(), NP, rng, nullptr, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B);
    for (int i = tid; i < NP; i += kThreads) S[(algo == MBX_ALGO_DE ? MBX_DE_ST_COST(NP, D) : MBX_PSO_ST_PBEST(NP, D)) + i] = L.NC[i];
    double gb; int g0;
    block_argmin(L.NC, NP, L.RED, gb, g0);
    if (algo == MBX_ALGO_PSO && tid < D) S[MBX_PSO_ST_GBPOS(NP, D) + tid] = L.X[g0 * D + tid];
    if (tid == 0) {
        for (int k = 0; k < MBX_NSCALAR; ++k) sc[k] = 0.;
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1; sc[MBX_SC_EPISODE] = episode;
        sc[MBX_NSCALAR] = gb;
        if (state_out) state_out[b] = (double)NP / bp.max_fes;
    }
}

// ------------------------------------------------------------------------------------------------ DE sweep (deap_de.py:48-82)
__global__ __launch_bounds__(kThreads) MBX_N4_WAVES void k_de_sweep(BatchParams bp, double* __restrict__ state_out, double* __restrict__ reward_out,
                                                       uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + bp.sc_off;
    if (sc[MBX_SC_DONE] != 0.) { if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; } return; }
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const ClLds L = cl_carve(smem, 1, NP, D, 1, 0);
    const uint64_t seed = bp.seeds[b];
    const double lb = P.lb, ub = P.ub, F = 0.5, Cr = 0.5;
    double* gX = S + MBX_DE_ST_X(NP, D);
    double* gC = S + MBX_DE_ST_COST(NP, D);
    stage_problem(P, L.eval());
    for (int e = tid; e < NE; e += kThreads) L.POP[e] = gX[e];
    for (int i = tid; i < NP; i += kThreads) L.COST[i] = gC[i];
    if (tid < MBX_NSCALAR) L.SC[tid] = sc[tid];
    __syncthreads();
    const int gen = (int)L.SC[MBX_SC_GEN] + 1;
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)(int)L.SC[MBX_SC_EPISODE]};
    int done = 0;
    for (int k = 0; k < NP && !done; ++k) {
        if (tid == 0) {                                              // tools.selTournament(pop, 3, tournsize = 3) + the forced index
            for (int j = 0; j < 3; ++j) {
                const U4 w = rng.draw((uint32_t)(3 * k + j), MBX_SITE_CLASSIC);
                int best = (int)__umulhi(w.x, (uint32_t)NP);
                const int a1 = (int)__umulhi(w.y, (uint32_t)NP), a2 = (int)__umulhi(w.z, (uint32_t)NP);
                if (L.COST[a1] < L.COST[best]) best = a1;
                if (L.COST[a2] < L.COST[best]) best = a2;
                L.RED[8 + j] = best;
                if (j == 0) L.RED[11] = (double)__umulhi(w.w, (uint32_t)D);
            }
        }
        __syncthreads();
        double y = 0.;
        if (tid < D) {
            const int a = (int)L.RED[8], bb = (int)L.RED[9], c = (int)L.RED[10], index = (int)L.RED[11];
            const U4 w = rng.draw((uint32_t)(k * D + tid), MBX_SITE_LDE_ELEM);
            if (u53(w.x, w.y) < Cr || tid == index) {
                const double v = L.POP[a * D + tid] + F * (L.POP[bb * D + tid] - L.POP[c * D + tid]);
                y = fmax(lb, fmin(v, ub));
            } else y = L.POP[k * D + tid];
            L.X[tid] = y;
        }
        __syncthreads();
        const double fy = cl_cost1(P, L, rng, (uint32_t)k);
        if (tid == 0) {
            const int better = fy < L.COST[k];
            double gbest = L.SC[MBX_SC_GBEST];
            if (better) { L.COST[k] = fy; gC[k] = fy; if (fy < gbest) gbest = fy; }
            const double fes = L.SC[MBX_SC_FES] + 1;
            int log_index = (int)L.SC[MBX_SC_LOG_INDEX], cost_len = (int)L.SC[MBX_SC_COST_LEN];
            const bool dn = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, sc + MBX_NSCALAR);
            L.SC[MBX_SC_GBEST] = gbest; L.SC[MBX_SC_FES] = fes; L.SC[MBX_SC_LOG_INDEX] = log_index; L.SC[MBX_SC_COST_LEN] = cost_len;
            L.SC[MBX_SC_DONE] = dn ? 1. : 0.;
            L.RED[12] = better;
        }
        __syncthreads();
        done = L.SC[MBX_SC_DONE] != 0.;
        if (tid < D && (int)L.RED[12]) { L.POP[k * D + tid] = y; gX[k * D + tid] = y; }
        __syncthreads();
    }
    if (tid == 0) L.SC[MBX_SC_GEN] = gen;
    __syncthreads();
    if (tid < MBX_NSCALAR) sc[tid] = L.SC[tid];
    if (tid == 0) {
        if (state_out) state_out[b] = L.SC[MBX_SC_FES] / bp.max_fes;
        if (reward_out) reward_out[b] = 0.;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ PSO sweep (deap_pso.py:31-48, 89-120)
__global__ __launch_bounds__(kThreads) MBX_N4_WAVES void k_pso_sweep(BatchParams bp, double* __restrict__ state_out, double* __restrict__ reward_out,
                                                        uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + bp.sc_off;
    if (sc[MBX_SC_DONE] != 0.) { if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; } return; }
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const ClLds L = cl_carve(smem, 1, NP, D, 0, 0);
    const uint64_t seed = bp.seeds[b];
    const double pmin = P.lb, pmax = P.ub, smax = 0.5 * pmax, smin = -smax;
    double* gX = S + MBX_PSO_ST_X(NP, D);
    double* gV = S + MBX_PSO_ST_SPEED(NP, D);
    double* gPB = S + MBX_PSO_ST_PBPOS(NP, D);
    double* gPBC = S + MBX_PSO_ST_PBEST(NP, D);
    double* gGB = S + MBX_PSO_ST_GBPOS(NP, D);
    stage_problem(P, L.eval());
    if (tid < D) L.GB[tid] = gGB[tid];
    if (tid < MBX_NSCALAR) L.SC[tid] = sc[tid];
    __syncthreads();
    const int gen = (int)L.SC[MBX_SC_GEN] + 1;
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)(int)L.SC[MBX_SC_EPISODE]};
    int done = 0;
    for (int k = 0; k < NP && !done; ++k) {
        double p = 0.;
        if (tid < D) {                                               // updateParticle: element (k, d) always belongs to thread d
            const int e = k * D + tid;
            const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_A);
            const double u1 = 0. + (2. - 0.) * u53(w.x, w.y), u2 = 0. + (2. - 0.) * u53(w.z, w.w);
            const double x = gX[e];
            double s = gV[e] + (u1 * (gPB[e] - x) + u2 * (L.GB[tid] - x));
            if (s < smin) s = smin; else if (s > smax) s = smax;
            p = x + s;
            if (p < pmin) p = pmin; else if (p > pmax) p = pmax;
            gV[e] = s; gX[e] = p; L.X[tid] = p;
        }
        __syncthreads();
        const double f = cl_cost1(P, L, rng, (uint32_t)k);
        if (tid == 0) {
            const int pb_better = f < gPBC[k];
            if (pb_better) gPBC[k] = f;
            double gbest = L.SC[MBX_SC_GBEST];
            const int gb_better = f < gbest;
            if (gb_better) gbest = f;
            const double fes = L.SC[MBX_SC_FES] + 1;
            int log_index = (int)L.SC[MBX_SC_LOG_INDEX], cost_len = (int)L.SC[MBX_SC_COST_LEN];
            const bool dn = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, sc + MBX_NSCALAR);
            L.SC[MBX_SC_GBEST] = gbest; L.SC[MBX_SC_FES] = fes; L.SC[MBX_SC_LOG_INDEX] = log_index; L.SC[MBX_SC_COST_LEN] = cost_len;
            L.SC[MBX_SC_DONE] = dn ? 1. : 0.;
            L.RED[12] = pb_better; L.RED[13] = gb_better;
        }
        __syncthreads();
        done = L.SC[MBX_SC_DONE] != 0.;
        if (tid < D) {
            if ((int)L.RED[12]) gPB[k * D + tid] = p;
            if ((int)L.RED[13]) { L.GB[tid] = p; gGB[tid] = p; }
        }
        __syncthreads();
    }
    if (tid == 0) L.SC[MBX_SC_GEN] = gen;
    __syncthreads();
    if (tid < MBX_NSCALAR) sc[tid] = L.SC[tid];
    if (tid == 0) {
        if (state_out) state_out[b] = L.SC[MBX_SC_FES] / bp.max_fes;
        if (reward_out) reward_out[b] = 0.;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

// eigh of the symmetric D x D matrix A (destroyed) by cyclic Jacobi rotations: eigenvalues ascending in w, eigenvectors in the columns
// of V.  One thread: D is 10-40 here and the operation order is then the oracle's, rotation by rotation.
__device__ void cl_jacobi_eigh(double* A, int D, double* w, double* V)
{
    for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) V[i * D + j] = i == j ? 1. : 0.;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0., diag = 0.;
        for (int p = 0; p < D; ++p) { diag += A[p * D + p] * A[p * D + p]; for (int q = p + 1; q < D; ++q) off += A[p * D + q] * A[p * D + q]; }
        if (off <= 1e-32 * diag) break;
        for (int p = 0; p < D - 1; ++p)
            for (int q = p + 1; q < D; ++q) {
                const double apq = A[p * D + q];
                if (apq == 0.) continue;
                const double theta = (A[q * D + q] - A[p * D + p]) / (2. * apq);
                const double t = (theta >= 0. ? 1. : -1.) / (fabs(theta) + sqrt(theta * theta + 1.));
                const double c = 1. / sqrt(t * t + 1.), s = t * c;
                for (int k = 0; k < D; ++k) {
                    const double akp = A[k * D + p], akq = A[k * D + q];
                    A[k * D + p] = c * akp - s * akq; A[k * D + q] = s * akp + c * akq;
                }
                for (int k = 0; k < D; ++k) {
                    const double apk = A[p * D + k], aqk = A[q * D + k];
                    A[p * D + k] = c * apk - s * aqk; A[q * D + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < D; ++k) {
                    const double vkp = V[k * D + p], vkq = V[k * D + q];
                    V[k * D + p] = c * vkp - s * vkq; V[k * D + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < D; ++i) w[i] = A[i * D + i];
    for (int i = 0; i < D - 1; ++i) {
        int m = i;
        for (int j = i + 1; j < D; ++j) if (w[j] < w[m]) m = j;
        if (m != i) {
            const double tw = w[i]; w[i] = w[m]; w[m] = tw;
            for (int k = 0; k < D; ++k) { const double tv = V[k * D + i]; V[k * D + i] = V[k * D + m]; V[k * D + m] = tv; }
        }
    }
}

// ------------------------------------------------------------------------------------------------ CMA-ES generation
// algorithms.eaGenerateUpdate(ngen = 1) with cma.Strategy(centroid = [ub] * D, sigma = 0.5, lambda_ = NP): generate, evaluate, hall
// of fame, Strategy.update (default parameters of computeParams), then the wrapper's bookkeeping (deap_cmaes.py:48-64).
__global__ __launch_bounds__(kThreads) void k_cmaes_generation(BatchParams bp, double* __restrict__ state_out, double* __restrict__ reward_out,
                                                               uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D, mu = NP / 2;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + bp.sc_off;
    if (sc[MBX_SC_DONE] != 0.) { if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; } return; }
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const ClLds L = cl_carve(smem, NP, NP, D, 0, 1);
    const int64_t DD = align2((int64_t)D * D), DV = align2(D);
    double* A = L.B + DD;                                            // Jacobi work matrix
    double* cen = L.VEC; double* dD = cen + DV; double* ps = dD + DV; double* pc = ps + DV; double* cd = pc + DV; double* t1 = cd + DV;
    double* old = t1 + DV; double* wts = old + 2 * DV;
    const uint64_t seed = bp.seeds[b];
    const int gen = (int)sc[MBX_SC_GEN] + 1;
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)(int)sc[MBX_SC_EPISODE]};
    double sigma = sc[MBX_SC_CMA_SIGMA];
    const int updates = (int)sc[MBX_SC_CMA_UPDATES];
    stage_problem(P, L.eval());
    for (int k = tid; k < D * D; k += kThreads) { L.C[k] = S[MBX_CMA_ST_C(NP, D) + k]; L.B[k] = S[MBX_CMA_ST_B(NP, D) + k]; }
    if (tid < D) {
        cen[tid] = S[MBX_CMA_ST_CENTROID(NP, D) + tid]; dD[tid] = S[MBX_CMA_ST_DIAGD(NP, D) + tid];
        ps[tid] = S[MBX_CMA_ST_PS(NP, D) + tid]; pc[tid] = S[MBX_CMA_ST_PC(NP, D) + tid];
    }
    // computeParams (defaults): weights "superlinear", mueff, cc, cs, ccov1, ccovmu, damps
    if (tid < mu) wts[tid] = m_log(mu + 0.5) - m_log((double)(tid + 1));
    __syncthreads();
    if (tid == 0) {
        double wsum = 0., w2 = 0.;
        for (int i = 0; i < mu; ++i) wsum += wts[i];
        for (int i = 0; i < mu; ++i) { wts[i] /= wsum; w2 += wts[i] * wts[i]; }
        L.RED[16] = 1. / w2;
    }
    // generate: arz ~ N(0, I) in Z, x = centroid + sigma * (arz . BD^T)
    for (int e = tid; e < NE; e += kThreads) {
        const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_A);
        double n0, n1;
        box_muller(u53(w.x, w.y), u53(w.z, w.w), n0, n1);
        L.Z[e] = n0;
    }
    __syncthreads();
    const FastDiv fd(D);
    for (int e = tid; e < NE; e += kThreads) {
        const int i = fd.div(e), d = e - i * D;
        double s = 0.;
        for (int k = 0; k < D; ++k) s += L.Z[i * D + k] * (L.B[d * D + k] * dD[k]);
        L.X[e] = cen[d] + sigma * s;
    }
    __syncthreads();
    population_costs(P, L.eval(), NP, rng, nullptr, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B);
    double cbv; int cb;
    block_argmin(L.NC, NP, L.RED, cbv, cb);
    double gbest = sc[MBX_SC_GBEST];
    if (cbv < gbest) gbest = cbv;                                    // HallOfFame(1)
    // Strategy.update: population sorted best first (stable)
    for (int i = tid; i < NP; i += kThreads) {
        const double ci = L.NC[i];
        int r = 0;
        for (int j = 0; j < NP; ++j) { const double cj = L.NC[j]; r += (cj < ci) || (cj == ci && j < i); }
        L.ORD[r] = i;
    }
    __syncthreads();
    const double mueff = L.RED[16], cc = 4. / (D + 4.), cs = (mueff + 2.) / (D + mueff + 3.);
    const double ccov1 = 2. / ((D + 1.3) * (D + 1.3) + mueff);
    double ccovmu = 2. * (mueff - 2. + 1. / mueff) / ((D + 2.) * (D + 2.) + mueff);
    if (ccovmu > 1 - ccov1) ccovmu = 1 - ccov1;
    const double damps = 1. + 2. * fmax(0., sqrt((mueff - 1.) / (D + 1.)) - 1.) + cs;
    const double chiN = sqrt((double)D) * (1. - 1. / (4. * D) + 1. / (21. * D * D));
    if (tid < D) {
        old[tid] = cen[tid];
        double s = 0.;
        for (int i = 0; i < mu; ++i) s += wts[i] * L.X[L.ORD[i] * D + tid];
        cen[tid] = s; cd[tid] = s - old[tid];
    }
    __syncthreads();
    if (tid < D) { double s = 0.; for (int d = 0; d < D; ++d) s += L.B[d * D + tid] * cd[d]; t1[tid] = s / dD[tid]; }
    __syncthreads();
    const double kps = sqrt(cs * (2. - cs) * mueff) / sigma;
    if (tid < D) { double s = 0.; for (int k = 0; k < D; ++k) s += L.B[tid * D + k] * t1[k]; ps[tid] = (1. - cs) * ps[tid] + kps * s; }
    __syncthreads();
    if (tid == 0) { double n2 = 0.; for (int d = 0; d < D; ++d) n2 += ps[d] * ps[d]; L.RED[17] = sqrt(n2); }
    __syncthreads();
    const double nps = L.RED[17];
    const double hsig = (nps / sqrt(1. - m_pow(1. - cs, 2. * (updates + 1.))) / chiN < (1.4 + 2. / (D + 1.))) ? 1. : 0.;
    const double kpc = hsig * sqrt(cc * (2. - cc) * mueff) / sigma;
    if (tid < D) pc[tid] = (1. - cc) * pc[tid] + kpc * cd[tid];
    __syncthreads();
    const double keep = 1. - ccov1 - ccovmu + (1. - hsig) * ccov1 * cc * (2. - cc), s2 = sigma * sigma;
    for (int k = tid; k < D * D; k += kThreads) {
        const int a = k / D, c2 = k - a * D;
        double r = 0.;
        for (int i = 0; i < mu; ++i) { const double* x = L.X + L.ORD[i] * D; r += wts[i] * (x[a] - old[a]) * (x[c2] - old[c2]); }
        const double v = keep * L.C[k] + ccov1 * (pc[a] * pc[c2]) + ccovmu * r / s2;
        L.C[k] = v; A[k] = v; S[MBX_CMA_ST_C(NP, D) + k] = v;
    }
    sigma *= m_exp((nps / chiN - 1.) * cs / damps);
    __syncthreads();
    if (tid == 0) {
        cl_jacobi_eigh(A, D, dD, L.B);
        for (int d = 0; d < D; ++d) dD[d] = sqrt(dD[d]);
    }
    __syncthreads();
    for (int k = tid; k < D * D; k += kThreads) S[MBX_CMA_ST_B(NP, D) + k] = L.B[k];
    if (tid < D) {
        S[MBX_CMA_ST_CENTROID(NP, D) + tid] = cen[tid]; S[MBX_CMA_ST_DIAGD(NP, D) + tid] = dD[tid];
        S[MBX_CMA_ST_PS(NP, D) + tid] = ps[tid]; S[MBX_CMA_ST_PC(NP, D) + tid] = pc[tid];
    }
    if (tid == 0) {
        const double fes = sc[MBX_SC_FES] + NP;
        int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
        const bool done = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, sc + MBX_NSCALAR);
        sc[MBX_SC_GBEST] = gbest; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
        sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_GEN] = gen; sc[MBX_SC_CMA_SIGMA] = sigma; sc[MBX_SC_CMA_UPDATES] = updates + 1;
        if (state_out) state_out[b] = fes / bp.max_fes;
        if (reward_out) reward_out[b] = 0.;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

}  // namespace mbx
