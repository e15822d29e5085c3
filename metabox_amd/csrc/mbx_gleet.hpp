// mbx_gleet.hpp — fused GLEET generation kernels for gfx950 (reference: src/optimizer/gleet_optimizer.py:6-314).
//
// One env step = one PSO generation whose per-particle action splits the acceleration c = 4.1 between the particle's pbest
// and the swarm's gbest, followed by the state the policy consumes: 9 features per particle (observe(), :127-152) plus the
// features the particle had when it last improved (exploration memory) and those of the gbest particle when gbest last
// improved (exploitation memory) -- [NP, 27].  Everything happens in one launch: draws, velocity / position, objective,
// pbest / gbest, stagnation counters, reward and the feature epilogue, which reads the new positions straight from LDS.
// One workgroup per instance; thread i owns particle i (NP <= 256) for all per-particle state, elements are spread over
// the block for the velocity phase and the evaluation.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, align2

namespace mbx {

struct GlLds {
    double *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *NC, *RED, *GB, *K1, *K2, *GF;
    int* IMPR;
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC}; }
};

__host__ __device__ inline int64_t gl_lds_doubles(int NP, int D)
{
    const int64_t NE = align2((int64_t)NP * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    return NE + eval_t_doubles(NP, D) + SC + 2 * DD + 5 * align2(D) + 3 * P + 16 + 10 + align2((P + 1) / 2);
}

__device__ __forceinline__ GlLds gl_carve(double* base, int NP, int D)
{
    const int64_t NE = align2((int64_t)NP * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    GlLds L;
    double* p = base;
    L.X = p; p += NE;  L.T = p; p += eval_t_doubles(NP, D);  L.Z = p; p += SC;  L.M1T = p; p += DD;  L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);  L.V0 = p; p += align2(D);  L.V1 = p; p += align2(D);  L.V2 = p; p += align2(D);  L.GB = p; p += align2(D);
    L.NC = p; p += P;  L.K1 = p; p += P;  L.K2 = p; p += P;  L.RED = p; p += 16;  L.GF = p; p += 10;
    L.IMPR = reinterpret_cast<int*>(p);
    return L;
}

// observe() for particle i (:127-152).  x: its row of L.X; pb: its pbest row (== x when it has just improved).
__device__ __forceinline__ void gl_features(const double* x, const double* pb, const double* gb, int D, double ccost, double pbest,
                                            double gbest, double max_cost, double fes, double max_fes, double pni, double no_improve,
                                            double max_step, double max_dist, double* f)
{
    f[0] = ccost / max_cost;
    f[1] = (ccost - gbest) / max_cost;
    f[2] = (ccost - pbest) / max_cost;
    f[3] = (max_fes - fes) / max_fes;
    f[4] = pni / max_step;
    f[5] = no_improve / max_step;
    double sg = 0., sp = 0., dot = 0.;
    for (int d = 0; d < D; ++d) {
        const double gv = gb[d] - x[d], pv = pb[d] - x[d];
        sg += gv * gv; sp += pv * pv; dot += pv * gv;
    }
    f[6] = sqrt(sg) / max_dist;
    f[7] = sqrt(sp) / max_dist;
    const double c8 = dot / (sqrt(sp) * sqrt(sg) + 1e-5);
    f[8] = isnan(c8) ? 0. : c8;
}

// ------------------------------------------------------------------------------------------------ reset (init_population :80-112)
__global__ __launch_bounds__(kThreads) void k_gleet_reset(BatchParams bp, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const GlLds L = gl_carve(smem, NP, D);
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_GLEET_ST_SCALARS(NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb);
    stage_problem(P, L.eval());
    for (int e = tid; e < NE; e += kThreads) {
        double up, uv;
        if (tape) { up = tape[MBX_GLEET_TAPE_POS(NP, D) + e]; uv = tape[MBX_GLEET_TAPE_VEL(NP, D) + e]; }
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_R); up = u53(w.x, w.y); uv = u53(w.z, w.w); }
        const double x = lb + (ub - lb) * up;
        L.X[e] = x;
        S[MBX_GLEET_ST_POS(NP, D) + e] = x;
        S[MBX_GLEET_ST_PBPOS(NP, D) + e] = x;
        S[MBX_GLEET_ST_VEL(NP, D) + e] = -vmax + (vmax - (-vmax)) * uv;
    }
    __syncthreads();
    population_costs(P, L.eval(), NP, rng, tape ? tape + MBX_GLEET_TAPE_NOISE_INIT(NP, D) : nullptr, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B);
    double gb; int g0;
    block_argmin(L.NC, NP, L.RED, gb, g0);
    if (tid < D) { L.GB[tid] = L.X[g0 * D + tid]; S[MBX_GLEET_ST_GBPOS(NP, D) + tid] = L.X[g0 * D + tid]; }
    __syncthreads();
    const double max_cost = gb;                                     // "max_cost" = np.min(c_cost) (:51)
    const double max_step = (double)(bp.max_fes / NP), max_dist = sqrt((ub - lb) * (ub - lb) * D);
    double f[MBX_GLEET_NFEAT];
    if (tid < NP) {
        const double c = L.NC[tid];
        S[MBX_GLEET_ST_CCOST(NP, D) + tid] = c; S[MBX_GLEET_ST_PBEST(NP, D) + tid] = c; S[MBX_GLEET_ST_PNI(NP, D) + tid] = 0.;
        const double* x = L.X + tid * D;
        gl_features(x, x, L.GB, D, c, c, gb, max_cost, (double)NP, (double)bp.max_fes, 0., 0., max_step, max_dist, f);
        for (int k = 0; k < MBX_GLEET_NFEAT; ++k) S[MBX_GLEET_ST_PFEAT(NP, D) + tid * 9 + k] = f[k];
        if (tid == g0) for (int k = 0; k < MBX_GLEET_NFEAT; ++k) { L.GF[k] = f[k]; S[MBX_GLEET_ST_GFEAT(NP, D) + k] = f[k]; }
    }
    __syncthreads();
    if (tid < NP && state_out) {
        double* so = state_out + ((int64_t)b * NP + tid) * 27;
        for (int k = 0; k < 9; ++k) { so[k] = f[k]; so[9 + k] = f[k]; so[18 + k] = L.GF[k]; }
    }
    if (tid == 0) {
        for (int k = 0; k < MBX_NSCALAR; ++k) if (k != MBX_SC_EPISODE) sc[k] = 0.;
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1; sc[MBX_SC_EPISODE] = episode;
        sc[MBX_SC_GBEST_IDX] = g0; sc[MBX_SC_GLEET_W] = 0.9; sc[MBX_SC_GLEET_MAXCOST] = max_cost; sc[MBX_SC_GLEET_NOIMPROVE] = 0;
        sc[MBX_NSCALAR] = gb;
    }
}

// ------------------------------------------------------------------------------------------------ step (update :187-314)
// 29 KB of LDS per workgroup leave room for five workgroups per CU, but left alone the compiler takes 148 VGPRs (three waves per SIMD);
// capped at four waves it needs 114 and spills nothing: 138.6 -> 122.2 us per generation of 4096 swarms (five waves: 44 spills, slower)
#ifndef MBX_GLEET_WAVES
#define MBX_GLEET_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#endif
// NPC / DC: population and dimension fixed at compile time (0 = taken from the batch), see k_rlepso_step
template <int NPC = 0, int DC = 0>
__global__ __launch_bounds__(kThreads) MBX_GLEET_WAVES void k_gleet_step(BatchParams bp, const float* __restrict__ actions, double* __restrict__ state_out,
                                                         double* __restrict__ reward_out, uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = NPC ? NPC : bp.NP, D = DC ? DC : bp.D, NE = NP * D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_GLEET_ST_SCALARS(NP, D);
    if (sc[MBX_SC_DONE] != 0.) {
        if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; }
        return;
    }
    ConstProblem& P = *(ConstProblem*)(bp.problems + bp.problem_idx[b]);   // scalar loads on demand, no SGPR-resident copy
    const GlLds L = gl_carve(smem, NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const float* act = actions + (int64_t)b * NP;
    const int gen = (int)sc[MBX_SC_GEN] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)(int)sc[MBX_SC_EPISODE], true};
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb);
    double* gPos = S + MBX_GLEET_ST_POS(NP, D);
    double* gVel = S + MBX_GLEET_ST_VEL(NP, D);
    double* gPB = S + MBX_GLEET_ST_PBPOS(NP, D);
    double gbest = sc[MBX_SC_GBEST];
    int gbest_idx = (int)sc[MBX_SC_GBEST_IDX];
    const double pre_gbest = gbest, max_cost = sc[MBX_SC_GLEET_MAXCOST];
    const double w = sc[MBX_SC_GLEET_W] - 0.5 / ((double)bp.max_fes / NP);
    double no_improve = sc[MBX_SC_GLEET_NOIMPROVE];
    const double fes = sc[MBX_SC_FES] + NP;

    stage_problem<eval_dc(DC)>(P, L.eval());
    if (tid < D) L.GB[tid] = S[MBX_GLEET_ST_GBPOS(NP, D) + tid];
    if (tid < NP) {                                                 // :200-201, 206-210: float32 products c*a and c*(1-a), then float64
        double r1, r2;
        if (tape) { r1 = tape[MBX_GLEET_TAPE_RAND1(NP, D) + tid]; r2 = tape[MBX_GLEET_TAPE_RAND2(NP, D) + tid]; }
        else { const U4 q = rng.draw((uint32_t)tid, MBX_SITE_PART); r1 = u53(q.x, q.y); r2 = u53(q.z, q.w); }
        const float a = act[tid];
        const float one_minus = 1.f - a;
        L.K1[tid] = (double)(4.1f * a) * r1;
        L.K2[tid] = (double)(4.1f * one_minus) * r2;
    }
    __syncthreads();
    const FastDiv fd(D);
    for (int e = tid; e < NE; e += kThreads) {
        const int i = fd.div(e), d = e - i * D;
        const double x = gPos[e];
        double nv = w * gVel[e] + L.K1[i] * (gPB[e] - x) + L.K2[i] * (L.GB[d] - x);
        nv = fmin(fmax(nv, -vmax), vmax);
        const double nx = fmin(fmax(x + nv, lb), ub);
        gVel[e] = nv; gPos[e] = nx; L.X[e] = nx;
    }
    __syncthreads();
    population_costs<eval_dc(DC)>(P, L.eval(), NP, rng, tape ? tape + MBX_GLEET_TAPE_NOISE(NP, D) : nullptr, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B);

    // ---- pbest / stagnation per particle (:236-266); thread i owns particle i
    double ccost = 0., pbest = 0., pni = 0.;
    int impr = 0;
    if (tid < NP) {
        ccost = L.NC[tid];
        pbest = S[MBX_GLEET_ST_PBEST(NP, D) + tid];
        impr = ccost < pbest;
        if (impr) { pbest = ccost; S[MBX_GLEET_ST_PBEST(NP, D) + tid] = ccost; }
        const double old = S[MBX_GLEET_ST_CCOST(NP, D) + tid];
        pni = ccost < old ? 0. : S[MBX_GLEET_ST_PNI(NP, D) + tid] + 1;  // against the previous CURRENT cost
        S[MBX_GLEET_ST_CCOST(NP, D) + tid] = ccost; S[MBX_GLEET_ST_PNI(NP, D) + tid] = pni;
        L.IMPR[tid] = impr;
    }
    double cbv; int cb;
    block_argmin(L.NC, NP, L.RED, cbv, cb);                          // publishes IMPR as well
    const bool better = cbv < gbest;
    if (better) { gbest = cbv; gbest_idx = cb; no_improve = 0; } else no_improve += 1;
    if (better && tid < D) { L.GB[tid] = L.X[cb * D + tid]; S[MBX_GLEET_ST_GBPOS(NP, D) + tid] = L.X[cb * D + tid]; }
    for (int e = tid; e < NE; e += kThreads) if (L.IMPR[fd.div(e)]) gPB[e] = L.X[e];
    __syncthreads();

    // ---- feature epilogue (:284-296): the new positions are still in LDS
    const double max_step = (double)(bp.max_fes / NP), max_dist = sqrt((ub - lb) * (ub - lb) * D);
    double f[MBX_GLEET_NFEAT], pf[MBX_GLEET_NFEAT];
    if (tid < NP) {
        const double* x = L.X + tid * D;
        const double* pb = impr ? x : gPB + tid * D;               // an unimproved row was last written by an earlier launch
        gl_features(x, pb, L.GB, D, ccost, pbest, gbest, max_cost, fes, (double)bp.max_fes, pni, no_improve, max_step, max_dist, f);
        double* gPF = S + MBX_GLEET_ST_PFEAT(NP, D) + tid * 9;
        for (int k = 0; k < 9; ++k) {
            if (pni == 0.) { pf[k] = f[k]; gPF[k] = f[k]; } else pf[k] = gPF[k];
        }
        if (no_improve == 0. && tid == gbest_idx) for (int k = 0; k < 9; ++k) { L.GF[k] = f[k]; S[MBX_GLEET_ST_GFEAT(NP, D) + k] = f[k]; }
    }
    if (no_improve != 0. && tid < 9) L.GF[tid] = S[MBX_GLEET_ST_GFEAT(NP, D) + tid];
    __syncthreads();
    if (tid < NP && state_out) {
        double* so = state_out + ((int64_t)b * NP + tid) * 27;
        for (int k = 0; k < 9; ++k) { so[k] = f[k]; so[9 + k] = pf[k]; so[18 + k] = L.GF[k]; }
    }
    if (tid == 0) {
        int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
        const bool done = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, sc + MBX_NSCALAR);
        const double reward = (pre_gbest - gbest) / max_cost * 100.;
        sc[MBX_SC_GBEST] = gbest; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
        sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_RETURN] += reward; sc[MBX_SC_GEN] = gen; sc[MBX_SC_GBEST_IDX] = gbest_idx;
        sc[MBX_SC_GLEET_W] = w; sc[MBX_SC_GLEET_NOIMPROVE] = no_improve;
        if (reward_out) reward_out[b] = reward;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

}  // namespace mbx
