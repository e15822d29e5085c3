// mbx_qlpso.hpp — QLPSO step kernels for gfx950 (reference: src/optimizer/qlpso_optimizer.py:7-125 and the tabular policy of
// src/agent/qlpso_agent.py:35-38).
//
// One env step moves ONE particle of a 30-particle swarm: ring-neighbourhood best (size 4 / 8 / 16 / 30 = the action),
// velocity, clipping, one evaluation, the swarm diversity mean_i ||x_i - mean(x)||, and a reward in {2, 1, 0, -2} from
// (cost improved?, diversity grew?).  The state is the action the next particle took on its previous turn.  Like RL-PSO this
// is latency, not work, so the step kernel also runs with the policy inside -- softmax over a row of the 4 x 4 Q-table and
// numpy's choice rule -- and `n_steps` steps per launch (mbx_qlpso_rollout).
// The diversity decides the reward through `d_new > d_old`; its sums follow numpy's pairwise order (np.sum(., 1), np.mean(.)).
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, align2

namespace mbx {
// four waves per SIMD: left alone the compiler takes 126-160 VGPRs for the multi-step / sweep kernels (three resident workgroups per CU
// although the LDS would hold five); capped at 128 they spill little or nothing (QLPSO rollout 82 -> 69 us, RL-PSO rollout 60 -> 53 us per step)
#ifndef MBX_N4_WAVES
#define MBX_N4_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#endif

// np.add.reduce over n <= 128 contiguous values produced by elem(k): 8 accumulators, then the tail (numpy's pairwise_sum).
template <class F>
__device__ __forceinline__ double np_sum_block(F elem, int n)
{
    if (n < 8) { double s = 0.; for (int k = 0; k < n; ++k) s += elem(k); return s; }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = elem(k);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] += elem(i + k);
    }
    double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) s += elem(i);
    return s;
}

// ... and for n <= 256 (one halving step above 128 elements)
template <class F>
__device__ __forceinline__ double np_sum(F elem, int n)
{
    if (n <= 128) return np_sum_block(elem, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum_block(elem, n2) + np_sum_block([&](int k) { return elem(n2 + k); }, n - n2);
}

struct QlLds {
    double *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *NC, *RED, *POP, *COST, *MEAN, *DIST, *SC;
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC}; }
};

// rows = evaluation rows of the launch: NP for the reset (its X doubles as the population), 1 for a step
__host__ __device__ inline int64_t ql_lds_doubles(int rows, int NP, int D)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    return NE + eval_t_doubles(rows, D) + SC + 2 * DD + 5 * align2(D) + align2(rows) + 16 + align2((int64_t)NP * D) + 2 * P + MBX_NSCALAR;
}

__device__ __forceinline__ QlLds ql_carve(double* base, int rows, int NP, int D)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    QlLds L;
    double* p = base;
    L.X = p; p += NE;  L.T = p; p += eval_t_doubles(rows, D);  L.Z = p; p += SC;  L.M1T = p; p += DD;  L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);  L.V0 = p; p += align2(D);  L.V1 = p; p += align2(D);  L.V2 = p; p += align2(D);  L.MEAN = p; p += align2(D);
    L.NC = p; p += align2(rows);  L.RED = p; p += 16;  L.POP = p; p += align2((int64_t)NP * D);  L.COST = p; p += P;  L.DIST = p; p += P;
    L.SC = p;
    return L;
}

// __cal_diversity (:45-46) of the population in L.POP.  All threads call; the value is returned to every thread.
__device__ __forceinline__ double ql_diversity(const QlLds& L, int NP, int D)
{
    const int tid = threadIdx.x;
    for (int d = tid; d < D; d += kThreads) {                       // np.mean(pop, 0): row after row
        double s = 0.;
        for (int i = 0; i < NP; ++i) s += L.POP[i * D + d];
        L.MEAN[d] = s / NP;
    }
    __syncthreads();
    for (int i = tid; i < NP; i += kThreads) {
        const double* x = L.POP + i * D;
        L.DIST[i] = sqrt(np_sum([&](int d) { const double t = x[d] - L.MEAN[d]; return t * t; }, D));
    }
    __syncthreads();
    if (tid == 0) L.RED[0] = np_sum([&](int i) { return L.DIST[i]; }, NP) / NP;
    __syncthreads();
    return L.RED[0];
}

// ------------------------------------------------------------------------------------------------ reset (init_population :76-90)
__global__ __launch_bounds__(kThreads) void k_qlpso_reset(BatchParams bp, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const QlLds L = ql_carve(smem, NP, NP, D);
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_QLPSO_ST_SCALARS(NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    const int pointer = (int)sc[MBX_SC_QLPSO_POINTER];             // survives the reset, as in the reference (:33)
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub;
    stage_problem(P, L.eval());
    for (int e = tid; e < NE; e += kThreads) {
        double u;
        if (tape) u = tape[MBX_QLPSO_TAPE_POS(NP, D) + e];
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_LDE_ELEM); u = u53(w.x, w.y); }
        const double x = u * (ub - lb) + lb;
        L.X[e] = x; L.POP[e] = x;
        S[MBX_QLPSO_ST_POP(NP, D) + e] = x; S[MBX_QLPSO_ST_PBPOS(NP, D) + e] = x; S[MBX_QLPSO_ST_VEL(NP, D) + e] = 0.;
    }
    __syncthreads();
    const double diversity = ql_diversity(L, NP, D);
    eval_rows(P, L.eval(), NP);
    for (int i = tid; i < NP; i += kThreads) {
        double f = L.NC[i];
        if (P.noise_kind != MBX_NOISE_NONE) {
            double a, bb, c;
            if (tape) { const double* t = tape + MBX_QLPSO_TAPE_NOISE_INIT(NP, D); a = t[i]; bb = t[NP + i]; c = t[2 * NP + i]; }
            else philox_noise(rng, (uint32_t)i, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B, P.noise_kind, a, bb, c);
            f = apply_noise(P, f, a, bb, c);
        }
        f = isnan(P.optimum) ? f : f - P.optimum;
        L.COST[i] = f; S[MBX_QLPSO_ST_COST(NP, D) + i] = f;
        double s0;
        if (tape) s0 = tape[MBX_QLPSO_TAPE_SSTATE(NP, D) + i];
        else { const U4 w = rng.draw((uint32_t)i, MBX_SITE_PART); s0 = (double)__umulhi(w.x, 4u); }
        S[MBX_QLPSO_ST_SSTATE(NP, D) + i] = s0;
        if (i == pointer && state_out) state_out[b] = s0;
    }
    __syncthreads();
    double gb; int g0;
    block_argmin(L.COST, NP, L.RED, gb, g0);
    if (tid == 0) {
        for (int k = 0; k < MBX_NSCALAR; ++k) if (k != MBX_SC_EPISODE && k != MBX_SC_QLPSO_POINTER) sc[k] = 0.;
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1; sc[MBX_SC_EPISODE] = episode;
        sc[MBX_SC_QLPSO_DIVERSITY] = diversity;
        sc[MBX_NSCALAR] = gb;
    }
}

// QLPSO_Agent.__get_action (qlpso_agent.py:35-38): p = softmax(Q[state]); np.random.choice(4, p = p) with one uniform u:
// index = searchsorted(cumsum(p) / cumsum(p)[-1], u, side = 'right').
__device__ __forceinline__ int ql_choose(const double* __restrict__ q_row, double u)
{
    double e[4], s = 0., cdf[4], c = 0.;
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = m_exp(q_row[k]); s += e[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { c += e[k] / s; cdf[k] = c; }
    int idx = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) idx += (cdf[k] / cdf[3]) <= u;
    return idx;
}

// ------------------------------------------------------------------------------------------------ step (update :92-125)
// MULTI = false: exactly one step (mbx_step, or a one-step rollout); MULTI = true: the n_steps loop (see mbx_rlpso.hpp for why the
// two are separate instantiations).
template <bool MULTI>
__global__ __launch_bounds__(kThreads) MBX_N4_WAVES void k_qlpso_step(BatchParams bp, const int32_t* __restrict__ actions, const double* __restrict__ q_table,
                                                         int n_steps, double* __restrict__ state_out, double* __restrict__ reward_out,
                                                         uint8_t* __restrict__ done_out, int32_t* __restrict__ actions_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_QLPSO_ST_SCALARS(NP, D);
    if (sc[MBX_SC_DONE] != 0.) {
        if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; }
        return;
    }
    ConstProblem& P = *(ConstProblem*)(bp.problems + bp.problem_idx[b]);   // scalar loads on demand, no SGPR-resident copy
    const QlLds L = ql_carve(smem, 1, NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const uint64_t seed = bp.seeds[b];
    const double lb = P.lb, ub = P.ub, W = 0.729844, C = 1.49618;
    double* gPop = S + MBX_QLPSO_ST_POP(NP, D);
    double* gVel = S + MBX_QLPSO_ST_VEL(NP, D);
    double* gPB = S + MBX_QLPSO_ST_PBPOS(NP, D);
    double* gCost = S + MBX_QLPSO_ST_COST(NP, D);
    double* gSS = S + MBX_QLPSO_ST_SSTATE(NP, D);

    stage_problem(P, L.eval());
    for (int e = tid; e < NE; e += kThreads) L.POP[e] = gPop[e];
    for (int i = tid; i < NP; i += kThreads) L.COST[i] = gCost[i];
    if (tid < MBX_NSCALAR) L.SC[tid] = sc[tid];
    __syncthreads();
    const int episode = (int)L.SC[MBX_SC_EPISODE];
    double reward_sum = 0.;
    int done = 0;
    for (int it = 0; it < (MULTI ? n_steps : 1) && !done; ++it) {
        const int step = (int)L.SC[MBX_SC_GEN] + 1, i = (int)L.SC[MBX_SC_QLPSO_POINTER];
        const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)episode, true};
        if (tid == 0) {                                              // decision, neighbourhood best, the two uniforms
            int action;
            if (q_table) {
                double u;
                if (tape) u = tape[MBX_QLPSO_TAPE_CHOICE(NP, D)];
                else { const U4 w = rng.draw(0u, MBX_SITE_POLICY); u = u53(w.x, w.y); }
                action = ql_choose(q_table + 4 * (int)gSS[i], u);
                if (actions_out) actions_out[b] = action;
            } else action = actions[b];
            const int k = action == 0 ? 4 : action == 1 ? 8 : action == 2 ? 16 : action == 3 ? 30 : 0;
            int nb = i; double nbc = INFINITY;
            for (int j = -(k / 2); j <= k / 2; ++j) {                 // range(-k//2, k//2 + 1), first strict minimum (:62-67)
                const int idx = (((i + j) % NP) + NP) % NP;
                if (L.COST[idx] < nbc) { nbc = L.COST[idx]; nb = idx; }
            }
            double ra, rb;
            if (tape) { ra = tape[MBX_QLPSO_TAPE_RAND(NP, D)]; rb = tape[MBX_QLPSO_TAPE_RAND(NP, D) + 1]; }
            else { const U4 w = rng.draw(0u, MBX_SITE_PART); ra = u53(w.x, w.y); rb = u53(w.z, w.w); }
            L.RED[8] = action; L.RED[9] = nb; L.RED[10] = C * ra; L.RED[11] = C * rb;
        }
        __syncthreads();
        const int action = (int)L.RED[8], nb = (int)L.RED[9];
        const double ca = L.RED[10], cb = L.RED[11];
        double nx = 0.;
        if (tid < D) {
            const int e = i * D + tid;
            const double x = L.POP[e];
            const double nv = W * gVel[e] + ca * (L.POP[nb * D + tid] - x) + cb * (gPB[e] - x);   // nbest may be the particle itself
            nx = fmin(fmax(x + nv, lb), ub);
            gVel[e] = nv; gPop[e] = nx;
            L.X[tid] = nx;
        }
        __syncthreads();
        if (tid < D) L.POP[i * D + tid] = nx;                        // after every lane has read its nbest coordinate
        eval_rows(P, L.eval(), 1);
        __syncthreads();
        const double d_new = ql_diversity(L, NP, D);
        if (tid == 0) {
            double f_new = L.NC[0];
            if (P.noise_kind != MBX_NOISE_NONE) {
                double a, bb, cc;
                if (tape) { a = tape[MBX_QLPSO_TAPE_NOISE(NP, D)]; bb = tape[MBX_QLPSO_TAPE_NOISE(NP, D) + 1]; cc = tape[MBX_QLPSO_TAPE_NOISE(NP, D) + 2]; }
                else philox_noise(rng, 0u, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B, P.noise_kind, a, bb, cc);
                f_new = apply_noise(P, f_new, a, bb, cc);
            }
            f_new = isnan(P.optimum) ? f_new : f_new - P.optimum;
            const double f_old = L.COST[i], d_old = L.SC[MBX_SC_QLPSO_DIVERSITY];
            const int better = f_new < f_old;
            const double reward = better ? (d_new > d_old ? 2. : 1.) : (d_new > d_old ? 0. : -2.);   // cal_reward :7-16
            L.COST[i] = f_new; gCost[i] = f_new;
            double gbest = L.SC[MBX_SC_GBEST];
            for (int q = 0; q < NP; ++q) gbest = L.COST[q] < gbest ? L.COST[q] : gbest;
            gSS[i] = action;
            const int pointer = (i + 1) % NP;
            const double fes = L.SC[MBX_SC_FES] + 1;
            int log_index = (int)L.SC[MBX_SC_LOG_INDEX], cost_len = (int)L.SC[MBX_SC_COST_LEN];
            double* cost = sc + MBX_NSCALAR;
            const bool dn = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, cost);
            L.SC[MBX_SC_GBEST] = gbest; L.SC[MBX_SC_FES] = fes; L.SC[MBX_SC_LOG_INDEX] = log_index; L.SC[MBX_SC_COST_LEN] = cost_len;
            L.SC[MBX_SC_DONE] = dn ? 1. : 0.; L.SC[MBX_SC_RETURN] += reward; L.SC[MBX_SC_GEN] = step;
            L.SC[MBX_SC_QLPSO_DIVERSITY] = d_new; L.SC[MBX_SC_QLPSO_POINTER] = pointer;
            L.RED[12] = better; L.RED[13] = reward;
        }
        __syncthreads();
        reward_sum += L.RED[13];
        done = L.SC[MBX_SC_DONE] != 0.;
        if (tid < D && (int)L.RED[12]) gPB[i * D + tid] = nx;          // "pbest" follows the previous CURRENT cost (:112-113)
        __syncthreads();
    }
    if (tid < MBX_NSCALAR) sc[tid] = L.SC[tid];
    if (tid == 0) {
        if (state_out) state_out[b] = gSS[(int)L.SC[MBX_SC_QLPSO_POINTER]];
        if (reward_out) reward_out[b] = reward_sum;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

}  // namespace mbx
