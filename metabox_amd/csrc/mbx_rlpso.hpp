// mbx_rlpso.hpp — RL-PSO step kernels for gfx950 (reference: src/optimizer/rl_pso_optimizer.py:7-148).
//
// One env step moves ONE particle (round robin): velocity / position update whose gbest-attraction weight is the agent's
// action, one objective evaluation, pbest / gbest, reward.  That is D elements of arithmetic and one evaluation per
// launch -- pure launch latency if every step is a launch and the policy a second set of launches.  k_rlpso_step
// therefore has two modes:
//   * actions given  (mbx_step):          one step per launch, the plugin route (agent.act outside);
//   * policy given   (mbx_rlpso_rollout): the 2D -> h1 -> h2 -> 1 actor is evaluated inside the kernel and `n_steps`
//                                         consecutive steps run in ONE launch; nothing returns to the host in between.
// One workgroup per instance.  The particle table stays in HBM/L2 (a step touches one row; element (i, d) is always
// read and written by thread d, scalars by thread 0, so no cross-thread traffic goes through global memory); the
// gbest position, the next particle's position, the actor's weights and activations and the problem constants live in LDS.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, align2, GaussMlp, sample_normal

namespace mbx {
// four waves per SIMD: left alone the compiler takes 126-160 VGPRs for the multi-step / sweep kernels (three resident workgroups per CU
// although the LDS would hold five); capped at 128 they spill little or nothing (QLPSO rollout 82 -> 69 us, RL-PSO rollout 60 -> 53 us per step)
#ifndef MBX_N4_WAVES
#define MBX_N4_WAVES __attribute__((amdgpu_waves_per_eu(4)))
#endif

struct RpLds {
    double *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *NC, *RED, *GB, *XC, *SC;
    float* ACTV;            // policy activations: input [2D] | h1 [2 h1] | h2 [2 h2] | out [2]
    float* WTS;             // packed actor weights (mbx_rlpso_rollout only)
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC}; }
};

constexpr int kRpActDoubles = 192;       // room for 2D + 2 h1 + 2 h2 + 2 floats (checked on the host)

// `rows` = evaluation rows the launch needs: NP for the reset, 1 for a step (17 KB instead of 27 KB at D = 10: the step kernel is a
// chain of short latency-bound phases, so the number of resident workgroups per CU is what buys throughput).
__host__ __device__ inline int64_t rp_lds_doubles(int rows, int D, int weight_floats)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(rows);
    return NE + eval_t_doubles(rows, D) + SC + 2 * DD + 6 * align2(D) + P + 16 + MBX_NSCALAR + kRpActDoubles + align2((weight_floats + 1) / 2);
}

__device__ __forceinline__ RpLds rp_carve(double* base, int rows, int D)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(rows);
    RpLds L;
    double* p = base;
    L.X = p; p += NE;  L.T = p; p += eval_t_doubles(rows, D);  L.Z = p; p += SC;  L.M1T = p; p += DD;  L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);  L.V0 = p; p += align2(D);  L.V1 = p; p += align2(D);  L.V2 = p; p += align2(D);
    L.GB = p; p += align2(D);  L.XC = p; p += align2(D);
    L.NC = p; p += P;  L.RED = p; p += 16;  L.SC = p; p += MBX_NSCALAR;  L.ACTV = reinterpret_cast<float*>(p); p += kRpActDoubles;
    L.WTS = reinterpret_cast<float*>(p);
    return L;
}

// ------------------------------------------------------------------------------------------------ reset (init_population :30-60)
__global__ __launch_bounds__(kThreads) void k_rlpso_reset(BatchParams bp, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const RpLds L = rp_carve(smem, NP, D);
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_RLPSO_ST_SCALARS(NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb);
    stage_problem(P, L.eval());
    for (int e = tid; e < NE; e += kThreads) {
        double up, uv;
        if (tape) { up = tape[MBX_RLPSO_TAPE_POS(NP, D) + e]; uv = tape[MBX_RLPSO_TAPE_VEL(NP, D) + e]; }
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_R); up = u53(w.x, w.y); uv = u53(w.z, w.w); }
        const double x = lb + (ub - lb) * up;
        L.X[e] = x;
        S[MBX_RLPSO_ST_POS(NP, D) + e] = x;
        S[MBX_RLPSO_ST_PBPOS(NP, D) + e] = x;
        S[MBX_RLPSO_ST_VEL(NP, D) + e] = -vmax + (vmax - (-vmax)) * uv;
    }
    __syncthreads();
    eval_rows(P, L.eval(), NP);
    double* NEG = L.Z;                                              // free after the evaluation
    for (int i = tid; i < NP; i += kThreads) {
        double f = L.NC[i];
        if (P.noise_kind != MBX_NOISE_NONE) {
            double a, bb, c;
            if (tape) { const double* t = tape + MBX_RLPSO_TAPE_NOISE_INIT(NP, D); a = t[i]; bb = t[NP + i]; c = t[2 * NP + i]; }
            else philox_noise(rng, (uint32_t)i, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B, P.noise_kind, a, bb, c);
            f = apply_noise(P, f, a, bb, c);
        }
        f = isnan(P.optimum) ? f : f - P.optimum;
        L.NC[i] = f; NEG[i] = -f;
        S[MBX_RLPSO_ST_CCOST(NP, D) + i] = f;
        S[MBX_RLPSO_ST_PBEST(NP, D) + i] = f;
    }
    __syncthreads();
    double gb, negmax; int g0, iw;
    block_argmin(L.NC, NP, L.RED, gb, g0);
    block_argmin(NEG, NP, L.RED, negmax, iw);
    if (tid < D) {
        S[MBX_RLPSO_ST_GBPOS(NP, D) + tid] = L.X[g0 * D + tid];
        if (state_out) { state_out[(int64_t)b * 2 * D + tid] = L.X[g0 * D + tid]; state_out[(int64_t)b * 2 * D + D + tid] = L.X[tid]; }
    }
    if (tid == 0) {
        for (int k = 0; k < MBX_NSCALAR; ++k) if (k != MBX_SC_EPISODE) sc[k] = 0.;
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1; sc[MBX_SC_EPISODE] = episode;
        sc[MBX_SC_GBEST_IDX] = g0; sc[MBX_SC_RLPSO_W] = 0.9; sc[MBX_SC_RLPSO_MAXCOST] = -negmax; sc[MBX_SC_RLPSO_CUR] = 0;
        sc[MBX_NSCALAR] = gb;
    }
}

// PolicyNetwork.forward (src/agent/rl_pso_agent.py:9-47) for ONE state held in LDS as float32: mu = (tanh + 1)/2,
// sigma = clamp((tanh + 1)/2, min, max), action = Normal(mu, sigma) sample, re-folded by the reference's rule when it leaves
// [0, 1).  Same accumulation order as k_gauss_mlp_policy (bias first, inputs in ascending order).  All threads call.
__device__ __forceinline__ float rp_policy(const GaussMlp& net, const RpLds& L, int D, const Rng& rng)
{
    const int tid = threadIdx.x, IN = 2 * D, H1 = net.h1, H2 = net.h2;
    const int NW = gauss_mlp_net_floats(IN, H1, H2, 1);
    const int o_b1 = IN * H1, o_w2 = o_b1 + H1, o_b2 = o_w2 + H1 * H2, o_w3 = o_b2 + H2, o_b3 = o_w3 + H2;
    float* sin_ = L.ACTV; float* h1v = sin_ + IN; float* h2v = h1v + 2 * H1; float* outv = h2v + 2 * H2;
    if (tid < D) { sin_[tid] = (float)L.GB[tid]; sin_[D + tid] = (float)L.XC[tid]; }
    __syncthreads();
    for (int j = tid; j < 2 * H1; j += kThreads) {
        const int n = j >= H1, o = j - n * H1;
        const float* W = L.WTS + n * NW;
        float acc = W[o_b1 + o];
#pragma unroll 2
        for (int k = 0; k < IN; ++k) acc += sin_[k] * W[k * H1 + o];
        h1v[j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    for (int j = tid; j < 2 * H2; j += kThreads) {
        const int n = j >= H2, o = j - n * H2;
        const float* W = L.WTS + n * NW;
        float acc = W[o_b2 + o];
#pragma unroll 2
        for (int k = 0; k < H1; ++k) acc += h1v[n * H1 + k] * W[o_w2 + k * H2 + o];
        h2v[j] = fmaxf(acc, 0.f);
    }
    __syncthreads();
    if (tid < 2) {
        const float* W = L.WTS + tid * NW;
        float acc = W[o_b3];
#pragma unroll 2
        for (int k = 0; k < H2; ++k) acc += h2v[tid * H2 + k] * W[o_w3 + k];
        outv[tid] = acc;
    }
    __syncthreads();
    float mu, sigma;
    gauss_head(net, outv[0], outv[1], mu, sigma);
    return sample_action(rng, 0, mu, sigma, net.variant);
}

// ------------------------------------------------------------------------------------------------ step (update :76-148)
// MULTI = false: exactly one step, no loop -- 82 VGPRs; MULTI = true: the n_steps loop, in which the evaluator's loop-invariant
// address arithmetic gets hoisted and stays live across iterations (141 VGPRs, 3 waves per SIMD instead of 5).
template <bool MULTI>
__global__ __launch_bounds__(kThreads) MBX_N4_WAVES void k_rlpso_step(BatchParams bp, const float* __restrict__ actions, GaussMlp net, int n_steps,
                                                         double* __restrict__ state_out, double* __restrict__ reward_out,
                                                         uint8_t* __restrict__ done_out, float* __restrict__ actions_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_RLPSO_ST_SCALARS(NP, D);
    if (sc[MBX_SC_DONE] != 0.) {
        if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; }
        return;
    }
    ConstProblem& P = *(ConstProblem*)(bp.problems + bp.problem_idx[b]);   // scalar loads on demand, no SGPR-resident copy
    const RpLds L = rp_carve(smem, 1, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const uint64_t seed = bp.seeds[b];
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb), c = 2.05;
    if (net.w) {
        const int nw = 2 * gauss_mlp_net_floats(2 * D, net.h1, net.h2, 1);
        for (int k = tid; k < nw; k += kThreads) L.WTS[k] = net.w[k];
    }
    double* gPos = S + MBX_RLPSO_ST_POS(NP, D);
    double* gVel = S + MBX_RLPSO_ST_VEL(NP, D);
    double* gPB = S + MBX_RLPSO_ST_PBPOS(NP, D);
    double* gCC = S + MBX_RLPSO_ST_CCOST(NP, D);
    double* gPBC = S + MBX_RLPSO_ST_PBEST(NP, D);
    double* gGB = S + MBX_RLPSO_ST_GBPOS(NP, D);

    stage_problem(P, L.eval());
    if (tid < MBX_NSCALAR) L.SC[tid] = sc[tid];
    if (tid < D) { L.GB[tid] = gGB[tid]; L.XC[tid] = gPos[(int)sc[MBX_SC_RLPSO_CUR] * D + tid]; }
    __syncthreads();
    const int episode = (int)L.SC[MBX_SC_EPISODE];
    double reward_sum = 0.;
    int done = 0;
    MBX_PHASE_BEGIN
    for (int it = 0; it < (MULTI ? n_steps : 1) && !done; ++it) {
        const int step = (int)L.SC[MBX_SC_GEN] + 1, j = (int)L.SC[MBX_SC_RLPSO_CUR];
        const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)step, (uint32_t)episode, true};
        float action;
        if (net.w) {
            action = rp_policy(net, L, D, rng);
            if (actions_out && tid == 0) actions_out[b] = action;
        } else action = actions[b];
        MBX_PHASE(0);                                                // actor
        const double w = L.SC[MBX_SC_RLPSO_W] - 0.5 / ((double)bp.max_fes / NP);          // every call, not every generation
        double rand1;
        if (tape) rand1 = tape[MBX_RLPSO_TAPE_RAND1(NP, D)];
        else { const U4 r = rng.draw(0u, MBX_SITE_PART); rand1 = u53(r.x, r.y); }
        const double c1 = c * rand1;
        const double c2 = (double)(2.05f * action);                  // float32 product, as numpy >= 2 computes c * float32 action
        double nx = 0.;
        if (tid < D) {
            const int e = j * D + tid;
            const double x = L.XC[tid], v = gVel[e];
            double nv = w * v + c1 * (gPB[e] - x) + c2 * (L.GB[tid] - x);
            nv = fmin(fmax(nv, -vmax), vmax);
            nx = x + nv;
            nx = fmin(fmax(nx, lb), ub);
            gVel[e] = nv; gPos[e] = nx;
            L.X[tid] = nx;
        }
        __syncthreads();
        MBX_PHASE(1);                                                // move
        eval_rows(P, L.eval(), 1);
        MBX_PHASE(2);                                                // evaluation
        if (tid == 0) {
            double nc = L.NC[0];
            if (P.noise_kind != MBX_NOISE_NONE) {
                double a, bb, cc;
                if (tape) { a = tape[MBX_RLPSO_TAPE_NOISE(NP, D)]; bb = tape[MBX_RLPSO_TAPE_NOISE(NP, D) + 1]; cc = tape[MBX_RLPSO_TAPE_NOISE(NP, D) + 2]; }
                else philox_noise(rng, 0u, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B, P.noise_kind, a, bb, cc);
                nc = apply_noise(P, nc, a, bb, cc);
            }
            nc = isnan(P.optimum) ? nc : nc - P.optimum;
            const double pre_cost = gCC[j];
            gCC[j] = nc;
            const int pb_better = nc < gPBC[j];
            if (pb_better) gPBC[j] = nc;
            double gbest = L.SC[MBX_SC_GBEST];
            const int gb_better = nc < gbest;
            if (gb_better) { gbest = nc; L.SC[MBX_SC_GBEST_IDX] = j; }
            const double fes = L.SC[MBX_SC_FES] + 1;
            int log_index = (int)L.SC[MBX_SC_LOG_INDEX], cost_len = (int)L.SC[MBX_SC_COST_LEN];
            const bool dn = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, sc + MBX_NSCALAR);
            const double reward = (pre_cost - nc) / (L.SC[MBX_SC_RLPSO_MAXCOST] - gbest);
            L.SC[MBX_SC_GBEST] = gbest; L.SC[MBX_SC_FES] = fes; L.SC[MBX_SC_LOG_INDEX] = log_index; L.SC[MBX_SC_COST_LEN] = cost_len;
            L.SC[MBX_SC_DONE] = dn ? 1. : 0.; L.SC[MBX_SC_RETURN] += reward; L.SC[MBX_SC_GEN] = step; L.SC[MBX_SC_RLPSO_W] = w;
            L.SC[MBX_SC_RLPSO_CUR] = (j + 1) % NP;
            L.RED[8] = pb_better; L.RED[9] = gb_better; L.RED[10] = reward;
        }
        __syncthreads();
        MBX_PHASE(3);                                                // bookkeeping
        const int pb_better = (int)L.RED[8], gb_better = (int)L.RED[9];
        reward_sum += L.RED[10];
        done = L.SC[MBX_SC_DONE] != 0.;
        if (tid < D) {
            if (pb_better) gPB[j * D + tid] = nx;
            if (gb_better) { gGB[tid] = nx; L.GB[tid] = nx; }
            // the next particle's position: row j + 1 was last written by this same thread (or by the reset kernel)
            L.XC[tid] = gPos[((j + 1) % NP) * D + tid];
        }
        __syncthreads();
        MBX_PHASE(4);                                                // commit + next particle
    }
    if (tid < MBX_NSCALAR) sc[tid] = L.SC[tid];
    if (tid < D && state_out) { state_out[(int64_t)b * 2 * D + tid] = L.GB[tid]; state_out[(int64_t)b * 2 * D + D + tid] = L.XC[tid]; }
    if (tid == 0) {
        if (reward_out) reward_out[b] = reward_sum;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

}  // namespace mbx
