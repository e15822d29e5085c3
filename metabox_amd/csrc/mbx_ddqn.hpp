// mbx_ddqn.hpp — fused DE-DDQN step kernels for gfx950 (reference: src/optimizer/de_ddqn_optimizer.py:7-220 and
// src/optimizer/operators/{mutate,crossover,boundary_control}.py).
//
// One env step = one trial vector of one individual: mutation with the operator the Q-network chose, clipping,
// (Cr = 1) crossover, ONE objective evaluation, reward, the operator-credit records (N_tot / N_succ / OM per operator x
// metric x generation as rings of gen_max = 10, the OM_W window of 50), selection, and the 99-feature state for the
// next decision.  One workgroup per instance; the records live in HBM and are staged through LDS.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, align2

namespace mbx {

struct DqLds {
    double *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *NC, *COST, *REC, *FEAT, *RED, *MISC;
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC}; }
    // REC: N_tot[40] | N_succ[160] | OM_sum[160] | OM_max[160] | OM_W[300]
    __device__ __forceinline__ double* ntot() const { return REC; }
    __device__ __forceinline__ double* nsucc() const { return REC + 40; }
    __device__ __forceinline__ double* omsum() const { return REC + 200; }
    __device__ __forceinline__ double* ommax() const { return REC + 360; }
    __device__ __forceinline__ double* omw() const { return REC + 520; }
};
constexpr int kDqRec = 820;

// rows: how many candidates the evaluator sees at once -- NP in k_dq_reset, ONE in k_dq_step (a step evaluates a single trial vector:
// 17.9 KB instead of 40.9 KB at NP = 100, D = 12, so the register budget, not the LDS, decides how many workgroups share a CU)
__host__ __device__ inline int64_t dq_lds_doubles(int rows, int NP, int D)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    return NE + eval_t_doubles(rows, D) + SC + 2 * DD + 4 * align2(D) + 2 * P + kDqRec + 100 + 16 + 16 + 2 * align2(D);
}

__device__ __forceinline__ DqLds dq_carve(double* base, int rows, int NP, int D)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D), P = align2(NP);
    DqLds L;
    double* p = base;
    L.X = p; p += NE;  L.T = p; p += eval_t_doubles(rows, D);  L.Z = p; p += SC;  L.M1T = p; p += DD;  L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);  L.V0 = p; p += align2(D);  L.V1 = p; p += align2(D);  L.V2 = p; p += align2(D);
    L.NC = p; p += P;  L.COST = p; p += P;  L.REC = p; p += kDqRec;  L.FEAT = p; p += 100;  L.RED = p; p += 16;  L.MISC = p;
    return L;
}

__device__ __forceinline__ int dq_slot(int g, int gen) { return (((g - gen) % MBX_DQ_GENMAX) + MBX_DQ_GENMAX) % MBX_DQ_GENMAX; }

// __get_state (:76-129).  cost in L.COST, records in L.REC; row(k) gives a pointer to X[k] (global, post-selection).
// Draws r[5] for the next update (Philox counter `ctr` or tape) and stores it at gR.  All threads call.
template <class RowFn, class PT>
__device__ __forceinline__ void dq_features(const PT& P, const DqLds& L, int NP, int D, const BatchParams& bp, const Rng& rng,
                                            const double* tape, RowFn row, const double* gbpos, const double* prepos, double gbest,
                                            double gworst, double cpre, int pointer, int gen, double stag, double fes, int omw_len,
                                            double* gR, double* state_out)
{
    const int tid = threadIdx.x;
    for (int k = tid; k < MBX_DQ_NFEAT; k += MBX_NT) L.FEAT[k] = 0.;
    // mean / std of the cost vector (np.mean, np.std :79-80)
    double part = 0.;
    for (int i = tid; i < NP; i += MBX_NT) part += L.COST[i];
    const double mean = block_sum(part, L.RED) / NP;
    part = 0.;
    for (int i = tid; i < NP; i += MBX_NT) { const double t = fabs(L.COST[i] - mean); part += t * t; }
    const double var = block_sum(part, L.RED) / NP;
    int* R = reinterpret_cast<int*>(L.MISC);
    if (tid < 2) {                                                   // the two Philox calls side by side (lane 0: r0..r3, lane 1: r4)
        if (tape) { for (int j = 4 * tid; j < (tid ? 5 : 4); ++j) R[j] = (int)tape[MBX_DQ_TAPE_R(NP, D) + j]; }
        else {
            const U4 w = rng.draw((uint32_t)tid, MBX_SITE_DQ_R);
            if (tid == 0) {
                R[0] = (int)__umulhi(w.x, (uint32_t)NP); R[1] = (int)__umulhi(w.y, (uint32_t)NP);
                R[2] = (int)__umulhi(w.z, (uint32_t)NP); R[3] = (int)__umulhi(w.w, (uint32_t)NP);
            } else R[4] = (int)__umulhi(w.x, (uint32_t)NP);
        }
        for (int j = 4 * tid; j < (tid ? 5 : 4); ++j) gR[j] = (double)R[j];
    }
    __syncthreads();
    double md = 0.;
    for (int d = 0; d < D; ++d) md += (P.ub - P.lb) * (P.ub - P.lb);
    const double max_dist = sqrt(md), range = gworst - gbest, cp = L.COST[pointer];
    if (tid == 0) {
        L.FEAT[0] = (cp - gbest) / range;
        L.FEAT[1] = (mean - gbest) / range;
        L.FEAT[2] = sqrt(var) / (range / 2);
        L.FEAT[3] = (bp.max_fes - fes) / bp.max_fes;
        L.FEAT[4] = 1.;
        L.FEAT[5] = stag / bp.max_fes;
        L.FEAT[17] = (cp - cpre) / range;
    }
    if (tid < 7) {                                                  // 5 random peers, prebest, gbest (:86-93)
        const double* xp = row(pointer);
        const double* other = tid < 5 ? row(R[tid]) : (tid == 5 ? prepos : gbpos);
        double s = 0.;
        for (int d = 0; d < D; ++d) { const double t = xp[d] - other[d]; s += t * t; }
        const double dist = sqrt(s) / max_dist;
        if (tid < 5) { L.FEAT[6 + tid] = dist; L.FEAT[12 + tid] = (cp - L.COST[R[tid]]) / range; }
        else if (tid == 5) L.FEAT[11] = dist;
        else L.FEAT[18] = dist;
    }
    // (the three groups below sit in different waves when the workgroup has four; with two waves the distances have the first wave to themselves and the
    // second takes the window sums -- on all 64 lanes, a quarter of the window each -- and then the credit statistics; one wave runs all three in turn)
    const int t1 = MBX_NT >= 256 ? 64 : (MBX_NT >= 128 ? 64 : 16), t2 = MBX_NT >= 256 ? 128 : (MBX_NT >= 128 ? 64 : 32);
    const bool spread_window = MBX_NT == 128;
    if (spread_window && tid >= 64) {                               // OM_W window sums (:127-129), 16 (operator, metric) pairs x 4 interleaved window slices
        const int l = tid - 64, q = l & 15, op = q >> 2, m = q & 3;
        double s = 0.;
        for (int w = l >> 4; w < omw_len; w += 4) if ((int)L.omw()[w * 6] == op) s += L.omw()[w * 6 + 1 + m];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (l < 16) L.FEAT[83 + q] = s;
    }
    if (tid >= t1 && tid < t1 + 16) {                               // operator-credit statistics (:94-126)
        const int q = tid - t1, op = q >> 2;
        const int G = gen < MBX_DQ_GENMAX ? gen : MBX_DQ_GENMAX;
        const double *nt = L.ntot() + op * MBX_DQ_GENMAX, *ns = L.nsucc() + q * MBX_DQ_GENMAX, *os = L.omsum() + q * MBX_DQ_GENMAX,
                     *ox = L.ommax() + q * MBX_DQ_GENMAX;
        double sum_nt = 0., a = 0., b = 0., c = 0.;
        for (int g = 0; g < G; ++g) {
            const int s = dq_slot(g, gen);
            sum_nt += nt[s];
            if (nt[s] > 0) a += ns[s] / nt[s];
            b += os[s];
            if (ns[s] > 0) c += ox[s];
        }
        L.FEAT[19 + q] = a;
        L.FEAT[35 + q] = sum_nt > 0 ? b / sum_nt : b;
        L.FEAT[67 + q] = c;
        if (gen >= 2) {
            const int s0 = dq_slot(0, gen), s1 = dq_slot(1, gen);
            const double dn = nt[s0] - nt[s1];
            if (dn != 0 && ns[s0] > 0 && ns[s1] > 0) L.FEAT[51 + q] = (ox[s0] - ox[s1]) / (ox[s1] * fabs(dn));
        }
    }
    if (!spread_window && tid >= t2 && tid < t2 + 16) {             // OM_W window sums (:127-129)
        const int q = tid - t2, op = q >> 2, m = q & 3;
        double s = 0.;
        for (int w = 0; w < omw_len; ++w) if ((int)L.omw()[w * 6] == op) s += L.omw()[w * 6 + 1 + m];
        L.FEAT[83 + q] = s;
    }
    __syncthreads();
    for (int k = tid; k < MBX_DQ_NFEAT; k += MBX_NT) state_out[k] = L.FEAT[k];
}

// ------------------------------------------------------------------------------------------------ reset
__global__ __launch_bounds__(kThreads) void k_dq_reset(BatchParams bp, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = bp.NP, D = bp.D, NE = NP * D;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const DqLds L = dq_carve(smem, NP, NP, D);
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_DQ_ST_SCALARS(NP, D);
    double* ex = S + MBX_DQ_ST_EXTRA(NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub;
    stage_problem(P, L.eval());
    double* gX = S + MBX_DQ_ST_X(NP, D);
    for (int e = tid; e < NE; e += kThreads) {                      // X = rand * (ub - lb) + lb  (:48)
        double u;
        if (tape) u = tape[MBX_DQ_TAPE_POS(NP, D) + e];
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_LDE_ELEM); u = u53(w.x, w.y); }
        const double x = u * (ub - lb) + lb;
        L.X[e] = x; gX[e] = x;
    }
    for (int k = tid; k < kDqRec; k += kThreads) { L.REC[k] = 0.; S[MBX_DQ_ST_NTOT(NP, D) + k] = 0.; }
    __syncthreads();
    eval_rows(P, L.eval(), NP);
    for (int i = tid; i < NP; i += kThreads) {
        double f = L.NC[i];
        if (P.noise_kind != MBX_NOISE_NONE) {
            double a, bb, c;
            if (tape) { const double* t = tape + MBX_DQ_TAPE_NOISE_INIT(NP, D); a = t[i]; bb = t[NP + i]; c = t[2 * NP + i]; }
            else philox_noise(rng, (uint32_t)i, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B, P.noise_kind, a, bb, c);
            f = apply_noise(P, f, a, bb, c);
        }
        f = isnan(P.optimum) ? f : f - P.optimum;
        L.COST[i] = f; S[MBX_DQ_ST_COST(NP, D) + i] = f;
        L.NC[i] = -f;                                               // for the maximum below
    }
    __syncthreads();
    double gb, negw; int g0, iw;
    block_argmin(L.COST, NP, L.RED, gb, g0);
    block_argmin(L.NC, NP, L.RED, negw, iw);
    const double gworst = -negw;
    if (tid < D) { S[MBX_DQ_ST_GBPOS(NP, D) + tid] = L.X[g0 * D + tid]; S[MBX_DQ_ST_PREPOS(NP, D) + tid] = L.X[g0 * D + tid]; }
    auto row = [&](int k) -> const double* { return L.X + k * D; };
    dq_features(P, L, NP, D, bp, rng, tape, row, L.X + g0 * D, L.X + g0 * D, gb, gworst, gb, 0, 0, 0., (double)NP, 0,
                S + MBX_DQ_ST_R(NP, D), state_out + (int64_t)b * MBX_DQ_NFEAT);
    if (tid == 0) {
        ex[MBX_DQ_X_GWORST] = gworst; ex[MBX_DQ_X_CPRE] = gb; ex[MBX_DQ_X_POINTER] = 0; ex[MBX_DQ_X_GEN] = 0; ex[MBX_DQ_X_STAG] = 0;
        ex[MBX_DQ_X_OMWLEN] = 0; ex[MBX_DQ_X_G0] = g0; ex[MBX_DQ_X_GBVIEW] = 1; ex[MBX_DQ_X_PREVIEW] = 1;
        ex[MBX_DQ_X_MEDLO] = NAN; ex[MBX_DQ_X_MEDHI] = NAN;
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1; sc[MBX_SC_DONE] = 0;
        sc[MBX_SC_RETURN] = 0; sc[MBX_SC_GEN] = 0; sc[MBX_SC_EPISODE] = episode;
        sc[MBX_NSCALAR] = gb;
    }
}

// ------------------------------------------------------------------------------------------------ step
// TWO waves per instance.  A step is a chain of ~20 short phases, most of them a handful of lanes wide (one trial vector of D coordinates, 16 credit
// statistics, single-lane bookkeeping); with four waves per instance every phase boundary was a four-wave barrier and 2240 instances needed 2.2 rounds
// of resident workgroups (99 VGPRs: four per CU).  Small workgroups (17.9 KB of LDS: nine per CU) keep all 2240 instances of config 4's share resident
// at once; the only wide phase, the protein energy, is split over the workgroup's lanes.  Measured (k_dq_step<100, 12>, 2240 instances, one box; threads /
// waves per SIMD the compiler may assume): 64 / 3 60.7 us, **128 / 5 53.3-54.3 us**, 128 / 4 65.9, 128 / 6 57.6, 192 / 6 74.8, 256 / 6 65.2.
// (The helpers take the workgroup size from the launch.)
#ifndef MBX_DQ_WAVES
#define MBX_DQ_WAVES __attribute__((amdgpu_waves_per_eu(5)))      // nine two-wave workgroups per CU: 4.5 waves per SIMD, 96 VGPRs each
#endif
#ifndef MBX_DQ_STEP_THREADS
#define MBX_DQ_STEP_THREADS 128
#endif
constexpr int kDqStepThreads = MBX_DQ_STEP_THREADS;
// NPC / DC: population and dimension fixed at compile time (0 = taken from the batch), see k_rlepso_step
template <int NPC = 0, int DC = 0>
__global__ __launch_bounds__(kDqStepThreads) MBX_DQ_WAVES void k_dq_step(BatchParams bp, const int32_t* __restrict__ actions, double* __restrict__ state_out,
                                                      double* __restrict__ reward_out, uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = NPC ? NPC : bp.NP, D = DC ? DC : bp.D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_DQ_ST_SCALARS(NP, D);
    double* ex = S + MBX_DQ_ST_EXTRA(NP, D);
    if (sc[MBX_SC_DONE] != 0.) {
        if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; }
        return;
    }
    ConstProblem& P = *(ConstProblem*)(bp.problems + bp.problem_idx[b]);   // scalar loads on demand, no SGPR-resident copy
    MBX_PHASE_BEGIN
    const DqLds L = dq_carve(smem, 1, NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int action = actions[b];
    const int steps = (int)sc[MBX_SC_GEN] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)steps, (uint32_t)(int)sc[MBX_SC_EPISODE], true};
    const double lb = P.lb, ub = P.ub, F = 0.5;
    double* gX = S + MBX_DQ_ST_X(NP, D);
    double* gGB = S + MBX_DQ_ST_GBPOS(NP, D);
    double* gPRE = S + MBX_DQ_ST_PREPOS(NP, D);

    double gbest = sc[MBX_SC_GBEST], fes = sc[MBX_SC_FES];
    double gworst = ex[MBX_DQ_X_GWORST];
    const double cpre = ex[MBX_DQ_X_CPRE];
    const int p = (int)ex[MBX_DQ_X_POINTER], g0 = (int)ex[MBX_DQ_X_G0];
    int gen = (int)ex[MBX_DQ_X_GEN], omw_len = (int)ex[MBX_DQ_X_OMWLEN];
    int gb_view = (int)ex[MBX_DQ_X_GBVIEW], pre_view = (int)ex[MBX_DQ_X_PREVIEW];
    double stag = ex[MBX_DQ_X_STAG];
    int r[5];
    for (int j = 0; j < 5; ++j) r[j] = (int)S[MBX_DQ_ST_R(NP, D) + j];

    // ---- stage records, cost vector, problem constants
    for (int k = tid; k < kDqRec; k += MBX_NT) L.REC[k] = S[MBX_DQ_ST_NTOT(NP, D) + k];
    for (int i = tid; i < NP; i += MBX_NT) L.COST[i] = S[MBX_DQ_ST_COST(NP, D) + i];
    stage_problem<eval_dc(DC)>(P, L.eval());
    double* PRE = L.MISC + 8;            // prebest position used by the features (after a possible re-bind)
    double* GBP = PRE + align2(D) ;      // gbest position used by mutation / features
    __syncthreads();

    // ---- start of a population sweep (:132-142): prebest <- gbest, open a new generation slot in every ring
    if (p == 0) {
        pre_view = gb_view;
        gen += 1;
        const int s = dq_slot(0, gen);
        if (tid < 4) L.ntot()[tid * MBX_DQ_GENMAX + s] = 0.;
        if (tid < 16) { L.nsucc()[tid * MBX_DQ_GENMAX + s] = 0.; L.omsum()[tid * MBX_DQ_GENMAX + s] = 0.; L.ommax()[tid * MBX_DQ_GENMAX + s] = 0.; }
    }
    if (tid < D) {
        const double gbv = gb_view ? gX[g0 * D + tid] : gGB[tid];
        GBP[tid] = gbv;
        double prev = pre_view ? gX[g0 * D + tid] : gPRE[tid];
        if (p == 0 && !gb_view) { prev = gbv; gPRE[tid] = gbv; }
        PRE[tid] = prev;
    }
    __syncthreads();
    MBX_PHASE(0);                                                 // staging

    // ---- mutation (operators/mutate.py), clipping, Cr = 1 => trial = donor (crossover.py:6-18)
    if (tid < D) {
        const int d = tid;
        const double x0 = gX[r[0] * D + d], x1 = gX[r[1] * D + d], x2 = gX[r[2] * D + d], x3 = gX[r[3] * D + d], x4 = gX[r[4] * D + d];
        const double xp = gX[p * D + d];
        double v;
        if (action == 0) v = x0 + F * (x1 - x2);
        else if (action == 1) v = x0 + F * (x1 - x2 + x3 - x4);
        else if (action == 2) v = x0 + F * (GBP[d] - x0 + x1 - x2 + x3 - x4);
        else v = xp + F * (x0 - xp + x1 - x2);
        L.X[d] = fmin(fmax(v, lb), ub);
    }
    __syncthreads();
    MBX_PHASE(1);                                                 // mutation
    eval_rows<eval_dc(DC)>(P, L.eval(), 1);
    MBX_PHASE(2);                                                 // evaluation
    // ---- median of the current costs (:171) by rank counting
    // c_i is the k-th smallest (0-based) iff #{c_j < c_i} <= k < #{c_j <= c_i}: equal costs are interchangeable for the VALUE at rank k, so no
    // index tie-break is needed (two compares per pair instead of three and their logic).
    // A step replaces at most one cost, so the two order statistics the previous step found are usually still the ones: they are cached in the
    // state (MBX_DQ_X_MEDLO / MEDHI) and re-validated by ONE counting pass over the cost vector (each thread a few elements, the four counts
    // packed into one reduction); only when the replaced cost crossed them does the workgroup rank the whole vector (NP^2 compares, a tenth
    // of the step's instructions when it ran every time).
    {
        const int K = NP / 2;
        const double mlo = ex[MBX_DQ_X_MEDLO], mhi = ex[MBX_DQ_X_MEDHI];
        double packed = 0.;                                          // four counts <= NP <= 256 in 12-bit fields of one exactly represented integer
        for (int i = tid; i < NP; i += MBX_NT) {
            const double c = L.COST[i];
            packed += (double)((c < mhi) + ((c <= mhi) << 12)) + 16777216. * (double)((c < mlo) + ((c <= mlo) << 12));
        }
        const unsigned long long all = (unsigned long long)block_sum(packed, L.RED);
        const int nl_hi = (int)(all & 4095), ne_hi = (int)((all >> 12) & 4095), nl_lo = (int)((all >> 24) & 4095), ne_lo = (int)((all >> 36) & 4095);
        const bool valid = nl_hi <= K && K < ne_hi && ((NP & 1) || (nl_lo <= K - 1 && K - 1 < ne_lo));
        if (valid) {
            if (tid == 0) { L.RED[8] = mhi; L.RED[9] = mlo; }
        } else {
            for (int i = tid; i < NP; i += MBX_NT) {
                const double ci = L.COST[i];
                int nless = 0, nle = 0;
#pragma unroll 4
                for (int j = 0; j < NP; ++j) { const double cj = L.COST[j]; nless += cj < ci; nle += cj <= ci; }
                if (nless <= K && K < nle) L.RED[8] = ci;
                if (nless <= K - 1 && K - 1 < nle) L.RED[9] = ci;
            }
        }
    }
    __syncthreads();
    MBX_PHASE(3);                                                 // median

    // ---- OM_W window (:178-187): when full, evict the first entry of the same operator, else the worst trial (first maximum), and
    // close the gap.  Depends on the action only, so it is done here by the whole block instead of inside the one-lane section
    // below (the serial search + 300-element shift were a quarter of the step's latency).
    if (omw_len >= MBX_DQ_W) {
        double* W = L.omw();
        if (tid < 64) {
            const bool in = tid < omw_len;
            const bool same = in && (int)W[tid * 6] == action;
            const unsigned long long m = __ballot(same);
            double v = in ? -W[tid * 6 + 5] : INFINITY;
            int idx = tid;
            wave_argmin(v, idx);
            if (tid == 0) L.MISC[5] = m ? (double)(__ffsll((long long)m) - 1) : (double)idx;
        }
        __syncthreads();
        const int del = (int)L.MISC[5], last = (omw_len - 1) * 6;
        constexpr int NV = (MBX_DQ_W * 6 + kDqStepThreads - 1) / kDqStepThreads;        // window words per thread
        double v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) { const int k = del * 6 + tid + j * MBX_NT; v[j] = k < last ? W[k + 6] : 0.; }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NV; ++j) { const int k = del * 6 + tid + j * MBX_NT; if (k < last) W[k] = v[j]; }
        omw_len -= 1;
        __syncthreads();
    }
    MBX_PHASE(4);                                                 // window eviction
    // ---- sequential bookkeeping (:164-203) by one lane
    if (tid == 0) {
        double tc = L.NC[0];
        if (P.noise_kind != MBX_NOISE_NONE) {
            double a, bb, c;
            if (tape) { a = tape[MBX_DQ_TAPE_NOISE(NP, D)]; bb = tape[MBX_DQ_TAPE_NOISE(NP, D) + 1]; c = tape[MBX_DQ_TAPE_NOISE(NP, D) + 2]; }
            else philox_noise(rng, 0u, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B, P.noise_kind, a, bb, c);
            tc = apply_noise(P, tc, a, bb, c);
        }
        tc = isnan(P.optimum) ? tc : tc - P.optimum;
        const double cpv = L.COST[p];
        const double reward = fmax(cpv - tc, 0.);
        const int s0 = dq_slot(0, gen);
        L.ntot()[action * MBX_DQ_GENMAX + s0] += 1;
        const double median = (NP & 1) ? L.RED[8] : (L.RED[9] + L.RED[8]) / 2;
        ex[MBX_DQ_X_MEDHI] = L.RED[8]; ex[MBX_DQ_X_MEDLO] = (NP & 1) ? L.RED[8] : L.RED[9];
        const double om[4] = {cpv - tc, cpre - tc, gbest - tc, median - tc};
        for (int m = 0; m < 4; ++m)
            if (om[m] > 0) {
                const int k = (action * 4 + m) * MBX_DQ_GENMAX + s0;
                if (L.nsucc()[k] == 0 || om[m] > L.ommax()[k]) L.ommax()[k] = om[m];
                L.nsucc()[k] += 1; L.omsum()[k] += om[m];
            }
        double* W = L.omw();
        double* e = W + omw_len * 6;                                // the window was already trimmed above
        e[0] = action; e[1] = om[0]; e[2] = om[1]; e[3] = om[2]; e[4] = om[3]; e[5] = tc;
        omw_len += 1;
        if (tc >= gbest) stag += 1;
        int sel = 0, newbest = 0;
        if (tc <= cpv) { sel = 1; L.COST[p] = tc; if (tc <= gbest) { gbest = tc; newbest = 1; } }
        if (tc > gworst) gworst = tc;
        L.RED[10] = tc; L.RED[11] = reward; L.RED[12] = sel; L.RED[13] = newbest; L.RED[14] = gbest; L.RED[15] = gworst;
        L.MISC[6] = stag; L.MISC[7] = omw_len;
    }
    __syncthreads();
    MBX_PHASE(5);                                                 // bookkeeping
    const double tc = L.RED[10], reward = L.RED[11];
    const int sel = (int)L.RED[12], newbest = (int)L.RED[13];
    gbest = L.RED[14]; gworst = L.RED[15]; stag = L.MISC[6]; omw_len = (int)L.MISC[7]; fes += 1;
    (void)tc;
    if (sel && tid < D) {
        gX[p * D + tid] = L.X[tid];
        if (pre_view && p == g0) PRE[tid] = L.X[tid];               // the numpy view follows the row it aliases
        if (newbest) { gGB[tid] = L.X[tid]; GBP[tid] = L.X[tid]; }
    }
    if (newbest) gb_view = 0;
    if (sel && tid == 0) S[MBX_DQ_ST_COST(NP, D) + p] = L.COST[p];
    const int pointer = (p + 1) % NP;
    __syncthreads();

    MBX_PHASE(6);                                                 // selection
    // ---- next state (:209) — rows are read from HBM except the one this step rewrote (still in LDS)
    auto row = [&](int k) -> const double* { return (sel && k == p) ? L.X : gX + k * D; };
    dq_features(P, L, NP, D, bp, rng, tape, row, GBP, PRE, gbest, gworst, cpre, pointer, gen, stag, fes, omw_len,
                S + MBX_DQ_ST_R(NP, D), state_out + (int64_t)b * MBX_DQ_NFEAT);
    MBX_PHASE(7);                                                 // features
    for (int k = tid; k < kDqRec; k += MBX_NT) S[MBX_DQ_ST_NTOT(NP, D) + k] = L.REC[k];
    if (tid == 0) {
        int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
        double* cost = sc + MBX_NSCALAR;
        const bool done = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, cost);
        ex[MBX_DQ_X_GWORST] = gworst; ex[MBX_DQ_X_POINTER] = pointer; ex[MBX_DQ_X_GEN] = gen; ex[MBX_DQ_X_STAG] = stag;
        ex[MBX_DQ_X_OMWLEN] = omw_len; ex[MBX_DQ_X_GBVIEW] = gb_view; ex[MBX_DQ_X_PREVIEW] = pre_view;
        sc[MBX_SC_GBEST] = gbest; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
        sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_RETURN] += reward; sc[MBX_SC_GEN] = steps;
        if (reward_out) reward_out[b] = reward;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

}  // namespace mbx
