// mbx_lde.hpp — fused LDE generation kernels for gfx950 (reference: src/optimizer/lde_optimizer.py:6-198).
//
// One workgroup owns one instance: DE/current-to-pbest/1 mutation with per-individual F and CR supplied by the
// policy, binomial crossover, midpoint boundary repair, objective evaluation, selection, then the fitness-sorted
// population and the [NP + 10] histogram feature vector (the LSTM policy's next input) are produced in the same
// launch.  The population is kept sorted by fitness in HBM, which is what the reference's __order_by_f leaves
// behind after every __get_feature.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, align2

namespace mbx {

struct LdeLds {
    double *P, *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *FIT, *NC, *SF, *CR, *SORTED, *RED, *HS;
    int *PIDX, *R0, *R1, *JR, *HIST;
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC}; }
};

// BASELINE config 3 as written (pop = 100 at D = 30): the kernels of this geometry read the two D x D maps from global memory through scalar
// loads (matvec_rows_scalar) and keep no copy of them in LDS: 90.8 -> 76.4 KB per workgroup, i.e. TWO resident 512-thread workgroups per CU.
#ifndef MBX_LDE50_MAPS_IN_LDS
#define MBX_LDE50_MAPS_IN_LDS 0
#endif
__host__ __device__ constexpr bool lde_maps_in_lds(int NP, int D) { return !((NP == 100 || (NP == 50 && !MBX_LDE50_MAPS_IN_LDS)) && D == 30); }
// chunk of the register-resident row in matvec_rows_scalar_kc (0: whole row, for the 128-VGPR instantiation)
__host__ __device__ constexpr int lde_matvec_chunk(int NP, int D) { return NP == 50 && D == 30 ? 15 : 0; }

__host__ __device__ inline int64_t lde_lds_doubles(int NP, int D, bool maps = true)
{
    const int64_t NE = align2((int64_t)NP * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = maps ? align2((int64_t)D * D) : 0,
                  P = align2(NP), PI = align2((P + 1) / 2);
    const int64_t TS = eval_t_doubles(NP, D), PT = NE > TS ? NE : TS;     // parents and the evaluator's scratch T share storage
    return NE + PT + SC + 2 * DD + 4 * align2(D) + 2 * P + 16 + 8 + 4 * PI + 8;      // SF and CR live in Z (SC >= 512 >= 2 P)
}

__device__ __forceinline__ LdeLds lde_carve(double* base, int NP, int D, bool maps = true)
{
    const int64_t NE = align2((int64_t)NP * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = maps ? align2((int64_t)D * D) : 0,
                  P = align2(NP), PI = align2((P + 1) / 2);
    LdeLds L;
    double* p = base;
    const int64_t TS = eval_t_doubles(NP, D);
    // P (the parents) is dead between the mutation and the selection, exactly while the evaluator needs its scratch T: they share
    // storage and the unchanged parents are re-read from HBM after the evaluation (12 KB less LDS at NP = 50, D = 30: three resident
    // workgroups per CU instead of two).
    L.P = p; L.T = p; p += NE > TS ? NE : TS;  L.X = p; p += NE;  L.Z = p; p += SC;
    L.M1T = p; p += DD;  L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);  L.V0 = p; p += align2(D);  L.V1 = p; p += align2(D);  L.V2 = p; p += align2(D);
    L.FIT = p; p += P;  L.NC = p; p += P;
    L.SF = L.Z;  L.CR = L.Z + P;                                 // scale factors / crossover rates: last read by the mutation, Z first written by the evaluator
    L.SORTED = L.NC;                                             // the trials' costs are dead once the selection is done
    L.RED = p; p += 16;  L.HS = p; p += 8;
    L.PIDX = (int*)p; p += PI;  L.R0 = (int*)p; p += PI;  L.R1 = (int*)p; p += PI;  L.JR = (int*)p; p += PI;
    L.HIST = (int*)p;
    return L;
}

// np.histogram(a, 5) bin of one value over uniform bins on [first, last] (numpy/lib/_histograms_impl.py)
__device__ __forceinline__ int lde_hist_bin(double v, double first, double last)
{
    const int nb = MBX_LDE_BINS;
    const double step = (last - first) / nb;
    int idx = (int)((v - first) / (last - first) * nb);
    if (idx == nb) idx -= 1;
    const double lo = idx == nb ? last : first + idx * step;
    if (v < lo) idx -= 1;
    const double hi = idx + 1 == nb ? last : first + (idx + 1) * step;
    if (v >= hi && idx != nb - 1) idx += 1;
    return idx;
}

// __maxmin_norm + np.histogram of an ASCENDING fitness vector F[0..NP) (lde_optimizer.py:81-87,148-151):
// NORM[i] = normalised fitness, HIST[0..5) = bin counts.  All threads call.
// HIST_ZEROED: the caller cleared HIST before its last barrier (one barrier interval less); NORMF: a float32 copy of the normalised values (the policy's input)
template <bool HIST_ZEROED = false>
__device__ __forceinline__ void lde_norm_hist(const double* F, int NP, double* NORM, int* HIST, float* NORMF = nullptr)
{
    const int tid = threadIdx.x;
    if constexpr (!HIST_ZEROED) {
        if (tid < MBX_LDE_BINS) HIST[tid] = 0;
        __syncthreads();
    }
    const double mn = F[0], mx = F[NP - 1];
    double first = 0., last = mx != mn ? (mx - mn) / (mx - mn) : 0.;
    if (first == last) { first -= 0.5; last += 0.5; }
    for (int i = tid; i < NP; i += MBX_NT) {
        const double v = mx != mn ? (F[i] - mn) / (mx - mn) : 0.;
        if (NORM) NORM[i] = v;
        if (NORMF) NORMF[i] = (float)v;
        atomicAdd(&HIST[lde_hist_bin(v, first, last)], 1);
    }
    __syncthreads();
}

// the five bin counts (each <= NP <= 256) as one exactly represented integer, 10 bits per bin
__device__ __forceinline__ double lde_pack_hist(const int* H)
{
    unsigned long long v = 0;
    for (int k = MBX_LDE_BINS - 1; k >= 0; --k) v = (v << 10) | (unsigned long long)(H[k] & 1023);
    return (double)v;
}
__device__ __forceinline__ int lde_unpack_hist(double packed, int k) { return (int)(((unsigned long long)packed >> (10 * k)) & 1023); }

// __order_by_f (stable) + __get_feature (lde_optimizer.py:74-79,145-157): rows of L.P with fitness L.FIT are written
// to HBM in ascending-fitness order and the [NP+10] state vector is emitted.  hs/hcount = past_histo sum / length.
// PIDX_ZEROED: the caller cleared L.PIDX before its last barrier (k_lde_step does so in the selection loop: one barrier interval less)
template <bool PIDX_ZEROED = false>
__device__ __forceinline__ void lde_sort_emit(const LdeLds& L, int NP, int D, double* gPop, double* gFit, const double* hs,
                                              double hcount, double* state_out)
{
    const int tid = threadIdx.x;
    // stable rank of every individual; the NP x NP comparisons are spread over the whole workgroup (thread (i, part) counts over a
    // slice of j, partial counts meet in LDS) instead of NP threads walking NP entries each
    if constexpr (!PIDX_ZEROED) {
        for (int i = tid; i < NP; i += MBX_NT) L.PIDX[i] = 0;
        __syncthreads();
    }
    {
        const int parts = MBX_NT / NP > 0 ? MBX_NT / NP : 1;
        for (int w = tid; w < parts * NP; w += MBX_NT) {
            const int part = w / NP, i = w - part * NP;
            const int j0 = part * NP / parts, j1 = (part + 1) * NP / parts;
            const double fi = L.FIT[i];
            int cnt = 0;
            for (int j = j0; j < j1; ++j) { const double fj = L.FIT[j]; cnt += (fj < fi) || (fj == fi && j < i); }
            if (cnt) atomicAdd(&L.PIDX[i], cnt);
        }
    }
    __syncthreads();
    for (int i = tid; i < NP; i += MBX_NT) {
        const double fi = L.FIT[i];
        const int rank = L.PIDX[i];
        L.SORTED[rank] = fi;
        gFit[rank] = fi;
    }
    __syncthreads();
    const int NE = NP * D;
    const FastDiv fd(D);
    for (int e = tid; e < NE; e += MBX_NT) {
        const int i = fd.div(e), d = e - i * D;
        gPop[L.PIDX[i] * D + d] = L.P[e];
    }
    lde_norm_hist(L.SORTED, NP, state_out, L.HIST);
    if (tid < MBX_LDE_BINS) {
        state_out[NP + tid] = (double)L.HIST[tid];
        state_out[NP + MBX_LDE_BINS + tid] = hs[tid] / hcount;
    }
}

// ------------------------------------------------------------------------------------------------ reset
template <int THREADS, int NPC = 0, int DC = 0>
__global__ __launch_bounds__(THREADS) void k_lde_reset(BatchParams bp, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = NPC ? NPC : bp.NP, D = DC ? DC : bp.D, NE = NP * D;
    constexpr bool MAPS = lde_maps_in_lds(NPC, DC);
    constexpr int MD = MAPS ? 0 : DC;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const LdeLds L = lde_carve(smem, NP, D, MAPS);
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_LDE_ST_SCALARS(NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub;
    stage_problem<0, MAPS>(P, L.eval());
    for (int e = tid; e < NE; e += MBX_NT) {                     // pop = lb + U * (ub - lb)   (:65-71,133)
        double u;
        if (tape) u = tape[MBX_LDE_TAPE_CROSS(NP, D) + e];
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_LDE_ELEM); u = u53(w.x, w.y); }
        const double x = lb + u * (ub - lb);
        L.X[e] = x;
    }
    __syncthreads();
    population_costs<0, MD>(P, L.eval(), NP, rng, tape ? tape + MBX_LDE_TAPE_NOISE(NP, D) : nullptr, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B);
    for (int e = tid; e < NE; e += MBX_NT) L.P[e] = L.X[e];      // T is free again: the population moves into P for the sort
    for (int i = tid; i < NP; i += MBX_NT) L.FIT[i] = L.NC[i];
    if (tid < 8) { L.HS[tid] = tid < MBX_LDE_BINS ? (double)NP / MBX_LDE_BINS : 0.; S[MBX_LDE_ST_HSUM(NP, D) + tid] = L.HS[tid]; }
    double gb; int gi;
    block_argmin(L.NC, NP, L.RED, gb, gi);
    lde_sort_emit(L, NP, D, S + MBX_LDE_ST_POP(NP, D), S + MBX_LDE_ST_FIT(NP, D), L.HS, 1., state_out + (int64_t)b * (NP + 10));
    if (tid == 0) {
        S[MBX_LDE_ST_HSUM(NP, D) + MBX_LDE_BINS] = lde_pack_hist(L.HIST);            // the histogram of the state just emitted: the next update()'s past_histo entry
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1;
        sc[MBX_SC_DONE] = 0; sc[MBX_SC_RETURN] = 0; sc[MBX_SC_GEN] = 0; sc[MBX_SC_EPISODE] = episode; sc[MBX_SC_HCOUNT] = 1;
        sc[MBX_NSCALAR] = gb;
    }
}

// ------------------------------------------------------------------------------------------------ step
// Six waves per SIMD (80 VGPRs): at NP = 50, D = 30 the 512-thread workgroup needs 53.2 KB of LDS, so three of them share a CU (24 waves).
// The generic kernel fits in 72 VGPRs; the compile-time-geometry instantiation spills (55 VGPRs) and is still the faster one
// (all 30 noisy functions, 16 384 instances: 792 us at two resident workgroups -> 627 us at three).
// (pop = 100 / D = 30: two resident workgroups = four waves per SIMD, 128 VGPRs: room for the register-resident rows of matvec_rows_scalar)
#ifndef MBX_LDE_WAVES
#define MBX_LDE_WAVES __attribute__((amdgpu_waves_per_eu(4)))     // pop 100: two 512-thread workgroups per CU; pop 50: four 256-thread workgroups (round 3; three of 512 at 6 waves before)
#endif
// NPC / DC: population and dimension fixed at compile time (0 = taken from the batch), see k_rlepso_step
template <int THREADS, int NPC = 0, int DC = 0>
__global__ __launch_bounds__(THREADS) MBX_LDE_WAVES void k_lde_step(BatchParams bp, const float* __restrict__ actions,
                                                       double* __restrict__ state_out, double* __restrict__ reward_out,
                                                       uint8_t* __restrict__ done_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = NPC ? NPC : bp.NP, D = DC ? DC : bp.D, NE = NP * D;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_LDE_ST_SCALARS(NP, D);
    if (sc[MBX_SC_DONE] != 0.) {                                  // finished: state_out keeps the last features
        if (tid == 0) { if (reward_out) reward_out[b] = 0.; if (done_out) done_out[b] = 1; }
        return;
    }
    ConstProblem& P = *(ConstProblem*)(bp.problems + bp.problem_idx[b]);   // scalar loads on demand, no SGPR-resident copy
    MBX_PHASE_BEGIN
    constexpr bool MAPS = lde_maps_in_lds(NPC, DC);
    constexpr int MD = MAPS ? 0 : DC;
    const LdeLds L = lde_carve(smem, NP, D, MAPS);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const float* act = actions + (int64_t)b * (2 * NP);
    const int gen = (int)sc[MBX_SC_GEN] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)(int)sc[MBX_SC_EPISODE], true};
    const double lb = P.lb, ub = P.ub;
    double fes = sc[MBX_SC_FES];
    const double hcount = sc[MBX_SC_HCOUNT];

    const double* gPop = S + MBX_LDE_ST_POP(NP, D);
    for (int e = tid; e < NE; e += MBX_NT) L.P[e] = gPop[e];
    for (int i = tid; i < NP; i += MBX_NT) L.FIT[i] = S[MBX_LDE_ST_FIT(NP, D) + i];
    if (tid < 8) L.HS[tid] = S[MBX_LDE_ST_HSUM(NP, D) + tid];
    stage_problem<eval_dc(DC), MAPS>(P, L.eval());
    // p-best bound (:101-105): p = max(0, (P_MIN - P_INI) fes/maxFEs + P_INI), P_MIN = 2/NP, P_INI = 1
    const double p_rate = (2. / NP - 1) * fes / bp.max_fes + 1;
    const int bound = (int)ceil(NP * fmax(0., p_rate));
    for (int i = tid; i < NP; i += MBX_NT) {
        int pidx, r0, r1, jr;
        if (tape) {
            pidx = (int)tape[MBX_LDE_TAPE_PIDX(NP, D) + i]; r0 = (int)tape[MBX_LDE_TAPE_R0(NP, D) + i];
            r1 = (int)tape[MBX_LDE_TAPE_R1(NP, D) + i]; jr = (int)tape[MBX_LDE_TAPE_JRAND(NP, D) + i];
        } else {
            const U4 w = rng.draw((uint32_t)i, MBX_SITE_LDE_PART);
            pidx = (int)__umulhi(w.x, (uint32_t)bound);
            r0 = (int)__umulhi(w.y, (uint32_t)(NP - 1)); if (r0 >= i) r0 += 1;
            r1 = (int)__umulhi(w.z, (uint32_t)(NP - 2));
            const int lo = i < r0 ? i : r0, hi = i < r0 ? r0 : i;
            if (r1 >= lo) r1 += 1;
            if (r1 >= hi) r1 += 1;
            jr = (int)__umulhi(w.w, (uint32_t)D);
        }
        L.PIDX[i] = pidx; L.R0[i] = r0; L.R1[i] = r1; L.JR[i] = jr;
        const float sf32 = act[i];
        L.SF[i] = (double)sf32; L.CR[i] = (double)act[NP + i];
    }
    __syncthreads();
    MBX_PHASE(0);                                                 // staging + per-individual draws
    // histogram of the pre-update (sorted) fitness, appended to past_histo at :186.  It is the histogram the previous update() / reset computed for the state
    // it emitted (same sorted fitness vector), kept packed in the spare slot of the state's HSUM block: no second min-max + histogram pass over the population
    // (two barrier intervals and NP divisions per generation).
    int my_hist = tid < MBX_LDE_BINS ? lde_unpack_hist(L.HS[MBX_LDE_BINS], tid) : 0;

    MBX_PHASE(1);                                                 // histogram of the parents
    // ---- mutation + crossover + boundary repair (:88-130, 44-50, 31-38)
    const FastDiv fd(D);
    // The parents' storage doubles as evaluator scratch; a thread's own four parent coordinates stay in registers across the
    // evaluation, so the rows that lose the selection need no second trip to HBM (larger populations re-read them).
    // (KG = 2 groups per thread would cover pop = 100 at D = 30 on 512 threads; measured: the eight more live doubles across the evaluation cost
    // that instantiation more than the second read, 1.167 -> 1.239 ms per generation, so larger populations keep re-reading)
    constexpr int KG = 1;
    const bool kept = NE <= 4 * KG * MBX_NT;
    double keep[KG][4];
#pragma unroll
    for (int gk = 0; gk < KG; ++gk) { keep[gk][0] = keep[gk][1] = keep[gk][2] = keep[gk][3] = 0.; }
    // A thread owns groups of four consecutive elements: ONE Philox call (site LDE_ELEM, index e >> 2) carries their four crossover uniforms,
    // 32 bits each (u <= CR against a float32 rate: 2^-32 resolution; include/mbx_layout.h section 3) -- a call per element spent a quarter of
    // the kernel's integer work on draws of which half the words were thrown away.
    auto mutate_group = [&](int t, double (&kp)[4]) {
        U4 w{0, 0, 0, 0};
        if (!tape) w = rng.draw((uint32_t)t, MBX_SITE_LDE_ELEM);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * t + q;
            if (e >= NE) continue;
            const int i = fd.div(e), d = e - i * D;
            double u;
            if (tape) u = tape[MBX_LDE_TAPE_CROSS(NP, D) + e];
            else u = u32d(q == 0 ? w.x : q == 1 ? w.y : q == 2 ? w.z : w.w);
            if (d == L.JR[i]) u = 0.;
            const double xi = L.P[e], sf = L.SF[i], om = (double)(1.f - (float)sf);      // 1 - sf in float32, like the policy's tensor
            if (kept) kp[q] = xi;
            const int pidx = L.PIDX[i];
            double m;
            if (pidx == i) m = xi;
            else if (pidx < i) m = sf * L.P[pidx * D + d] + om * xi;
            else m = om * xi + sf * L.P[pidx * D + d];
            m = m + sf * (L.P[L.R0[i] * D + d] - L.P[L.R1[i] * D + d]);
            double c = u <= L.CR[i] ? m : xi;
            if (c < lb) c = (xi + lb) / 2.;
            else if (c > ub) c = (xi + ub) / 2.;
            L.X[e] = c;
        }
    };
    if constexpr (KG == 1) { for (int t = tid; 4 * t < NE; t += MBX_NT) mutate_group(t, keep[0]); }
    else {                                                           // two groups per thread, spelled out so that keep[][] stays in registers
        if (4 * tid < NE) mutate_group(tid, keep[0]);
        if (4 * (tid + MBX_NT) < NE) mutate_group(tid + MBX_NT, keep[KG - 1]);
    }
    __syncthreads();
    MBX_PHASE(2);                                                 // mutation + crossover
    population_costs<eval_dc(DC), MD, ConstProblem, lde_matvec_chunk(NPC, DC)>(P, L.eval(), NP, rng, tape ? tape + MBX_LDE_TAPE_NOISE(NP, D) : nullptr, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B);
    fes += NP;
    MBX_PHASE(3);                                                 // evaluation

    // ---- selection (:55-59): offspring survives when it is better than or equal to its parent
    const double bsf = L.FIT[0];                                   // population is sorted: minimum first
    for (int i = tid; i < NP; i += MBX_NT) {
        const int surv = L.NC[i] <= L.FIT[i];
        L.R0[i] = surv;
        if (surv) L.FIT[i] = L.NC[i];
        L.PIDX[i] = 0;                                             // the sort's rank accumulator (the p-best indices are dead since the mutation)
    }
    __syncthreads();
    // survivors take the trial vector; the other rows come back from the registers above (P's storage served as evaluator scratch meanwhile)
    auto select_group = [&](int t, const double (&kp)[4]) {         // same element ownership as the mutation
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * t + q;
            if (e >= NE) continue;
            const double parent = !kept ? gPop[e] : kp[q];
            L.P[e] = L.R0[fd.div(e)] ? L.X[e] : parent;
        }
    };
    if constexpr (KG == 1) { for (int t = tid; 4 * t < NE; t += MBX_NT) select_group(t, keep[0]); }
    else {
        if (4 * tid < NE) select_group(tid, keep[0]);
        if (4 * (tid + MBX_NT) < NE) select_group(tid + MBX_NT, keep[KG - 1]);
    }
    if (tid < MBX_LDE_BINS) { L.HS[tid] += (double)my_hist; S[MBX_LDE_ST_HSUM(NP, D) + tid] = L.HS[tid]; }
    __syncthreads();
    MBX_PHASE(4);                                                 // selection, survivors
    lde_sort_emit<true>(L, NP, D, S + MBX_LDE_ST_POP(NP, D), S + MBX_LDE_ST_FIT(NP, D), L.HS, hcount + 1, state_out + (int64_t)b * (NP + 10));
    if (tid == 0) S[MBX_LDE_ST_HSUM(NP, D) + MBX_LDE_BINS] = lde_pack_hist(L.HIST);
    const double bsf_next = L.SORTED[0];                           // the new best-so-far is the head of the sorted fitness vector: no argmin pass of its own

    MBX_PHASE(5);                                                 // sort + write-back + features
    if (tid == 0) {
        const double reward = (bsf - bsf_next) / bsf;             // :170
        int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
        const bool done = log_and_terminate(bp, P, fes, bsf_next, log_index, cost_len, sc + MBX_NSCALAR);
        sc[MBX_SC_GBEST] = bsf_next; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
        sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_RETURN] += reward; sc[MBX_SC_GEN] = gen; sc[MBX_SC_HCOUNT] = hcount + 1;
        if (reward_out) reward_out[b] = reward;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
}

}  // namespace mbx
