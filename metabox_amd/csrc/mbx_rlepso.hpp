// mbx_rlepso.hpp — fused RLEPSO generation kernels for gfx950.
//
// One workgroup (256 threads = 4 waves) owns one (problem x run) instance.  Per launch it streams the
// instance's state block in from HBM, keeps pbest positions / candidate positions / evaluation scratch
// in LDS, performs one whole RLEPSO_Optimizer.update() (reference: src/optimizer/rlepso_optimizer.py:
// 173-263) including the objective evaluation, and streams the state back out.
//
// Mapping: element e = i*D + d (particle i, dimension d) is owned by thread e % 256; particle-level work
// (pbest/stagnation bookkeeping) is done by thread i; argmin uses wave shuffles.
#pragma once
#include "mbx_device.hpp"

namespace mbx {

struct BatchParams {
    const DevProblem* problems;
    const int32_t* problem_idx;
    const uint64_t* seeds;
    double* state;               // [B][state_stride]
    int64_t state_stride;
    const double* tape;          // nullptr -> Philox
    int64_t tape_stride;
    const int32_t* order;        // launch order: workgroup w owns instance order[w] (most expensive objectives first)
    const double* pci;           // [NP] RLEPSO learning-probability curve pci_i (rlepso_optimizer.py:23-24), computed at batch creation
    int32_t NP, D, max_fes, log_interval, n_logpoint, early_stop, n_group, B;
    int64_t sc_off;              // offset of the scalar block inside an instance's state (per algorithm)
    unsigned long long* clk;     // nullptr, or 2 words {sum of shader cycles (s_memtime), sum of 100 MHz ticks (s_memrealtime)} over the lifetimes of the resident RLEPSO kernel's
                                 // workgroups (mbx_debug_clock_slots; bench.py prices the kernel's vector-issue bound at the clock its own waves saw)
};

// optimizer.cost bookkeeping shared by every update() of the reference (e.g. rlepso_optimizer.py:241-261): append gbest when fes
// reaches the next log point (one append per call at most), decide termination, and on termination overwrite the last entry if the
// curve is already n_logpoint + 1 long, else append.  `cost` is the instance's curve in HBM; returns is_done.
template <class PT>
__device__ __forceinline__ bool log_and_terminate(const BatchParams& bp, const PT& P, double fes, double gbest, int& log_index,
                                                  int& cost_len, double* __restrict__ cost)
{
    if (fes >= (double)log_index * bp.log_interval) { log_index += 1; cost[cost_len++] = gbest; }
    bool done = fes >= bp.max_fes;
    if (!isnan(P.optimum) && bp.early_stop) done = done || gbest <= 1e-8;
    if (done) {
        if (cost_len >= bp.n_logpoint + 1) cost[cost_len - 1] = gbest;
        else cost[cost_len++] = gbest;
    }
    return done;
}

// LDS carve-up (all offsets in doubles; base is 16-byte aligned, every array starts 16-byte aligned)
struct RlLds {
    double *PB, *X, *Z, *T, *M1T, *M2T, *DSH, *V0, *V1, *V2, *PBC, *NC, *PNI, *CMUT, *R1, *R2, *GB, *COEF, *RED;
    int *IMPR, *MASK, *RANK;
    // KB (NE bytes): rank of the FDR exemplar of every (rank, dimension), carved out of the part of Z that R1 / R2 / COEF leave free
    // (all dead before the evaluator first writes Z)
    uint8_t* KB;
    // FL (one uint16 per work item of the FDR pass) / FLN: the items whose scan met a near-tie, appended by their owners and settled wave-cooperatively (fdr_settle); same
    // free part of Z, behind KB
    uint16_t* FL;
    int* FLN;
    // NCS [NP]: the cost column the FDR scan reads -- NC with the rows that are bitwise copies of the row one rank up (same cost, same position: particles sitting on
    // the same corner of the box) replaced by a huge value, so that the scan never takes them and never mistakes them for near-ties (rl_mark_copies); same free part of Z
    double* NCS;
    int zd;                       // doubles of Z (rl_z_doubles)
    __device__ __forceinline__ EvalLds eval() const { return EvalLds{X, Z, T, M1T, M2T, DSH, V0, V1, V2, NC, zd}; }
};

__host__ __device__ inline int64_t align2(int64_t n) { return (n + 1) & ~(int64_t)1; }

// doubles of the move phase's tables inside Z: KB (NE bytes, padded to 16) | FLN + pad (16 bytes) | FL (up to NE uint16)
__host__ __device__ inline int64_t rl_aux_doubles(int NP, int D) { const int64_t NE = (int64_t)NP * D; return ((NE + 15) / 16 * 16 + 16 + 2 * NE + 7) / 8; }

// evaluator scratch Z: n*D doubles, at least 2 per thread for the block reductions, and room for the move phase's R1 | R2 | COEF | NCS | tables
__host__ __device__ inline int64_t rl_z_doubles(int NP, int D)
{
    const int64_t NE = align2((int64_t)NP * D);
    int64_t need = 3 * align2(NP) + 96 + rl_aux_doubles(NP, D);
    if (need < 2 * kThreads) need = 2 * kThreads;
    if (D == 12 && need < 4 * 4 * 100) need = 4 * 4 * 100;      // protein docking (D = 12, 100 atoms): one 32-byte-per-atom table per wave of a 256-thread workgroup (eval_rows_protein)
    return align2(NE > need ? NE : need);
}

// chunk of the register-resident row in the resident kernel's matvec (matvec_rows_scalar_kc; 0: whole row): at D = 40 the 80 row registers collide
// with the resident velocities / pbest positions
// (config 5, A/B on one box: whole row 2.00 ms, chunks of 20 / 10 / 8: 1.886 / 1.830 / 1.854 ms per generation of 8192 instances in round 2; with the
// fma-chain matvec, round 3: whole row 2.53, chunks of 20 / 10 / 8 / 5 / 4: 1.667 / 1.672 / 1.628 / 1.887 / 1.700 ms -- register allocation, not arithmetic, decides)
#ifndef MBX_RUN_KC40
#define MBX_RUN_KC40 8
#endif
#ifndef MBX_RUN_KC30
#define MBX_RUN_KC30 0
#endif
__host__ __device__ constexpr int rl_run_matvec_chunk(int D) { return D == 40 ? MBX_RUN_KC40 : D == 30 ? MBX_RUN_KC30 : 0; }

// The bbob D = 30 geometry of the reference (NP = 100, rlepso_optimizer.py:9; `--dim 30`): its kernels read the two D x D maps from global memory
// (matvec_rows_scalar) and keep no LDS copy: 90.0 -> 75.6 KB per workgroup, i.e. TWO resident 512-thread workgroups per CU instead of one.
__host__ __device__ constexpr bool rl_maps_in_lds(int NP, int D) { return !(NP == 100 && D == 30); }

__host__ __device__ inline int64_t rl_lds_doubles(int NP, int D, bool maps = true)
{
    const int64_t NE = align2((int64_t)NP * D), SC = rl_z_doubles(NP, D), DD = maps ? align2((int64_t)D * D) : 0,
                  P = align2(NP), TS = eval_t_doubles(NP, D);
    // PB (aliased by the evaluator's scratch T once the velocity phase is over), X: NE each; Z: SC (R1, R2, COEF and the byte tables of
    // the move phase, all dead before the first evaluation, live in it); M1T, M2T; DSH, V0, V1, V2, GB: D each; PBC, NC, PNI, CMUT: P each;
    // RED: 16; 3 int arrays  (29.8 KB at NP = 100, D = 10: five workgroups per CU)
    return TS + NE + SC + 2 * DD + 4 * P + 5 * align2(D) + 16 + 3 * align2((P + 1) / 2);
}

__device__ __forceinline__ RlLds rl_carve(double* base, int NP, int D, bool maps = true)
{
    const int64_t NE = align2((int64_t)NP * D), SC = rl_z_doubles(NP, D), DD = maps ? align2((int64_t)D * D) : 0,
                  P = align2(NP), TS = eval_t_doubles(NP, D);
    RlLds L;
    double* p = base;
    L.PB = p; L.T = p; p += TS;      // T reuses PB's storage (see rl_commit)
    L.X = p; p += NE;
    L.Z = p; L.zd = (int)SC;
    L.R1 = p; L.R2 = p + P; L.COEF = p + 2 * P;   // per-particle draws and group coefficients: last read in the move phase, Z first written by the evaluator
    L.NCS = p + 2 * P + 96;
    L.KB = (uint8_t*)(p + 3 * P + 96);
    L.FLN = (int*)(L.KB + ((int64_t)NP * D + 15) / 16 * 16);
    L.FL = (uint16_t*)(L.FLN + 4);
    p += SC;
    L.M1T = p; p += DD;
    L.M2T = p; p += DD;
    L.DSH = p; p += align2(D);
    L.V0 = p; p += align2(D);
    L.V1 = p; p += align2(D);
    L.V2 = p; p += align2(D);
    L.PBC = p; p += P;
    L.NC = p; p += P;
    L.PNI = p; p += P;
    L.CMUT = p; p += P;
    L.GB = p; p += align2(D);
    L.RED = p; p += 16;
    L.IMPR = (int*)p; p += align2((P + 1) / 2);
    L.MASK = (int*)p; p += align2((P + 1) / 2);
    L.RANK = (int*)p;
    return L;
}

// pbest / gbest bookkeeping shared by update() and __reinit() (rlepso_optimizer.py:200-222, 145-168).
// Candidate positions are in L.X, their costs in L.NC.  `stagnation` additionally updates per_no_improve
// against the previous c_cost (:225-233), which only update() does.
__device__ __forceinline__ void rl_commit(const RlLds& L, int NP, int D, bool stagnation, double& gbest, int& gbest_idx,
                                          double* __restrict__ gPB, double* __restrict__ gCC)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < NP; i += MBX_NT) {
        const double nc = L.NC[i];
        if (stagnation) L.PNI[i] = nc < gCC[i] ? 0. : L.PNI[i] + 1;   // c_cost lives in HBM only: thread i is its sole reader/writer
        const int impr = nc < L.PBC[i];
        if (impr) L.PBC[i] = nc;
        L.IMPR[i] = impr;
        gCC[i] = nc;
    }
    double cbv; int cb;
    block_argmin(L.NC, NP, L.RED, cbv, cb);         // contains the barriers that publish IMPR / CC
    const bool better = cbv < gbest;
    if (better) { gbest = cbv; gbest_idx = cb; }
    const int NE = NP * D;
    const FastDiv fd(D);
    for (int e = tid; e < NE; e += MBX_NT) {
        const int i = fd.div(e);
        if (L.IMPR[i]) gPB[e] = L.X[e];                // pbest_position <- new position, straight to HBM
    }
    if (better && tid < D) L.GB[tid] = L.X[cb * D + tid];
    __syncthreads();
}

// Everything the velocity phase needs, bundled so that the per-item routine can be a template on the item width.
struct MoveCtx {
    const RlLds& L; const double* pci; const double* tape; const Rng& rng; double* gPos; double* gVel;
    const int* ORDER; const int* NLESS; const int* RANK; int NP, D, G; double lb, ub, vmax; const FastDiv& fg;
};

// Two-headed Gaussian MLP policy (mbx_gauss_mlp in include/mbx.h): RLEPSO's Actor (src/agent/rlepso_agent.py:9-47) and RL-PSO's
// PolicyNetwork (src/agent/rl_pso_agent.py:9-47) share the architecture and differ in the heads.
struct GaussMlp {
    const float* w;              // packed weights
    int32_t in_dim, h1, h2, out_dim;
    float min_sigma, max_sigma;
    int32_t variant;             // MBX_POLICY_RLEPSO / MBX_POLICY_RLPSO
};

__host__ __device__ inline int gauss_mlp_net_floats(int in, int h1, int h2, int A) { return in * h1 + h1 + h1 * h2 + h2 + h2 * A + A; }

// (mu, sigma) from the two nets' raw outputs, float32 like the reference's torch modules
__device__ __forceinline__ void gauss_head(const GaussMlp& net, float am, float as, float& mu, float& sigma)
{
    mu = (tanhf(am) + 1.f) / 2.f;
    if (net.variant == MBX_POLICY_RLPSO) sigma = fminf(fmaxf((tanhf(as) + 1.f) / 2.f, net.min_sigma), net.max_sigma);   // rl_pso_agent.py:27-29
    else sigma = (tanhf(as) + 1.f) / 2.f * (net.max_sigma - net.min_sigma) + net.min_sigma;                               // rlepso_agent.py:25
}

// Normal(mu, sigma).sample() with the instance's Philox stream, then the agent's post-processing:
//   RLEPSO: clamp to [0, 1] (rlepso_agent.py:32);  RL-PSO: samples outside [0, 1) are re-folded as
//   (a + 3 sigma - mu) * (1/6 * sigma) (rl_pso_agent.py:34-35, precedence as written there).
// The policy is float32 and its noise has no counterpart in the reference (torch's global generator), so Box-Muller runs on the
// hardware float32 transcendentals: v_log_f32 (log2) and v_cos_f32 (argument in revolutions, i.e. cos(2 pi u) directly).
__device__ __forceinline__ float sample_action(const Rng& rng, int j, float mu, float sigma, int variant)
{
    const U4 w = rng.draw((uint32_t)j, MBX_SITE_POLICY);
    const float u1 = (float)((w.x >> 8) + 1u) * 5.9604644775390625e-8f;      // (0, 1]
    const float u2 = (float)(w.y >> 8) * 5.9604644775390625e-8f;             // [0, 1)
    const float r = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));     // sqrt(-2 ln u1)
    const float a = mu + sigma * (r * __builtin_amdgcn_cosf(u2));
    if (variant == MBX_POLICY_RLPSO) return fabsf(a - 0.5f) >= 0.5f ? (a + 3.f * sigma - mu) * ((1.f / 6.f) * sigma) : a;
    return fminf(fmaxf(a, 0.f), 1.f);
}

// Per-phase wall-cycle accounting of k_rlepso_step (instrumented builds only: -DMBX_PHASE_TIMING; tools/kbench.py --phases).
#ifdef MBX_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[8192 * 16];         // [block][phase], no atomics: plain accumulation by the owning block
#define MBX_PHASE_BEGIN unsigned long long ph_t = clock64();
#define MBX_PHASE(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = clock64(); g_phase_cycles[(blockIdx.x & 8191) * 16 + (k)] += t_ - ph_t; ph_t = t_; } } while (0)
#else
#define MBX_PHASE_BEGIN
#define MBX_PHASE(k)
#endif

// 256-thread workgroups: 29.8 KB of LDS lets five of them share a CU, so cap the registers at 96 for five waves per SIMD (-5.7 % against
// four; two VGPRs spill).  The 512- / 1024-thread instantiations own their CU and only need the four waves per SIMD a 1024-thread workgroup is.
#ifndef MBX_RL_WAVES
#define MBX_RL_WAVES __attribute__((amdgpu_waves_per_eu(THREADS == 256 ? 5 : 4)))
#endif
#ifndef MBX_FDR_UNROLL
#define MBX_FDR_UNROLL 4
#endif
#ifndef MBX_FDR_OWN_AT
#define MBX_FDR_OWN_AT 64           // flagged items per swarm and generation above which every lane settles its own items instead of one wave per coordinate (fdr_pass)
#endif

// if (lhs < rhs) { ab = a; bb = b; kb = k; } as three EXEC-masked moves (v_cmpx + v_mov_b64 x2 + v_mov_b32) instead of the five
// v_cndmask_b32 the compiler's if-conversion produces: the FDR scan below is VALU-issue bound and this is its innermost statement.
__device__ __forceinline__ void fdr_take(double lhs, double rhs, double& ab, double& bb, int& kb, double a, double b, int k)
{
#ifdef MBX_FDR_BRANCH
    // experiment (tools/fdr_sim/take_stats.py: on real swarms no lane of a wave takes in 50.5 % of the wave-steps): a wave-uniform branch around the masked moves
    if (__builtin_amdgcn_ballot_w64(lhs < rhs) == 0ull) return;
#endif
#ifdef MBX_FDR_SELECT
    if (lhs < rhs) { ab = a; bb = b; kb = k; }
#else
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
                 "v_cmpx_lt_f64_e32 %[l], %[r]\n\t"
                 "v_mov_b64 %[ab], %[a]\n\t"
                 "v_mov_b64 %[bb], %[b]\n\t"
                 "v_mov_b32 %[kb], %[k]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [ab] "+v"(ab), [bb] "+v"(bb), [kb] "+v"(kb), [sv] "=&s"(saved)
                 : [l] "v"(lhs), [r] "v"(rhs), [a] "v"(a), [b] "v"(b), [k] "s"(k)
                 : "vcc");
#endif
}

// ---- exact resolution of near-ties in the FDR scan (fdr_exact<.., TIE = true>: the DEFAULT of every RLEPSO kernel; TIE = false is the batch option
// MBX_F_FDR_FAST of include/mbx.h) -------------------------------------------------------------------------------------------------
// if (dif < 0) { ab = a; bb = b; kb = k; }: fdr_take on the SIGN of dif = a_j b* - fl(a* b_j) (one fma instead of the second product: same instruction count)
__device__ __forceinline__ void fdr_take_neg(double dif, double& ab, double& bb, int& kb, double a, double b, int k)
{
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_f64_e32 0, %[d]\n\t"
        "v_mov_b64 %[ab], %[a]\n\t"
        "v_mov_b64 %[bb], %[b]\n\t"
        "v_mov_b32 %[kb], %[k]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ab] "+v"(ab), [bb] "+v"(bb), [kb] "+v"(kb), [sv] "=&s"(saved)
        : [d] "v"(dif), [a] "v"(a), [b] "v"(b), [k] "s"(k)
        : "vcc");
}
// the two coordinates of a work item in ONE block: both differences are complete before the first compare issues, so neither compare waits on the
// instruction right in front of it
__device__ __forceinline__ void fdr_take_neg2(double dif0, double dif1, double& ab0, double& bb0, int& kb0, double& ab1, double& bb1, int& kb1,
                                              double a, double b0, double b1, int k)
{
    unsigned long long saved;
    asm("s_mov_b64 %[sv], exec\n\t"
        "v_cmpx_gt_f64_e32 0, %[d0]\n\t"
        "v_mov_b64 %[ab0], %[a]\n\t"
        "v_mov_b64 %[bb0], %[b0]\n\t"
        "v_mov_b32 %[kb0], %[k]\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_cmpx_gt_f64_e32 0, %[d1]\n\t"
        "v_mov_b64 %[ab1], %[a]\n\t"
        "v_mov_b64 %[bb1], %[b1]\n\t"
        "v_mov_b32 %[kb1], %[k]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ab0] "+v"(ab0), [bb0] "+v"(bb0), [kb0] "+v"(kb0), [ab1] "+v"(ab1), [bb1] "+v"(bb1), [kb1] "+v"(kb1), [sv] "=&s"(saved)
        : [d0] "v"(dif0), [d1] "v"(dif1), [a] "v"(a), [b0] "v"(b0), [b1] "v"(b1), [k] "s"(k)
        : "vcc");
}
// running minimum of |dif| over the pairs of a scan, kept on the HIGH dwords read as float32 (sign | exponent | 20 mantissa bits of a float64 are monotone in
// the float32 order for finite values): ONE v_min3_f32 covers the two coordinates of a work item
__device__ __forceinline__ float f64_hi_as_f32(double v) { return __int_as_float(__double2hiint(v)); }
__device__ __forceinline__ float fdr_min3_abs(float m, float x, float y)
{
    float r;
    asm("v_min3_f32 %[r], %[m], |%[x]|, |%[y]|" : [r] "=v"(r) : [m] "v"(m), [x] "v"(x), [y] "v"(y));
    return r;
}
// Rows of the rank-ordered pbest table that are bitwise copies of the row one rank up -- same cost, same position: particles resting on the same corner of the box (linear
// slope: most of the swarm, for dozens of generations) -- can never be np.argmin's choice (the copy with the lower index precedes them in rank order and ties with them in
// every coordinate), but in the scan each of them is an exact tie with the running best: every item above them would be flagged.  The scan therefore reads their cost as 1e300
// (NCS): a huge positive numerator, never taken, nowhere near a tie.  Called only when equal pbest costs exist in the swarm; thread r owns row r.  The caller synchronises.
// (Measured and dropped: marking single COORDINATES that repeat the one a rank up -- NaN in the table, readers walk up to the value: the walk in the move phase costs every
// batch 7 % and a collapsed linear-slope swarm 2.6x.)
__device__ __forceinline__ void rl_mark_copies(const RlLds& L, int NP, int D, int r)
{
    if (r > 0 && r < NP && L.NC[r] == L.NC[r - 1]) {
        bool same = true;
        for (int d = 0; d < D && same; ++d) same = L.PB[r * D + d] == L.PB[(r - 1) * D + d];
        if (same) L.NCS[r] = 1e300;
    }
}

// A coordinate of an item whose scan met a near-tie (fdr_exact returned true) is settled by its WAVE, lane = candidate: the reference's own rule (rlepso_optimizer.py:
// 100-102: rounded quotients, np.argmin = the first minimal one in PARTICLE order) applied to the scan's winner w and to every candidate that is not CLEARLY worse than w.
//  * dif_k = fma(a_k, b_w, -fl(a_w b_k)) > 2^-49 |a_w b_k| means the exact ratios differ by > 2^-49.2 relative, i.e. the rounded quotient of k is strictly
//    larger than w's: k cannot be the argmin (one pass over the candidates, 64 per step, no division);
//  * candidates bitwise identical to w (same cost, same coordinate: collapsed swarms, coordinates clipped onto a bound) share w's quotient; ranks are (cost, index)-ordered,
//    so the lowest RANK among w and its copies is np.argmin's lowest index -- no division either (the scan's fma may have taken a later copy: its rounding error has a sign);
//  * only if a DIFFERENT candidate is within the band -- a true near-tie, or one clearly better than w, should an earlier near-tie have misled the scan -- the contenders
//    compete with their rounded quotients, lowest particle index among equal ones (ORDER maps ranks to particle indices).
// Exact by construction.  All arguments are wave-uniform, all 64 lanes are active; returns the exemplar's rank (uniform).  Measured on whole episodes of the 24 functions:
// 0.02-0.15 % of the items are flagged, late in an episode up to 0.5 % (Gallagher); every lane redoing its own flagged item would stall the 64 lanes of each wave that holds
// one (6.7 % of the wave-passes at 0.1 %) for a second scan, this form costs the wave ~200 instructions per flagged coordinate.
__device__ __forceinline__ int fdr_settle(const RlLds& L, const int* __restrict__ ORDER, int D, int rk, int d, int nless, int kw, unsigned long long* cnt = nullptr)
{
    const int lane = threadIdx.x & 63;
    const double fi = L.NC[rk];
    const double pp = L.PB[rk * D + d];
    const double aw = L.NC[kw] - fi, bw = fabs(L.PB[kw * D + d] - pp) + 1e-5;
    int first_copy = 0x7fffffff;
    bool others = false;
    for (int k0 = 0; k0 < nless; k0 += 64) {
        const int k = k0 + lane;
        bool band = false, copy = false;
        if (k < nless && k != kw) {
            const double a = L.NC[k] - fi, b = fabs(L.PB[k * D + d] - pp) + 1e-5, pr = aw * b;
            band = __builtin_fma(a, bw, -pr) <= pr * -0x1p-49;      // (pr < 0: the band is 2^-49 |a_w b_k|, the product at hand -- tighter than the scan's flag)
            copy = band && a == aw && b == bw;
        }
        const unsigned long long mc = __builtin_amdgcn_ballot_w64(copy), mo = __builtin_amdgcn_ballot_w64(band && !copy);
        if (mc != 0ull && first_copy == 0x7fffffff) first_copy = k0 + (int)__builtin_ctzll(mc);
        others = others || mo != 0ull;
    }
    int kbest = kw < first_copy ? kw : first_copy;
#ifdef MBX_FDR_COUNT
    if (cnt && lane == 0) atomicAdd(cnt + (others ? 5 : first_copy != 0x7fffffff ? 4 : 6), 1ull);     // experiment: what the flagged coordinates turn out to be
#endif
    if (others) {                                             // wave-uniform, rare: a different candidate within (or below) the band
        double qb = aw / bw;
        int ib = ORDER[kbest];
        for (int k0 = 0; k0 < nless; k0 += 64) {
            const int k = k0 + lane;
            bool cont = false;
            double qk = 0.;
            int ik = 0;
            if (k < nless && k != kw) {
                const double a = L.NC[k] - fi, b = fabs(L.PB[k * D + d] - pp) + 1e-5, pr = aw * b;
                cont = __builtin_fma(a, bw, -pr) <= pr * -0x1p-49 && !(a == aw && b == bw);
                if (cont) { qk = a / b; ik = ORDER[k]; }
            }
            unsigned long long m = __builtin_amdgcn_ballot_w64(cont);
            while (m != 0ull) {
                const int l = (int)__builtin_ctzll(m);
                m &= m - 1ull;
                const double ql = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(qk), l), __builtin_amdgcn_readlane(__double2loint(qk), l));
                const int il = __builtin_amdgcn_readlane(ik, l);
                if (ql < qb || (ql == qb && il < ib)) { qb = ql; ib = il; kbest = k0 + l; }
            }
        }
    }
    return kbest;
}

// ---- FDR exemplar (rlepso_optimizer.py:97-109): argmin_j (pbest_j - pbest_i)/(|p_jd - p_id| + 1e-5), first minimum.
//  * j == i contributes the ratio 0, every particle with a larger pbest a positive one: the minimum is attained
//    among the strictly better particles (negative ratios) if there are any, otherwise it is 0 and np.argmin
//    returns the lowest index with pbest_j == pbest_i.  Only the `nless` better particles are scanned, in
//    ascending-cost order (ORDER / NC), halving the O(NP^2 D) work on average.
//  * ratios are compared by cross-multiplication (denominators >= 1e-5 > 0): a_j/b_j < a*/b* <=> a_j b* < a* b_j.
//    The reference compares ROUNDED quotients; the two orders can differ only where two candidates' quotients are within an ulp or two of each other
//    (tests/test_fdr_ties.py: 49 % / 9 % of crafted pairs 0 / 1 ulp apart resolve differently by cross-multiplication alone).
//  * TIE = true (default): every comparison also yields dif = a_j b* - fl(a* b_j); while |dif| > 8 ulp of the products both forms decide alike (a quotient pair
//    that rounds together or apart differs by <= 1 ulp, i.e. |a_j b* - a* b_j| <= 2^-52 |a* b_j|).  The scan keeps min |dif| (high dwords, one v_min3_f32 per
//    two coordinates and candidate) and reports whether it came within 2^-49 x the largest product an item can form, |a_0| x range; the caller then runs
//    fdr_settle on the item.  Bit-exact exemplars on any input whose pbest positions lie inside the box (|p_jd - p_id| <= ub - lb).
//  * TIE = false (MBX_F_FDR_FAST): the cross-multiplied order alone, two products and a compare; identical candidates (same pbest cost and same coordinate)
//    are adjacent in the (cost, index) order, so the strict `<` keeps the lower index like np.argmin.
// Exact float64 scan for W adjacent coordinates d0 .. d0+W-1 of the particle of pbest-rank rk; the exemplar's rank per coordinate goes to kb; returns the near-tie flag.
// The candidates' costs are read from NCS (= NC but for the rows rl_mark_copies took out of the scan).
// UN: candidates per unrolled group (MBX_FDR_UNROLL = 4 everywhere but config 5's resident kernel: 1024 threads, D = 40: 2 / 4 / 8 -> 1.619 / 1.641 / 1.703 ms per
// generation; the headline kernel: 2 / 3 / 4 / 6 / 8 -> 119.9 / 117.1 / 116.5 / 117.5 / 121.6 us)
// range: an upper bound of |p_jd - p_id| + 1e-5 (ub - lb + 1e-5), used by the near-tie flag only
template <int W, int UN = MBX_FDR_UNROLL, bool TIE = true>
__device__ __forceinline__ bool fdr_exact(const RlLds& L, int D, int rk, int d0, int nless, int kb[W], double range = 0.)
{
#pragma unroll
    for (int q = 0; q < W; ++q) kb[q] = nless;                   // rank of the exemplar when nobody is strictly better
    if (nless <= 0) return false;
    const double fi = L.NC[rk];
    double pp[W], ab[W], bb[W];
    const double a0 = L.NC[0] - fi;
#pragma unroll
    for (int q = 0; q < W; ++q) { pp[q] = L.PB[rk * D + d0 + q]; kb[q] = 0; ab[q] = a0; bb[q] = fabs(L.PB[d0 + q] - pp[q]) + 1e-5; }
    const double* col = L.PB + d0;
    if constexpr (TIE) {
    float tiem = __int_as_float(0x7f000000);
    // one candidate against the running best of the W coordinates: the W differences are complete before the first compare issues (fdr_take_neg2)
    auto step = [&](double au, const double (&x)[W], int k) {
        double b[W], difv[W];
#pragma unroll
        for (int q = 0; q < W; ++q) b[q] = fabs(x[q] - pp[q]) + 1e-5;
#pragma unroll
        for (int q = 0; q < W; ++q) difv[q] = ab[q] * b[q];
#pragma unroll
        for (int q = 0; q < W; ++q) difv[q] = __builtin_fma(au, bb[q], -difv[q]);
        if constexpr (W == 2) fdr_take_neg2(difv[0], difv[1], ab[0], bb[0], kb[0], ab[1], bb[1], kb[1], au, b[0], b[1], k);
        else {
#pragma unroll
            for (int q = 0; q < W; ++q) fdr_take_neg(difv[q], ab[q], bb[q], kb[q], au, b[q], k);
        }
        tiem = fdr_min3_abs(tiem, f64_hi_as_f32(difv[0]), f64_hi_as_f32(difv[W - 1]));
    };
    int k = 1;
    for (; k + UN <= nless; k += UN) {
        double a[UN], x[UN][W];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            a[u] = L.NCS[k + u];
#pragma unroll
            for (int q = 0; q < W; ++q) x[u][q] = col[(k + u) * D + q];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) step(a[u] - fi, x[u], k + u);
    }
    for (; k < nless; ++k) {
        double x[W];
#pragma unroll
        for (int q = 0; q < W; ++q) x[q] = col[k * D + q];
        step(L.NCS[k] - fi, x, k);
    }
    const double thr = fabs(a0) * range * 0x1p-49;               // |a* b_j| <= |a_0| x range for every pair of the item
    return !(tiem > f64_hi_as_f32(thr) * 1.0000005f);
    } else {
    // unrolled by hand (the compiler does not unroll around the inline assembly of fdr_take); the LDS reads of a group are
    // issued before its first comparison
    int k = 1;
    for (; k + UN <= nless; k += UN) {
        double a[UN], x[UN][W];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            a[u] = L.NCS[k + u];
#pragma unroll
            for (int q = 0; q < W; ++q) x[u][q] = col[(k + u) * D + q];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const double au = a[u] - fi;                      // shared by the W coordinates
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const double b = fabs(x[u][q] - pp[q]) + 1e-5;
                fdr_take(au * bb[q], ab[q] * b, ab[q], bb[q], kb[q], au, b, k + u);
            }
        }
    }
    for (; k < nless; ++k) {
        const double a = L.NCS[k] - fi;
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const double b = fabs(col[k * D + q] - pp[q]) + 1e-5;
            fdr_take(a * bb[q], ab[q] * b, ab[q], bb[q], kb[q], a, b, k);
        }
    }
    return false;
    }
}

// The same settlement by the item's OWN lane (both coordinates): what a swarm that has collapsed onto the rounding floor of its costs gets, where most items are flagged and
// one wave per coordinate would take longer than a second pass of everybody (fdr_pass switches at 64 flagged items).
template <int W>
__device__ __noinline__ void fdr_settle_own(const double* __restrict__ NC, const double* __restrict__ PB, const int* __restrict__ ORDER, int D, int rk, int d0, int nless, int* kb)
{
    const double fi = NC[rk];
    double pp[W], aw[W], bw[W];
    int kw[W], first_copy[W];
    bool others[W];
#pragma unroll
    for (int q = 0; q < W; ++q) {
        kw[q] = kb[q];
        pp[q] = PB[rk * D + d0 + q];
        aw[q] = NC[kw[q]] - fi;
        bw[q] = fabs(PB[kw[q] * D + d0 + q] - pp[q]) + 1e-5;
        first_copy[q] = 0x7fffffff; others[q] = false;
    }
    for (int k = 0; k < nless; ++k) {                              // pass 1: who is inside the band of the winner?  copies of it, or somebody else?
        const double a = NC[k] - fi;
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const double b = fabs(PB[k * D + d0 + q] - pp[q]) + 1e-5, pr = aw[q] * b;
            if (__builtin_fma(a, bw[q], -pr) <= pr * -0x1p-49 && k != kw[q]) {
                if (a == aw[q] && b == bw[q]) { if (first_copy[q] == 0x7fffffff) first_copy[q] = k; }
                else others[q] = true;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < W; ++q) {
        kb[q] = kw[q] < first_copy[q] ? kw[q] : first_copy[q];
        if (others[q]) {                                          // pass 2, rare: rounded quotients, lowest particle index among equal ones
            double qb = aw[q] / bw[q];
            int ib = ORDER[kb[q]];
            for (int k = 0; k < nless; ++k) {
                const double a = NC[k] - fi, b = fabs(PB[k * D + d0 + q] - pp[q]) + 1e-5, pr = aw[q] * b;
                if (__builtin_fma(a, bw[q], -pr) <= pr * -0x1p-49 && k != kw[q] && !(a == aw[q] && b == bw[q])) {
                    const double qk = a / b;
                    const int ik = ORDER[k];
                    if (qk < qb || (qk == qb && ik < ib)) { qb = qk; ib = ik; kb[q] = k; }
                }
            }
        }
    }
}

// The FDR pass of a generation: every (rank, coordinate pair) item of the population -> KB[rank * D + d] = rank of its exemplar.  Items are visited in
// pbest-rank order so that the lanes of a wave own particles of similar rank, i.e. similar trip counts; the trip count grows with the rank and wave w of
// every resident workgroup shares one SIMD, so odd passes run backwards (boustrophedon): each wave pairs a cheap slice of ranks with an expensive one.
// Items whose scan met a near-tie are appended to the workgroup's list FL (FLN was zeroed before the ranking barrier) and settled after a barrier: up to 64 of them one wave
// per coordinate (fdr_settle), more than that -- a swarm collapsed onto the rounding floor of its costs -- every lane its own (fdr_settle_own: a second pass of everybody is
// then the shorter one).  Ends with the barrier that publishes KB.  Measured and dropped: (a) every lane redoing its own flagged item right away -- 0.1 % of the items stall
// 6.7 % of the wave-passes for a second scan, +10.4 % per generation against +3.5 % of instructions; (b) the flagged lanes' items settled by their own wave right behind
// the pass (readlane + fdr_settle): the serial ~250-instruction chain then sits inside the scan of a wave the other three wait for, +4.8 % in the driver window against
// +3.9 % for the list, +10.6 % against +8.8 % over whole episodes.
template <int W, int UN, bool TIE, int THREADS>
__device__ __forceinline__ void fdr_pass(const RlLds& L, const int* ORDER, const int* NLESS, int NP, int D, int tid, double range, unsigned long long* cnt = nullptr)
{
    const int DW = D / W, NI = NP * DW;
    const FastDiv fw(DW);
    uint32_t mine = 0;                                            // bit p: this thread's item of pass p was flagged
    for (int base = 0, pass = 0; base < NI; base += THREADS, ++pass) {
        const int lim = base + THREADS < NI ? base + THREADS : NI;
        const int ps = (pass & 1) ? lim - 1 - tid : base + tid;
        if (ps >= base && ps < lim) {
            const int rk = fw.div(ps), d0 = W * (ps - rk * DW);
            int kb[W];
            const bool tie = fdr_exact<W, UN, TIE>(L, D, rk, d0, NLESS[ORDER[rk]], kb, range);
#pragma unroll
            for (int q = 0; q < W; ++q) L.KB[rk * D + d0 + q] = (uint8_t)kb[q];
#ifdef MBX_FDR_COUNT
            if (cnt) { atomicAdd(cnt + 3, 1ull); if (tie) atomicAdd(cnt + 2, 1ull); }       // experiment: how many items does the scan flag? (tools/exp/fdr_flag_rate.py)
#endif
            if (TIE && tie) { L.FL[atomicAdd(L.FLN, 1)] = (uint16_t)ps; mine |= 1u << (pass & 31); }
        }
    }
    __syncthreads();
    if constexpr (TIE) {
        const int n = *L.FLN;                                     // workgroup-uniform
        if (n > 0 && (n <= MBX_FDR_OWN_AT || NI > 32 * THREADS)) {
            for (int c = tid >> 6; c < W * n; c += THREADS / 64) {      // one wave per flagged COORDINATE
                const int ps = L.FL[c / W], rk = fw.div(ps), d = W * (ps - rk * DW) + c % W;
                const int kbest = fdr_settle(L, ORDER, D, rk, d, NLESS[ORDER[rk]], L.KB[rk * D + d], cnt);
                if ((tid & 63) == 0) L.KB[rk * D + d] = (uint8_t)kbest;
            }
            __syncthreads();
        } else if (n > 0) {                                       // a collapsed swarm: every lane settles its own items (passes <= 32: NP x D / W <= 32 x THREADS)
            for (int base = 0, pass = 0; mine != 0u; base += THREADS, ++pass, mine >>= 1) {
                if (!(mine & 1u)) continue;
                const int lim = base + THREADS < NI ? base + THREADS : NI;
                const int ps = (pass & 1) ? lim - 1 - tid : base + tid, rk = fw.div(ps), d0 = W * (ps - rk * DW);
                int kb[W];
#pragma unroll
                for (int q = 0; q < W; ++q) kb[q] = L.KB[rk * D + d0 + q];
                fdr_settle_own<W>(L.NC, L.PB, ORDER, D, rk, d0, NLESS[ORDER[rk]], kb);
#pragma unroll
                for (int q = 0; q < W; ++q) L.KB[rk * D + d0 + q] = (uint8_t)kb[q];
            }
            __syncthreads();
        }
    }
}

// Velocity / position update of W adjacent coordinates of particle i (rlepso_optimizer.py:179-195); FDR exemplars from KB.
// CLPSO (:76-95): one Philox call carries the uniforms of an element PAIR (site ELEM_A, index e >> 1), another one (site TOURN, same
// index) the two tournament pairs; the latter is only evaluated where a tournament is consumed, i.e. where !(u > pci_i).
// RES: the caller keeps positions / velocities on chip (k_rlepso_run): the new values come back in cur / vel instead of going to HBM.
template <int W, bool RES = false>
__device__ __forceinline__ void rl_move(const MoveCtx& c, int i, int d0, double cur[W], double vel[W])
{
    const RlLds& L = c.L;
    const int NP = c.NP, D = c.D;
    const int rk = c.RANK[i], e0 = i * D + d0, es0 = rk * D + d0;
    const double r1 = L.R1[i], r2 = L.R2[i];
    // one Philox call carries the four element-wise uniforms of an element PAIR, 32 bits each (site ELEM_A, index e >> 1: words 0 / 2 = CLPSO
    // uniform / FDR weight of the even element, 1 / 3 of the odd one); a two-coordinate work item starts on an even element (D even, d0 even)
    double uf[W];
    U4 wa{0, 0, 0, 0};
    if (c.tape) {
#pragma unroll
        for (int q = 0; q < W; ++q) uf[q] = c.tape[MBX_RLEPSO_TAPE_FDR(NP, D) + e0 + q];
    } else {
        wa = c.rng.draw((uint32_t)e0 >> 1, MBX_SITE_ELEM_A);
        if (W == 2) { uf[0] = u32d(wa.z); uf[W - 1] = u32d(wa.w); }
        else uf[0] = u32d((e0 & 1) ? wa.w : wa.z);
    }
    const int g = c.fg.div(i);
    double cw = 0., c1 = 0., c2 = 0., c3 = 0., c4 = 0.;
    if (g < c.G) { const double* k = L.COEF + g * 6; cw = k[1]; c1 = k[2]; c2 = k[3]; c3 = k[4]; c4 = k[5]; }
    double ucs[W]; int xrs[W];
    {
        const double pci = c.pci[i];
        bool need = false;
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const int e = e0 + q;
            if (c.tape) ucs[q] = c.tape[MBX_RLEPSO_TAPE_CLPSO(NP, D) + e];
            else ucs[q] = u32d((W == 2 ? q == 1 : (e & 1)) ? wa.y : wa.x);
            xrs[q] = rk;
            need = need || !(ucs[q] > pci);
        }
        if (need) {
            U4 wt{0, 0, 0, 0};
            if (!c.tape) wt = c.rng.draw((uint32_t)e0 >> 1, MBX_SITE_TOURN);
#pragma unroll
            for (int q = 0; q < W; ++q) {
                const int e = e0 + q;
                if (!(ucs[q] > pci)) {
                    int t1, t2;
                    if (c.tape) { t1 = (int)c.tape[MBX_RLEPSO_TAPE_TOURN(NP, D) + 2 * e]; t2 = (int)c.tape[MBX_RLEPSO_TAPE_TOURN(NP, D) + 2 * e + 1]; }
                    else {
                        const bool odd = W == 2 ? q == 1 : (e & 1);
                        t1 = (int)__umulhi(odd ? wt.z : wt.x, (uint32_t)NP); t2 = (int)__umulhi(odd ? wt.w : wt.y, (uint32_t)NP);
                    }
                    xrs[q] = c.RANK[L.PBC[t2] < L.PBC[t1] ? t2 : t1];   // binary tournament on pbest cost, first candidate wins ties
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const int e = e0 + q, d = d0 + q;
        const double pp = L.PB[es0 + q], uc = ucs[q];
        const int xr = xrs[q], kb = L.KB[es0 + q];
        const double ex = L.PB[xr * D + d];                        // CLPSO exemplar (own rank: the particle's own pbest)
        const double v_fdr = uf[q] * (L.PB[kb * D + d] - pp);
        const double v_pbest = r1 * (pp - cur[q]);
        const double v_gbest = r2 * (L.GB[d] - cur[q]);
        const double v_cl = uc * (ex - cur[q]);
        double nv = cw * vel[q] + c1 * v_cl + c2 * v_fdr + c3 * v_gbest + c4 * v_pbest;
        nv = fmin(fmax(nv, -c.vmax), c.vmax);
        double np_ = cur[q] + nv;
        np_ = fmin(fmax(np_, c.lb), c.ub);
        L.X[e] = np_;
        if (RES) { cur[q] = np_; vel[q] = nv; }
        else { c.gPos[e] = np_; c.gVel[e] = nv; }
    }
}

// ------------------------------------------------------------------------------------------------
// PBO_Env.reset(): init_population (rlepso_optimizer.py:39-65)
// ------------------------------------------------------------------------------------------------
// THREADS = 256, or 512 when the geometry leaves room for one workgroup per CU only (see rl_block_threads)
template <int THREADS, int NPC = 0, int DC = 0>
__global__ __launch_bounds__(THREADS) void k_rlepso_reset(BatchParams bp, double* __restrict__ state_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = NPC ? NPC : bp.NP, D = DC ? DC : bp.D, NE = NP * D;
    constexpr bool MAPS = rl_maps_in_lds(NPC, DC);
    constexpr int MD = MAPS ? 0 : DC;
    const DevProblem P = bp.problems[bp.problem_idx[b]];
    const RlLds L = rl_carve(smem, NP, D, MAPS);
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_RLEPSO_ST_SCALARS(NP, D);
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;

    const int episode = (int)sc[MBX_SC_EPISODE] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, (uint32_t)episode};
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb);

    stage_problem<0, MAPS>(P, L.eval());
    for (int e = tid; e < NE; e += MBX_NT) {
        double up, uv;
        if (tape) { up = tape[MBX_RLEPSO_TAPE_REPOS(NP, D) + e]; uv = tape[MBX_RLEPSO_TAPE_REVEL(NP, D) + e]; }
        else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_R); up = u53(w.x, w.y); uv = u53(w.z, w.w); }
        const double x = lb + (ub - lb) * up;
        L.X[e] = x;
        S[MBX_RLEPSO_ST_POS(NP, D) + e] = x;
        S[MBX_RLEPSO_ST_PBPOS(NP, D) + e] = x;
        S[MBX_RLEPSO_ST_VEL(NP, D) + e] = -vmax + (vmax - (-vmax)) * uv;
    }
    __syncthreads();
    population_costs<0, MD>(P, L.eval(), NP, rng, tape ? tape + MBX_RLEPSO_TAPE_NOISE1(NP, D) : nullptr, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B);
    double gb; int gi;
    block_argmin(L.NC, NP, L.RED, gb, gi);
    for (int i = tid; i < NP; i += MBX_NT) {
        const double c = L.NC[i];
        S[MBX_RLEPSO_ST_CCOST(NP, D) + i] = c;
        S[MBX_RLEPSO_ST_PBEST(NP, D) + i] = c;
        S[MBX_RLEPSO_ST_PNI(NP, D) + i] = 0.;
    }
    if (tid < D) S[MBX_RLEPSO_ST_GBPOS(NP, D) + tid] = L.X[gi * D + tid];
    if (tid == 0) {
        sc[MBX_SC_GBEST] = gb; sc[MBX_SC_FES] = NP; sc[MBX_SC_LOG_INDEX] = 1; sc[MBX_SC_COST_LEN] = 1;
        sc[MBX_SC_DONE] = 0; sc[MBX_SC_RETURN] = 0; sc[MBX_SC_GEN] = 0; sc[MBX_SC_EPISODE] = episode;
        sc[MBX_SC_GBEST_IDX] = gi; sc[MBX_SC_REINIT] = 0;
        sc[MBX_NSCALAR] = gb;                                    // cost = [gbest]
        if (state_out) state_out[b] = (double)NP / bp.max_fes;
    }
}

// ------------------------------------------------------------------------------------------------
// PBO_Env.step(action): RLEPSO_Optimizer.update (rlepso_optimizer.py:173-263)
// ------------------------------------------------------------------------------------------------
// Actions come either from `actions` [B, 7G] (PBO_Env.step(action)) or, when `policy_table` is given, are drawn here from
// the actor's (mu, sigma) at the instance's current state (agent.act + env.step in one launch, mbx_rlepso_act_step): row
// fes of the table built by k_gauss_mlp_policy, same Philox draws as mbx_gauss_policy, optionally echoed to actions_out.
// NPC / DC / GC: population, dimension and group count fixed at compile time (0 = taken from the batch).  The reference's own
// geometry (NP = 100 hard-coded in rlepso_optimizer.py:9, n_group = 5, and the D = 10 of its bbob configs) gets an instantiation of
// its own: index arithmetic, the divisions by D and the short D-loops of the evaluator fold into constants (-8 % per generation).
// TIE: the FDR scan flags near-ties and resolves them with the reference's rounded quotients (fdr_exact / fdr_settle): the default; TIE = false is the
// cross-multiplied order alone (MBX_F_FDR_FAST)
template <int THREADS, int NPC = 0, int DC = 0, int GC = 0, bool TIE = true>
__global__ __launch_bounds__(THREADS) MBX_RL_WAVES void k_rlepso_step(BatchParams bp, const float* __restrict__ actions,
                                                          double* __restrict__ state_out, double* __restrict__ reward_out,
                                                          uint8_t* __restrict__ done_out, const float* __restrict__ policy_table,
                                                          int table_rows, float* __restrict__ actions_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = bp.order[blockIdx.x], tid = threadIdx.x;
    const int NP = NPC ? NPC : bp.NP, D = DC ? DC : bp.D, NE = NP * D, G = GC ? GC : bp.n_group;
    double* S = bp.state + (int64_t)b * bp.state_stride;
    double* sc = S + MBX_RLEPSO_ST_SCALARS(NP, D);

    if (sc[MBX_SC_DONE] != 0.) {                                  // finished instances idle (block-uniform exit)
        if (tid == 0) {
            if (state_out) state_out[b] = sc[MBX_SC_FES] / bp.max_fes;
            if (reward_out) reward_out[b] = 0.;
            if (done_out) done_out[b] = 1;
        }
        return;
    }
    MBX_PHASE_BEGIN
    ConstProblem& P = *(ConstProblem*)(bp.problems + bp.problem_idx[b]);   // scalar loads on demand, no SGPR-resident copy
    const RlLds L = rl_carve(smem, NP, D, rl_maps_in_lds(NPC, DC));
    const double* tape = bp.tape ? bp.tape + (int64_t)b * bp.tape_stride : nullptr;
    const float* act = actions + (int64_t)b * (7 * G);

    const int gen = (int)sc[MBX_SC_GEN] + 1;
    const uint64_t seed = bp.seeds[b];
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, (uint32_t)(int)sc[MBX_SC_EPISODE], true};
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb);
    double gbest = sc[MBX_SC_GBEST];
    int gbest_idx = (int)sc[MBX_SC_GBEST_IDX];
    const double pre_gbest = gbest;
    double fes = sc[MBX_SC_FES];

    // ---- stage: pbest positions/costs, c_cost, stagnation counters, gbest position, linear maps
    const double* gPB = S + MBX_RLEPSO_ST_PBPOS(NP, D);
    double* gPos = S + MBX_RLEPSO_ST_POS(NP, D);
    double* gVel = S + MBX_RLEPSO_ST_VEL(NP, D);
    float* ACT = (float*)L.R1;                                    // sampled action; R1 is not written before the ranking barrier
    if (policy_table && tid < 7 * G) {
        const int A = 7 * G;
        int row = (int)fes;
        row = row < table_rows ? row : table_rows - 1;
        const float* ms = policy_table + (int64_t)row * 2 * A;
        const float a = sample_action(rng, tid, ms[tid], ms[A + tid], MBX_POLICY_RLEPSO);
        ACT[tid] = a;
        if (actions_out) actions_out[(int64_t)b * A + tid] = a;
    }
    for (int i = tid; i < NP; i += MBX_NT) {
        L.PBC[i] = S[MBX_RLEPSO_ST_PBEST(NP, D) + i];
        L.RANK[i] = 0; L.MASK[i] = 0;
        L.IMPR[i] = i;                                             // ORDER stays a valid index table even if a NaN cost breaks the ranking
        L.PNI[i] = S[MBX_RLEPSO_ST_PNI(NP, D) + i];
    }
    if (tid < D) L.GB[tid] = S[MBX_RLEPSO_ST_GBPOS(NP, D) + tid];
    if (tid == 0) *L.FLN = 0;                                     // list of the FDR scan's near-tie items (fdr_pass): empty before the ranking barrier
    stage_problem<eval_dc(DC), eval_md(DC) == 0>(P, L.eval());
    // __get_coe (:112-132): float32 arithmetic (numpy >= 2 keeps float32 for scalar*python-float), group g
    // reads actions[g*n_group : g*n_group+7]
    auto get_coe = [&](const float* a) {
        const float cm = a[0] * 0.01f;
        const float wv = a[1] * 0.8f + 0.1f;
        float den = a[3] + a[4];
        den = den + a[5]; den = den + a[6]; den = den + 1e-5f;
        float scale = 1.f / den;
        scale = scale * a[2]; scale = scale * 8.f;
        double* c = L.COEF + tid * 6;
        c[0] = (double)cm; c[1] = (double)wv;
        c[2] = (double)(scale * a[3]); c[3] = (double)(scale * a[4]);
        c[4] = (double)(scale * a[5]); c[5] = (double)(scale * a[6]);
    };
    if (!policy_table && tid < G) get_coe(act + tid * G);
    __syncthreads();
    // fused policy: the sampled action became visible with the barrier above; the coefficients are published by the ranking
    // barrier below, before their first use
    if (policy_table && tid < G) get_coe(ACT + tid * G);
    MBX_PHASE(0);                                                 // HBM -> LDS staging
    const int per_group = NP / G;
    const FastDiv fd(D), fg(per_group);
    // ---- rank the particles by (pbest cost, index); all 256 threads take part: thread (i, part) counts, over a slice of j,
    // the particles that are strictly better (nless) and those that are not worse (nle).  rank = nless unless another
    // particle has exactly the same pbest cost (nle - nless > 1: rare), in which case the equal ones are ordered by index.
    int* ORDER = L.IMPR;          // free until the first commit
    int* NLESS = L.MASK;
    int* RANK = L.RANK;           // doubles as the nle accumulator
    {
        const int parts = MBX_NT / NP > 0 ? MBX_NT / NP : 1;
        for (int w = tid; w < parts * NP; w += MBX_NT) {
            const int part = w / NP, i = w - part * NP;
            const int j0 = part * NP / parts, j1 = (part + 1) * NP / parts;
            const double fi = L.PBC[i];
            int nle = 0, nless = 0;
#ifndef MBX_ABLATE_RANK
#pragma unroll 4
            for (int j = j0; j < j1; ++j) {
                const double fj = L.PBC[j];
                nless += fj < fi;
                nle += fj <= fi;
            }
#else
            if (part == 0) { nle = i + 1; nless = i; }
#endif
            atomicAdd(&RANK[i], nle); atomicAdd(&NLESS[i], nless);
        }
    }
    __syncthreads();
    int equal_costs = 0;
    for (int i = tid; i < NP; i += MBX_NT) {                    // per-particle quantities
        const int g = fg.div(i);
        L.CMUT[i] = g < G ? L.COEF[g * 6] * L.PNI[i] : 0.;        // uses per_no_improve BEFORE this step's update (:120)
        if (tape) { L.R1[i] = tape[MBX_RLEPSO_TAPE_RAND1(NP, D) + i]; L.R2[i] = tape[MBX_RLEPSO_TAPE_RAND2(NP, D) + i]; }
        else { const U4 w = rng.draw((uint32_t)i, MBX_SITE_PART); L.R1[i] = u53(w.x, w.y); L.R2[i] = u53(w.z, w.w); }
        const double fi = L.PBC[i];
        int rank = NLESS[i];
        if (RANK[i] - rank > 1) {
            equal_costs = 1;
            for (int j = 0; j < i; ++j) rank += L.PBC[j] == fi;
        }
        RANK[i] = rank;                                            // thread i is the only reader / writer of entry i here
        ORDER[rank] = i; L.NC[rank] = fi; L.NCS[rank] = fi;        // NC: pbest costs in ascending order (free until eval); NCS: the column the FDR scan reads
    }
    equal_costs = __syncthreads_or(equal_costs);
    MBX_PHASE(1);                                                 // ranking + per-particle draws
    // pbest positions are staged in RANK order (row r = particle ORDER[r]): the FDR scan below then walks LDS linearly
    for (int e = tid; e < NE; e += MBX_NT) { const int i = fd.div(e), d = e - i * D; L.PB[RANK[i] * D + d] = gPB[e]; }
    __syncthreads();
    if (equal_costs) {                                            // workgroup-uniform: rows that are copies of the row one rank up leave the FDR scan (rl_mark_copies)
        for (int r = tid; r < NP; r += MBX_NT) rl_mark_copies(L, NP, D, r);
        __syncthreads();
    }
    MBX_PHASE(2);                                                 // pbest rows -> LDS in rank order

    // ---- FDR exemplars -> KB[rank * D + d]  (a pass of its own: the scan's registers are dead before the velocity update starts, which
    // keeps the kernel at 80 VGPRs without scratch; fused into the update it needed 96 and spilled)
#ifndef MBX_ABLATE_FDR
    if ((D & 1) == 0) fdr_pass<2, MBX_FDR_UNROLL, TIE, THREADS>(L, ORDER, NLESS, NP, D, tid, ub - lb + 1e-5);
    else fdr_pass<1, MBX_FDR_UNROLL, TIE, THREADS>(L, ORDER, NLESS, NP, D, tid, ub - lb + 1e-5);
#else
    for (int e = tid; e < NE; e += MBX_NT) L.KB[e] = 0;
    __syncthreads();
#endif

    // ---- velocity / position update (:179-195).  A work item is W adjacent dimensions of one particle (W = 2 when D is even).
    const MoveCtx mc{L, bp.pci, tape, rng, gPos, gVel, ORDER, NLESS, RANK, NP, D, G, lb, ub, vmax, fg};
    if ((D & 1) == 0) {
        const int HD = D >> 1, NI = NP * HD;
        const FastDiv fh(HD);
        for (int it = tid; it < NI; it += MBX_NT) {
            const int i = fh.div(it);
            const double2 c2 = *(const double2*)(gPos + 2 * it), v2 = *(const double2*)(gVel + 2 * it);
            double cur[2] = {c2.x, c2.y}, vel[2] = {v2.x, v2.y};
            rl_move<2>(mc, i, 2 * (it - i * HD), cur, vel);
        }
    } else {
        for (int e = tid; e < NE; e += MBX_NT) {
            const int i = fd.div(e);
            double cur[1] = {gPos[e]}, vel[1] = {gVel[e]};
            rl_move<1>(mc, i, e - i * D, cur, vel);
        }
    }
    __syncthreads();
    MBX_PHASE(3);                                                 // move (draws, exemplars, FDR scan, velocity)

    // ---- evaluate, update pbest/gbest and stagnation counters (:198-233)
    population_costs<eval_dc(DC), eval_md(DC)>(P, L.eval(), NP, rng, tape ? tape + MBX_RLEPSO_TAPE_NOISE0(NP, D) : nullptr, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B);
    MBX_PHASE(4);                                                 // evaluation
    fes += NP;
    rl_commit(L, NP, D, true, gbest, gbest_idx, S + MBX_RLEPSO_ST_PBPOS(NP, D), S + MBX_RLEPSO_ST_CCOST(NP, D));
    MBX_PHASE(5);                                                 // commit

    // ---- re-initialisation (:238-239, 134-168): P(i) = u < c_mutation_i * 0.01 * per_no_improve_i
    int mine = 0;
    for (int i = tid; i < NP; i += MBX_NT) {
        double u;
        if (tape) u = tape[MBX_RLEPSO_TAPE_REINIT(NP, D) + i];
        else { const U4 w = rng.draw((uint32_t)i, MBX_SITE_REINIT); u = u53(w.x, w.y); }
        const int m = u < L.CMUT[i] * 0.01 * L.PNI[i];
        L.MASK[i] = m;
        mine += m;
    }
    const int n_reinit = __syncthreads_count(mine);               // NP <= 256: at most one particle per thread
    MBX_PHASE(6);                                                 // re-init draw
    if (n_reinit > 0) {
        for (int e = tid; e < NE; e += MBX_NT) {
            const int i = fd.div(e);
            if (L.MASK[i]) {
                double up, uv;
                if (tape) { up = tape[MBX_RLEPSO_TAPE_REPOS(NP, D) + e]; uv = tape[MBX_RLEPSO_TAPE_REVEL(NP, D) + e]; }
                else { const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_R); up = u53(w.x, w.y); uv = u53(w.z, w.w); }
                const double x = lb + (ub - lb) * up;
                L.X[e] = x;
                gPos[e] = x;
                gVel[e] = -vmax + (vmax - (-vmax)) * uv;
            }
        }
        __syncthreads();
        // the whole population is re-evaluated but only the re-initialised particles are billed (:141-143)
        population_costs<eval_dc(DC), eval_md(DC)>(P, L.eval(), NP, rng, tape ? tape + MBX_RLEPSO_TAPE_NOISE1(NP, D) : nullptr, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B);
        fes += n_reinit;
        rl_commit(L, NP, D, false, gbest, gbest_idx, S + MBX_RLEPSO_ST_PBPOS(NP, D), S + MBX_RLEPSO_ST_CCOST(NP, D));
    }

    MBX_PHASE(7);                                                 // re-init (move, evaluation, commit)
    // ---- write back what changed
    for (int i = tid; i < NP; i += MBX_NT) {
        S[MBX_RLEPSO_ST_PBEST(NP, D) + i] = L.PBC[i];
        S[MBX_RLEPSO_ST_PNI(NP, D) + i] = L.PNI[i];
    }
    if (gbest < pre_gbest && tid < D) S[MBX_RLEPSO_ST_GBPOS(NP, D) + tid] = L.GB[tid];

    // ---- logging, termination, reward, next state (:241-261)
    if (tid == 0) {
        int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
        double* cost = sc + MBX_NSCALAR;
        const bool done = log_and_terminate(bp, P, fes, gbest, log_index, cost_len, cost);
        const double reward = gbest < pre_gbest ? 1. : -1.;
        sc[MBX_SC_GBEST] = gbest; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
        sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_RETURN] += reward; sc[MBX_SC_GEN] = gen;
        sc[MBX_SC_GBEST_IDX] = gbest_idx; sc[MBX_SC_REINIT] = n_reinit > 0 ? 1. : 0.;
        if (state_out) state_out[b] = fes / bp.max_fes;
        if (reward_out) reward_out[b] = reward;
        if (done_out) done_out[b] = done ? 1 : 0;
    }
    MBX_PHASE(8);                                                 // write-back, logging
}

// ------------------------------------------------------------------------------------------------
// Agent.rollout_episode's loop (src/agent/rlepso_agent.py:294-303), up to n_gens generations per launch, state ON CHIP in between
// ------------------------------------------------------------------------------------------------
// k_rlepso_step streams an instance's 26.5 KB state block in and out once per generation because a policy outside the kernel needs
// the state between generations.  With the policy inside (actor table, as in mbx_rlepso_act_step) nothing leaves the workgroup between
// two generations, so this kernel loads the block ONCE, runs up to n_gens generations and stores it ONCE: positions stay in LDS (X is
// only read by the evaluator), velocities, pbest positions and c_cost stay in registers of the thread that owns the element / particle,
// the per-particle arrays and the problem constants stay in LDS.  Every generation performs exactly the arithmetic of k_rlepso_step with
// the same Philox counters (gen is part of the counter), so n_gens launches of mbx_rlepso_act_step and one launch of this kernel give
// bit-identical states (tests/test_gpu_rlepso.py).  A workgroup leaves as soon as its instance is done and the hardware dispatcher hands
// its slot to the next workgroup: no lock-step idling, no refill queue needed.
// Compile-time geometry only (even D, NP <= THREADS: particle i's scalars live in thread i); other geometries are stepped by the host
// loop in mbx_rlepso_rollout.
struct RunOut {
    float* traj_actions;         // [n_gens][B][7 G] sampled actions, or nullptr
    double* traj_state;          // [n_gens][B] state after the generation (fes / maxFEs), or nullptr
    double* traj_reward;         // [n_gens][B]
    uint8_t* traj_done;          // [n_gens][B]
    double* state_out;           // [B] state after the last executed generation, or nullptr
    double* reward_out;          // [B] SUM of the rewards of the executed generations, or nullptr
    uint8_t* done_out;           // [B]
};

#ifndef MBX_RUN_WAVES
#define MBX_RUN_WAVES MBX_RL_WAVES
#endif
// The body of k_rlepso_run.  KIND != 0: the instance's function kind, known at compile time -- the evaluator (twice in the body: update and re-initialisation) holds that
// kind's code only.
// ARGS: a callable that returns (a reference to) the kernel's argument block -- the kernel's own parameters, or, out of line, the kernarg segment behind a pointer
// that is re-materialised at every use (each field access a fresh scalar load next to its use: what the compiler does with a kernel's parameters by itself, and
// what it cannot do with values a callee loaded once -- those stay in SGPRs across the whole body or are spilled).
template <int THREADS, int NPC, int DC, int GC, int KIND, int NOISE, bool TIE, class ARGS>
__device__ __forceinline__ void rl_run_body(ARGS ar)
{
    const int n_gens = ar().n_gens;
    static_assert(NPC > 0 && DC > 0 && GC > 0 && (DC & 1) == 0 && NPC <= THREADS, "k_rlepso_run: compile-time geometry, even D, NP <= THREADS");
    constexpr int NP = NPC, D = DC, G = GC, HD = D / 2, NI = NP * HD, IT = (NI + THREADS - 1) / THREADS, A = 7 * G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = ar().bp.order[blockIdx.x], tid0 = threadIdx.x, tid = tid0;
    double* S = ar().bp.state + (int64_t)b * ar().bp.state_stride;
    double* sc = S + MBX_RLEPSO_ST_SCALARS(NP, D);
    const int64_t B = ar().bp.B;

    if (sc[MBX_SC_DONE] != 0.) {                                  // finished before the launch: what k_rlepso_step reports for it, n_gens times
        const double st = sc[MBX_SC_FES] / ar().bp.max_fes;
        for (int g = tid; g < n_gens; g += THREADS) {
            if (ar().out.traj_state) ar().out.traj_state[g * B + b] = st;
            if (ar().out.traj_reward) ar().out.traj_reward[g * B + b] = 0.;
            if (ar().out.traj_done) ar().out.traj_done[g * B + b] = 1;
        }
        if (tid == 0) {
            if (ar().out.state_out) ar().out.state_out[b] = st;
            if (ar().out.reward_out) ar().out.reward_out[b] = 0.;
            if (ar().out.done_out) ar().out.done_out[b] = 1;
        }
        return;
    }
    // launch clock (diagnostics, off unless a slot pair is attached): s_memtime counters of different XCDs / CUs are offset against each other, so every workgroup
    // measures its OWN lifetime in both time bases and adds the two differences to the launch's sums
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    const bool clocked = ar().bp.clk != nullptr && tid0 == 0;
    if (clocked) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    ConstProblem& P = *(ConstProblem*)(ar().bp.problems + ar().bp.problem_idx[b]);
    const RlLds L = rl_carve(smem, NP, D, rl_maps_in_lds(NPC, DC));
    int gen = (int)sc[MBX_SC_GEN];
    const uint64_t seed = ar().bp.seeds[b];
    const uint32_t episode = (uint32_t)(int)sc[MBX_SC_EPISODE];
    const double lb = P.lb, ub = P.ub, vmax = 0.1 * (ub - lb);
    double gbest = sc[MBX_SC_GBEST];
    int gbest_idx = (int)sc[MBX_SC_GBEST_IDX];
    double fes = sc[MBX_SC_FES];
    int log_index = (int)sc[MBX_SC_LOG_INDEX], cost_len = (int)sc[MBX_SC_COST_LEN];
    double* cost = sc + MBX_NSCALAR;
    double ret = 0.;
    int n_reinit = 0;

    // ---- load the state block once: positions -> X (LDS), velocities / pbest positions -> registers of the element's owner
    // (work item it = tid + j THREADS owns the adjacent coordinates 2 it, 2 it + 1 of particle it / HD), c_cost -> register of thread i
    double* gPos = S + MBX_RLEPSO_ST_POS(NP, D);
    double* gVel = S + MBX_RLEPSO_ST_VEL(NP, D);
    double* gPB = S + MBX_RLEPSO_ST_PBPOS(NP, D);
    double vel[IT][2], pbp[IT][2], cc = 0.;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int it = tid + j * THREADS;
        vel[j][0] = vel[j][1] = pbp[j][0] = pbp[j][1] = 0.;
        if (it < NI) {
            const double2 v2 = *(const double2*)(gVel + 2 * it), p2 = *(const double2*)(gPB + 2 * it);
            vel[j][0] = v2.x; vel[j][1] = v2.y; pbp[j][0] = p2.x; pbp[j][1] = p2.y;
            *(double2*)(L.X + 2 * it) = *(const double2*)(gPos + 2 * it);
        }
    }
    if (tid < NP) {
        L.PBC[tid] = S[MBX_RLEPSO_ST_PBEST(NP, D) + tid];
        L.PNI[tid] = S[MBX_RLEPSO_ST_PNI(NP, D) + tid];
        cc = S[MBX_RLEPSO_ST_CCOST(NP, D) + tid];
    }
    if (tid < D) L.GB[tid] = S[MBX_RLEPSO_ST_GBPOS(NP, D) + tid];
    stage_problem<eval_dc(DC), eval_md(DC) == 0, ConstProblem>(P, L.eval());
    constexpr int per_group = NP / G;
    const FastDiv fg(per_group), fh(HD);
    int* ORDER = L.IMPR;          // free until the first commit of a generation
    int* NLESS = L.MASK;
    int* RANK = L.RANK;
    float* ACT = (float*)L.R1;    // sampled action; R1 is not written before the ranking barrier
    int* EQC = (int*)(L.RED + 14);  // equal-cost flag of the ranking (block_argmin uses RED[0..1] only)

    // pbest / gbest bookkeeping of update() and __reinit() (rl_commit with c_cost and the pbest positions in registers)
    auto commit = [&](bool stagnation, int tid) {
        if (tid < NP) {
            const double nc = L.NC[tid];
            if (stagnation) L.PNI[tid] = nc < cc ? 0. : L.PNI[tid] + 1;
            const int impr = nc < L.PBC[tid];
            if (impr) L.PBC[tid] = nc;
            L.IMPR[tid] = impr;
            cc = nc;
        }
        double cbv; int cb;
        block_argmin(L.NC, NP, L.RED, cbv, cb);         // contains the barriers that publish IMPR
        const bool better = cbv < gbest;
        if (better) { gbest = cbv; gbest_idx = cb; }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int it = tid + j * THREADS;
            if (it < NI && L.IMPR[fh.div(it)]) { const double2 x2 = *(const double2*)(L.X + 2 * it); pbp[j][0] = x2.x; pbp[j][1] = x2.y; }
        }
        if (better && tid < D) L.GB[tid] = L.X[cb * D + tid];
        __syncthreads();
    };

    bool done = false;
    int g = 0;
    __syncthreads();
    for (; g < n_gens && !done; ++g) {
        // the thread index is re-materialised every generation: otherwise the compiler hoists all the index arithmetic of the body out of
        // the loop and spills it (528 B of scratch per thread), which the one-generation kernel never has to carry
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        gen += 1;
        const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)gen, episode, true};
        const double pre_gbest = gbest;
        // ---- agent.act: the action of this generation from row fes of the actor table (same draws as mbx_gauss_policy)
        if (tid < A) {
            int row = (int)fes;
            row = row < ar().table_rows ? row : ar().table_rows - 1;
            const float* ms = ar().policy_table + (int64_t)row * 2 * A;
            const float a = sample_action(rng, tid, ms[tid], ms[A + tid], MBX_POLICY_RLEPSO);
            ACT[tid] = a;
            if (ar().out.traj_actions) ar().out.traj_actions[((int64_t)g * B + b) * A + tid] = a;
        }
        if (tid < NP) { NLESS[tid] = 0; ORDER[tid] = tid; }
        if (tid == 0) { *EQC = 0; *L.FLN = 0; }
        __syncthreads();
        if (tid < G) {                                            // __get_coe (:112-132), float32 like k_rlepso_step
            const float* a = ACT + tid * G;
            const float cm = a[0] * 0.01f;
            const float wv = a[1] * 0.8f + 0.1f;
            float den = a[3] + a[4];
            den = den + a[5]; den = den + a[6]; den = den + 1e-5f;
            float scale = 1.f / den;
            scale = scale * a[2]; scale = scale * 8.f;
            double* c = L.COEF + tid * 6;
            c[0] = (double)cm; c[1] = (double)wv;
            c[2] = (double)(scale * a[3]); c[3] = (double)(scale * a[4]);
            c[4] = (double)(scale * a[5]); c[5] = (double)(scale * a[6]);
        }
        // ---- rank the particles by (pbest cost, index).  k_rlepso_step counts, per pair, both `<` and `<=` so that equal costs can be ordered
        // by index; here only the strictly better particles are counted (ONE compare per pair: half the ranking's instructions) and the rank is
        // that count.  Two particles with exactly the same pbest cost then claim the same slot of ORDER: the one that lost the slot sees it
        // (ORDER[rank] != i), raises the workgroup's equal-cost flag, and the workgroup -- uniformly -- redoes the per-particle step with the index
        // tie-break and restages the rows.  Same ranks as k_rlepso_step in every case; the slow path only runs for instances that really hold
        // equal costs (collapsed swarms on F5 / F7 plateaus).
        {
            constexpr int parts = THREADS / NP > 0 ? THREADS / NP : 1;
            for (int w = tid; w < parts * NP; w += THREADS) {
                const int part = w / NP, i = w - part * NP;
                const int j0 = part * NP / parts, j1 = (part + 1) * NP / parts;
                const double fi = L.PBC[i];
                int nless = 0;
#ifndef MBX_ABLATE_RANK
#pragma unroll 5
                for (int j = j0; j < j1; ++j) nless += L.PBC[j] < fi;
#else
                if (part == 0) nless = i;
#endif
                atomicAdd(&NLESS[i], nless);
            }
        }
        __syncthreads();
        if (tid < NP) {
            const int i = tid, gi = fg.div(i);
            L.CMUT[i] = gi < G ? L.COEF[gi * 6] * L.PNI[i] : 0.;
            const U4 w = rng.draw((uint32_t)i, MBX_SITE_PART);
            L.R1[i] = u53(w.x, w.y); L.R2[i] = u53(w.z, w.w);
            const int rank = NLESS[i];
            RANK[i] = rank;
            ORDER[rank] = i; L.NC[rank] = L.PBC[i]; L.NCS[rank] = L.PBC[i];
        }
        __syncthreads();
        if (tid < NP && ORDER[RANK[tid]] != tid) *EQC = 1;
        // ---- pbest positions -> LDS in rank order, from the owners' registers
        auto stage_pbest = [&](int tid) {
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int it = tid + j * THREADS;
                if (it < NI) { const int i = fh.div(it), d0 = 2 * (it - i * HD); *(double2*)(L.PB + RANK[i] * D + d0) = double2{pbp[j][0], pbp[j][1]}; }
            }
        };
        stage_pbest(tid);
        __syncthreads();
        if (*EQC) {                                               // workgroup-uniform: equal pbest costs exist, order them by index
            int rank = 0;
            if (tid < NP) {
                const double fi = L.PBC[tid];
                rank = NLESS[tid];
                for (int j = 0; j < tid; ++j) rank += L.PBC[j] == fi;
            }
            __syncthreads();                                      // every thread has read the flag and the fast path's tables
            if (tid < NP) { RANK[tid] = rank; ORDER[rank] = tid; L.NC[rank] = L.PBC[tid]; L.NCS[rank] = L.PBC[tid]; }
            __syncthreads();
            stage_pbest(tid);
            __syncthreads();
            rl_mark_copies(L, NP, D, tid);                        // rows that are copies of the row one rank up leave the FDR scan
            __syncthreads();
        }
        // ---- FDR exemplars -> KB (see k_rlepso_step)
#ifdef MBX_ABLATE_FDR
        for (int e = tid; e < NP * D; e += THREADS) L.KB[e] = 0;
        __syncthreads();
#else
        fdr_pass<2, (DC == 40 ? 2 : MBX_FDR_UNROLL), TIE, THREADS>(L, ORDER, NLESS, NP, D, tid, ub - lb + 1e-5, ar().bp.clk);      // (ends with the barrier that publishes KB)
#endif
        // (Measured and dropped, round 3: dealing config 5's half-empty last pass -- 512 items of the highest ranks on 1024 threads -- as single
        // coordinates, one per thread, so that every wave scans: 1.80 -> 1.94 ms per generation; the one-coordinate scan repeats the cost difference
        // and the LDS reads per candidate, which costs more than the idle waves did.)
        // ---- velocity / position update (:179-195): new position -> X, new velocity -> register
        {
            const MoveCtx mc{L, ar().bp.pci, nullptr, rng, nullptr, nullptr, ORDER, NLESS, RANK, NP, D, G, lb, ub, vmax, fg};
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int it = tid + j * THREADS;
                if (it < NI) {
                    const int i = fh.div(it);
                    const double2 c2 = *(const double2*)(L.X + 2 * it);
                    double cur[2] = {c2.x, c2.y};
                    rl_move<2, true>(mc, i, 2 * (it - i * HD), cur, vel[j]);
                }
            }
        }
        __syncthreads();
        // ---- evaluate, update pbest / gbest and the stagnation counters (:198-233)
        population_costs<eval_dc(DC), eval_md(DC), ConstProblem, rl_run_matvec_chunk(DC), KIND, NOISE>(P, L.eval(), NP, rng, nullptr, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B);
        fes += NP;
        commit(true, tid);
        // ---- re-initialisation (:238-239, 134-168)
        int mine = 0;
        if (tid < NP) {
            const U4 w = rng.draw((uint32_t)tid, MBX_SITE_REINIT);
            const double u = u53(w.x, w.y);
            mine = u < L.CMUT[tid] * 0.01 * L.PNI[tid];
            L.MASK[tid] = mine;
        }
        n_reinit = __syncthreads_count(mine);
        if (n_reinit > 0) {
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int it = tid + j * THREADS;
                if (it < NI && L.MASK[fh.div(it)]) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = 2 * it + q;
                        const U4 w = rng.draw((uint32_t)e, MBX_SITE_ELEM_R);
                        const double up = u53(w.x, w.y), uv = u53(w.z, w.w);
                        L.X[e] = lb + (ub - lb) * up;
                        vel[j][q] = -vmax + (vmax - (-vmax)) * uv;
                    }
                }
            }
            __syncthreads();
            population_costs<eval_dc(DC), eval_md(DC), ConstProblem, rl_run_matvec_chunk(DC), KIND, NOISE>(P, L.eval(), NP, rng, nullptr, MBX_SITE_NOISE1_A, MBX_SITE_NOISE1_B);
            fes += n_reinit;
            commit(false, tid);
        }
        // ---- logging, termination, reward (:241-261); every thread keeps the (block-uniform) counters, thread 0 writes the curve
        if (fes >= (double)log_index * ar().bp.log_interval) { log_index += 1; if (tid == 0) cost[cost_len] = gbest; cost_len += 1; }
        done = fes >= ar().bp.max_fes;
        if (!isnan(P.optimum) && ar().bp.early_stop) done = done || gbest <= 1e-8;
        if (done) {
            if (cost_len >= ar().bp.n_logpoint + 1) { if (tid == 0) cost[cost_len - 1] = gbest; }
            else { if (tid == 0) cost[cost_len] = gbest; cost_len += 1; }
        }
        const double reward = gbest < pre_gbest ? 1. : -1.;
        ret += reward;
        if (tid == 0) {
            if (ar().out.traj_state) ar().out.traj_state[g * B + b] = fes / ar().bp.max_fes;
            if (ar().out.traj_reward) ar().out.traj_reward[g * B + b] = reward;
            if (ar().out.traj_done) ar().out.traj_done[g * B + b] = done ? 1 : 0;
        }
    }
    // ---- store the state block once
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int it = tid + j * THREADS;
        if (it < NI) {
            *(double2*)(gPos + 2 * it) = *(const double2*)(L.X + 2 * it);
            *(double2*)(gVel + 2 * it) = double2{vel[j][0], vel[j][1]};
            *(double2*)(gPB + 2 * it) = double2{pbp[j][0], pbp[j][1]};
        }
    }
    if (tid < NP) {
        S[MBX_RLEPSO_ST_PBEST(NP, D) + tid] = L.PBC[tid];
        S[MBX_RLEPSO_ST_PNI(NP, D) + tid] = L.PNI[tid];
        S[MBX_RLEPSO_ST_CCOST(NP, D) + tid] = cc;
    }
    if (tid < D) S[MBX_RLEPSO_ST_GBPOS(NP, D) + tid] = L.GB[tid];
    const double st = fes / ar().bp.max_fes;
    for (int t = g + tid; t < n_gens; t += THREADS) {            // generations after the instance finished
        if (ar().out.traj_state) ar().out.traj_state[t * B + b] = st;
        if (ar().out.traj_reward) ar().out.traj_reward[t * B + b] = 0.;
        if (ar().out.traj_done) ar().out.traj_done[t * B + b] = 1;
    }
    if (tid == 0) {
        sc[MBX_SC_GBEST] = gbest; sc[MBX_SC_FES] = fes; sc[MBX_SC_LOG_INDEX] = log_index; sc[MBX_SC_COST_LEN] = cost_len;
        sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_RETURN] += ret; sc[MBX_SC_GEN] = gen;
        sc[MBX_SC_GBEST_IDX] = gbest_idx; sc[MBX_SC_REINIT] = n_reinit > 0 ? 1. : 0.;
        if (ar().out.state_out) ar().out.state_out[b] = st;
        if (ar().out.reward_out) ar().out.reward_out[b] = ret;
        if (ar().out.done_out) ar().out.done_out[b] = done ? 1 : 0;
        if (clocked) {
            atomicAdd(ar().bp.clk, (unsigned long long)__builtin_amdgcn_s_memtime() - clk_c0);
            atomicAdd(ar().bp.clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_r0);
        }
    }
}


// The kernel's arguments as the kernarg segment lays them out (a callee cannot name the kernel's parameters; it is handed the segment's address)
struct RlRunArgs {
    BatchParams bp;
    const float* policy_table;
    int32_t table_rows, n_gens;
    RunOut out;
};

// rl_run_body out of line, one instantiation per function kind: register-allocated and scheduled without the other kinds' code (a build of the any-kind kernel with every
// kind but one compiled out of the evaluator ran a one-function batch 8-12 % faster: docs/EXPERIMENTS.md).  In a callee s[8:9] is the IMPLICIT argument pointer, so the
// address of the kernel's argument block is an argument; every field access goes through that address (see rl_run_body).
template <int THREADS, int NPC, int DC, int GC, int KIND, int NOISE, bool TIE>
__device__ __noinline__ void rl_run_body_of_kind(uint32_t karg_lo_, uint32_t karg_hi_)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)karg_lo_), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)karg_hi_);
    typedef const RlRunArgs __attribute__((address_space(4))) CArgs;
    auto ar = [&]() -> CArgs& {
        uintptr_t p = (uintptr_t)(((uint64_t)hi << 32) | lo);
        asm volatile("" : "+s"(p));
        return *(CArgs*)p;
    };
    rl_run_body<THREADS, NPC, DC, GC, KIND, NOISE, TIE>(ar);
}

// per-kind bodies for the geometries of configs 1 / 2 (D = 10) and 5 (D = 40): the 24 noise-free BBOB kinds with the noise models compiled out (another
// +2 % on the headline) + the eight kinds of the noisy suite with them (rl_run_kind_ok); the other geometries run the any-kind body inline.  Same-box A/B against the any-kind kernel: D = 10 117.6 -> 113.0 us per generation, D = 40 1.602 -> 1.585 ms.
__host__ __device__ constexpr bool rl_run_per_kind(int DC) { return DC == 10 || DC == 40; }
// a body exists for: the 24 noise-free BBOB kinds; with noise, the eight kinds of the noisy suite (problem/bbob.py: _NOISY)
__host__ __device__ constexpr bool rl_run_kind_ok(int kind, int noise_kind)
{
    return noise_kind == MBX_NOISE_NONE ? (kind >= 1 && kind <= 24)
                                        : (kind == 1 || kind == 7 || kind == 8 || kind == 10 || kind == 14 || kind == 17 || kind == 19 || kind == 21);
}

// TIE: see k_rlepso_step
template <int THREADS, int NPC, int DC, int GC, bool TIE = true>
__global__ __launch_bounds__(THREADS) MBX_RUN_WAVES void k_rlepso_run(BatchParams bp, const float* __restrict__ policy_table, int table_rows,
                                                                     int n_gens, RunOut out)
{
    if constexpr (rl_run_per_kind(DC)) {
        static_assert(sizeof(RlRunArgs) == sizeof(BatchParams) + 8 + 8 + sizeof(RunOut), "RlRunArgs mirrors the kernel's parameter list");
        const uint64_t karg = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        const uint32_t klo = (uint32_t)karg, khi = (uint32_t)(karg >> 32);
        const DevProblem* pr = bp.problems + bp.problem_idx[bp.order[blockIdx.x]];
        const int kind = __builtin_amdgcn_readfirstlane(pr->kind);
        const bool noisy = __builtin_amdgcn_readfirstlane(pr->noise_kind) != MBX_NOISE_NONE;
#define MBX_RL_KIND(K) case K: rl_run_body_of_kind<THREADS, NPC, DC, GC, K, 0, TIE>(klo, khi); break;
#define MBX_RL_NOISY(K) case K: rl_run_body_of_kind<THREADS, NPC, DC, GC, K, 1, TIE>(klo, khi); break;
        if (noisy) {
            switch (kind) {
            MBX_RL_NOISY(1) MBX_RL_NOISY(7) MBX_RL_NOISY(8) MBX_RL_NOISY(10) MBX_RL_NOISY(14) MBX_RL_NOISY(17) MBX_RL_NOISY(19) MBX_RL_NOISY(21)
            default: rl_run_body_of_kind<THREADS, NPC, DC, GC, 0, -1, TIE>(klo, khi);      // any other (kind, noise model) pair: the any-kind body, out of line like the rest
            }
        } else {
            switch (kind) {
            MBX_RL_KIND(1) MBX_RL_KIND(2) MBX_RL_KIND(3) MBX_RL_KIND(4) MBX_RL_KIND(5) MBX_RL_KIND(6) MBX_RL_KIND(7) MBX_RL_KIND(8)
            MBX_RL_KIND(9) MBX_RL_KIND(10) MBX_RL_KIND(11) MBX_RL_KIND(12) MBX_RL_KIND(13) MBX_RL_KIND(14) MBX_RL_KIND(15) MBX_RL_KIND(16)
            MBX_RL_KIND(17) MBX_RL_KIND(18) MBX_RL_KIND(19) MBX_RL_KIND(20) MBX_RL_KIND(21) MBX_RL_KIND(22) MBX_RL_KIND(23) MBX_RL_KIND(24)
            default: rl_run_body_of_kind<THREADS, NPC, DC, GC, 0, -1, TIE>(klo, khi);
            }
        }
#undef MBX_RL_NOISY
#undef MBX_RL_KIND
    } else {
        typedef const RlRunArgs __attribute__((address_space(4))) CArgs;
        rl_run_body<THREADS, NPC, DC, GC, 0, -1, TIE>([]() -> CArgs& {
            return *(CArgs*)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        });
    }
}

}  // namespace mbx
