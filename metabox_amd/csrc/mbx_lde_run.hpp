// mbx_lde_run.hpp — LDE with the instance resident on chip across the generations of a launch, the LSTM policy evaluated inside the
// workgroup (reference: src/optimizer/lde_optimizer.py:159-198 = update(), src/agent/lde_agent.py:8-29,147-163 = PolicyNet + rollout loop).
//
// k_lde_step + k_lstm_policy are two launches per generation with a state round trip through HBM, a workgroup of 512 threads that
// spends most of its barrier intervals 50-100 threads wide, and 76 KB of LDS (two workgroups per CU); the PMC pass of round 4
// (profiles/r04a_lde_step_pmc.json) has the vector pipes 41 % busy and the waves waiting 64 % of their cycles.  k_lde_run is organised
// around the wave instead of the workgroup:
//   * a wave owns a tile of 16 population rows from the mutation to the end of the element-wise part of the objective.  The trial vector
//     is computed directly in the A-operand layout of v_mfma_f64_16x16x4_f64 (lane (c, q) holds row c, coordinates 4 s + q), so the first
//     linear map is 16 matrix instructions fed from registers; the transforms run on the accumulator (C / D layout, lane = (rows q + 4 r,
//     coordinate 16 ct + c)) and only the values the row sums need are written to the wave's own 16 x D slice of ONE LDS array.  No
//     workgroup barrier from the policy's action to the row sums; seven (pop 100) / four (pop 50) waves run that chain independently.
//   * the parents stay in LDS in PHYSICAL rows; the fitness order is an index permutation (ORDER / RANK) instead of a copy of the
//     population per generation; a surviving trial is written over its parent from the registers that still hold it.
//   * LDS: parents + one tile array + ~6 KB = 53.9 KB at pop 100 / D 30 -> THREE workgroups per CU (27 KB at pop 50: five).
//   * the LSTM cell + both heads are ~52 k float32 multiply-adds per generation: one fma chain per output unit in ascending k, the
//     same chain v_mfma_f32_16x16x4_f32 evaluates in k_lstm_policy (bit-identical: tests/test_gpu_lde.py), weights from L2.
//   * HBM traffic: the state block is read once and written once per launch.
// Every generation is the arithmetic of k_lde_step with the same Philox counters, so n launches of mbx_lde_policy + mbx_step and one
// launch of mbx_lde_rollout leave bit-identical state blocks, (h, c), rewards and features.
//
// Objectives: the element-wise part runs in the accumulator layout, so it is written per function kind here (same expressions, same
// out-of-line math routines, same summation order as eval_rows).  Kinds that need the candidate itself or two arrays in the row sums
// (3, 4, 5, 15, 20, 24) are built where a SECOND tile array fits the LDS budget -- D <= 16, i.e. k_lde_run<50, 10>, the reference's own LDE setting
// (bbob --dim 10, the shipped LDE_Agent.pkl); at D = 30 (config 3: the noisy suite has none of them) mbx_lde_rollout steps batches that contain them with
// one launch per generation.  The boundary penalty of every kind is exactly +0 here (the midpoint repair keeps a trial inside [lb, ub]) and is added as
// the literal it is.
#pragma once
#include "mbx_lde.hpp"
#include "mbx_lstm_policy.hpp"

namespace mbx {

struct LdeRunOut {
    float* traj_actions;         // [n_gens][B][2 NP] sampled actions, or nullptr
    double* traj_state;          // [n_gens][B][NP + 10] features after the generation, or nullptr
    double* traj_reward;         // [n_gens][B]
    uint8_t* traj_done;          // [n_gens][B]
    double* state_out;           // [B][NP + 10] features after the last executed generation
    double* reward_out;          // [B] SUM of the rewards of the executed generations, or nullptr
    uint8_t* done_out;           // [B]
};

// k-blocked copy of the packed PolicyNet weights for k_lde_run: a thread of the in-kernel policy owns ONE output unit and walks k, so four consecutive k of its unit
// are stored side by side -- gates: [ceil(K1 / 4)][4H] float4, bias [4H], mu head: [ceil(H / 4)][A] float4, sigma head likewise, bmu [A], bsg [A]; the padding
// k >= K1 / H is zero -- one 16-byte load per four k-steps instead of four 4-byte loads with an address each (half the instructions of the gate phase).  Rebuilt by a
// tiny launch at every mbx_lde_rollout call (the weights may have changed) into a buffer the batch owns.
__host__ __device__ inline int64_t lde_run_pack_floats(int IN, int H, int A)
{
    const int64_t K1 = IN + H, G4 = 4 * H;
    return 4 * ((K1 + 3) / 4) * G4 + G4 + 2 * 4 * (int64_t)((H + 3) / 4) * A + 2 * A;
}
template <int UNUSED = 0>
__global__ void k_lde_repack(LstmPolicy net, float* __restrict__ dst)
{
    const int IN = net.in_dim, H = net.hidden, A = net.out_dim, K1 = IN + H, G4 = 4 * H, KB1 = (K1 + 3) / 4, KBH = (H + 3) / 4;
    const float* W = net.w;                                         // [K1][G4] | b [G4] | WmuT [H][A] | WsgT [H][A] | bmu [A] | bsg [A]
    const float* bg = W + (int64_t)K1 * G4;
    const float* Wmu = bg + G4;
    const float* Wsg = Wmu + (int64_t)H * A;
    const float* bmu = Wsg + (int64_t)H * A;
    float* dg = dst;
    float* dbg = dg + (int64_t)4 * KB1 * G4;
    float* dmu = dbg + G4;
    float* dsg = dmu + (int64_t)4 * KBH * A;
    float* dbm = dsg + (int64_t)4 * KBH * A;
    const int64_t n1 = (int64_t)4 * KB1 * G4, n2 = (int64_t)4 * KBH * A;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n1 + G4 + 2 * n2 + 2 * A; t += (int64_t)gridDim.x * blockDim.x) {
        if (t < n1) { const int kk = (int)(t & 3), u = (int)((t >> 2) % G4), kb = (int)((t >> 2) / G4), k = 4 * kb + kk; dg[t] = k < K1 ? W[(int64_t)k * G4 + u] : 0.f; }
        else if (t < n1 + G4) dbg[t - n1] = bg[t - n1];
        else if (t < n1 + G4 + 2 * n2) {
            const int64_t r = t - n1 - G4;
            const bool sg = r >= n2;
            const int64_t q = sg ? r - n2 : r;
            const int kk = (int)(q & 3), j = (int)((q >> 2) % A), kb = (int)((q >> 2) / A), k = 4 * kb + kk;
            (sg ? dsg : dmu)[q] = k < H ? (sg ? Wsg : Wmu)[(int64_t)k * A + j] : 0.f;
        } else dbm[t - n1 - G4 - 2 * n2] = bmu[t - n1 - G4 - 2 * n2];          // bmu | bsg are contiguous in both layouts
    }
}

// ONE kernel argument: the kernel reads its fields through the kernarg segment pointer (scalar loads where they are used) instead of holding
// ~60 argument SGPRs across the generation loop, where they collide with the loop's own uniform values and are spilled to VGPR lanes.
struct LdeRunArgs {
    BatchParams bp;
    LstmPolicy net;              // net.w: the k-BLOCKED copy (k_lde_repack), not the caller's packed weights
    const double* state_in;      // [B][NP + 10] features of the last reset / step / rollout
    float* hbuf;                 // [B][H]
    float* cbuf;                 // [B][H]
    int32_t n_gens;
    LdeRunOut out;
};
typedef const LdeRunArgs __attribute__((address_space(4))) LdeRunCArgs;
__device__ __forceinline__ LdeRunCArgs& lde_run_args()
{
    uintptr_t p = (uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                                   // not loop-invariant, not common with any other use: every field access is a fresh s_load
    return *(LdeRunCArgs*)p;
}

// one column tile (D <= 16): a second tile array TB2 exists and with it the six kinds whose row sums read two arrays / the candidate
__host__ __device__ constexpr bool lde_run_two_arrays(int D) { return D <= 16; }
// two: the instantiation carries the second tile array (always at D <= 16; at NP 50 / D 30 as the instantiation of its own k_lde_run<50, 30, 50, true>)
__host__ __device__ constexpr bool lde_run_kind_ok(int kind, int D, bool two)
{
    return two ? (kind >= 1 && kind <= 24)
                                 : (kind == 1 || kind == 2 || (kind >= 6 && kind <= 14) || (kind >= 16 && kind <= 19) || (kind >= 21 && kind <= 23));
}

__host__ __device__ constexpr int lde_run_tiles(int NP) { return (NP + 15) / 16; }
__host__ __device__ constexpr int lde_run_threads(int NP) { return 64 * lde_run_tiles(NP); }

struct LdeRunLds {
    double *P, *TB, *TB2, *FIT, *A1, *A2, *FEAT, *HS, *RED, *SCAL, *DSH;
    float *ACT, *HC;
    uint8_t *RK, *ORDER;         // rank of a physical row / row at a rank (NP <= 255)
    int *ACC, *HIST, *FLAG;      // ACC: the ranking's per-row counters
};

// LDS is allocated in 1280-byte granules: 3 workgroups per CU need <= 53 760 B each (pop 100: 52 768 B), 6 need <= 26 880 B (pop 50: 26 880 B)
__host__ __device__ inline int64_t lde_run_lds_doubles(int NP, int D, int H, bool two)
{
    const int64_t NE = align2((int64_t)NP * D), P = align2(NP);
    return (two ? 3 : 2) * NE + 2 * P + align2(NP + 2 * MBX_LDE_BINS) + 8 + 8 + 8 + P /* ACT: 2 NP floats */ + align2(H) /* h | c */ + align2((P + 1) / 2) + 2 * align2((P + 7) / 8) + 4 + align2(D);
}

__device__ __forceinline__ LdeRunLds lde_run_carve(double* base, int NP, int D, int H, bool two)
{
    const int64_t NE = align2((int64_t)NP * D), P = align2(NP), PI = align2((P + 1) / 2);
    LdeRunLds L;
    double* p = base;
    L.P = p; p += NE;  L.TB = p; p += NE;
    L.TB2 = nullptr;
    if (two) { L.TB2 = p; p += NE; }           // second tile array (kinds 3, 4, 5, 15, 20, 24)
    L.FIT = p; p += P;
    L.A1 = p; p += P;            // SORTED (from the ranking to the next row sums) | Gallagher: best key per row
    L.FEAT = p; L.A2 = p; p += align2(NP + 2 * MBX_LDE_BINS);   // features; dead between the policy's input staging and the next feature phase, where
                                 // the objective uses the storage: F0 (step ellipsoid: |z_hat_0|) | Gallagher: winning peak per row
    L.HS = p; p += 8;  L.RED = p; p += 8;  L.SCAL = p; p += 8;
    L.ACT = (float*)p; p += P;
    L.HC = (float*)p; p += align2(H);
    L.ACC = (int*)p; p += PI;  L.RK = (uint8_t*)p; p += align2((P + 7) / 8);  L.ORDER = (uint8_t*)p; p += align2((P + 7) / 8);
    L.HIST = (int*)p;            // [5] bin counts, [6] the done flag
    L.FLAG = L.HIST + 6;
    p += 4;
    L.DSH = p;                   // the problem's shift vector (0 where the objective has none)
    return L;
}

// LDS traffic between the lanes of ONE wave (a store by one lane, a load of that address by another): the DS unit executes a wave's
// operations in program order, so all that is needed is that the compiler keeps the order and that the data has landed
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// order-preserving map double -> uint64 (for the LDS max of the Gallagher keys)
__device__ __forceinline__ unsigned long long f64_sortable(double v)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// Gallagher: the winning peak of every row (block-wide: lane = row, wave = a subset of the peaks, scalar-operand peak tables).  Out of line on purpose: inlined, this
// block (unrolled 4 peaks x 6 coordinates) costs every OTHER objective kind 4 % (Sphere batch at pop 100: 0.599 -> 0.574 ms with it compiled out) through the register
// allocation and the code size of the generation body.  The LDS arrays arrive as generic pointers and are cast back to the LDS address space.
#ifndef MBX_LDE_GALL_ATTR
#define MBX_LDE_GALL_ATTR __noinline__          // (__forceinline__ for the A/B)
#endif
template <int NP, int D>
__device__ MBX_LDE_GALL_ATTR void lde_gallagher_search(double* TB_, unsigned long long* GK_, int* GI_, ConstProblem* Pp)
{
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) unsigned long long lds_u64;
    typedef __attribute__((address_space(3))) int lds_i32;
    lds_f64* TB = (lds_f64*)TB_;
    lds_u64* GK = (lds_u64*)GK_;
    lds_i32* GI = (lds_i32*)GI_;
    // function arguments arrive in VGPRs: without this the problem record, and with it the peak tables, would be read by per-lane vector loads instead of scalar ones
    const uint64_t pu_ = (uint64_t)(uintptr_t)Pp;
    const uint64_t pu = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pu_) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pu_ >> 32)) << 32);
    ConstProblem& P = *(ConstProblem*)(uintptr_t)pu;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            for (int i = tid; i < NP; i += (int)blockDim.x) { GK[i] = 0ull; GI[i] = 0x7fffffff; }
            __syncthreads();
            typedef const double __attribute__((address_space(4)))* kptr;
            const kptr py = (kptr)P.pyr, pcc = (kptr)P.pc, plw = (kptr)P.plogw;
            const int npk = P.n_peaks, nw = __builtin_amdgcn_readfirstlane(MBX_NW);
            const double cexp = -0.5 / D;
            constexpr int PB = 4, GC = D % 6 == 0 ? 6 : 5;
            static_assert(D % GC == 0, "whole chunks only");
            const int mine = wave < npk ? (npk - wave + nw - 1) / nw : 0;
            double bkey[2] = {-INFINITY, -INFINITY};
            int bk[2] = {0, 0};
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                if (64 * qq >= NP) break;
                const int i = lane + 64 * qq;
                const lds_f64* rx = TB + (i < NP ? i : NP - 1) * D;
                for (int j0 = 0; j0 < mine; j0 += PB) {
                    double acc[PB];
#pragma unroll
                    for (int j = 0; j < PB; ++j) acc[j] = 0.;
#pragma unroll 1
                    for (int c0 = 0; c0 < D; c0 += GC) {
                        double y[GC];
#pragma unroll
                        for (int k = 0; k < GC; ++k) y[k] = rx[c0 + k];
#pragma unroll
                        for (int j = 0; j < PB; ++j) {
                            if (j0 + j < mine) {
                                const int kk = wave + (j0 + j) * nw;
                                const kptr ry = py + (int64_t)kk * D + c0;
                                const kptr ck = pcc + (int64_t)kk * D + c0;
#pragma unroll
                                for (int k = 0; k < GC; ++k) { const double zd = y[k] - ry[k]; acc[j] = __builtin_fma(ck[k], zd * zd, acc[j]); }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        if (j0 + j < mine) {
                            const int kk = wave + (j0 + j) * nw;
                            const double key = plw[kk] + cexp * acc[j];
                            if (key > bkey[qq]) { bkey[qq] = key; bk[qq] = kk; }
                        }
                    }
                }
                if (i < NP && mine > 0) __hip_atomic_fetch_max(&GK[i], f64_sortable(bkey[qq]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const int i = lane + 64 * qq;
                if (i < NP && mine > 0 && f64_sortable(bkey[qq]) == GK[i]) __hip_atomic_fetch_min(&GI[i], bk[qq], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // equal keys: the lower peak index (eval_rows' rule)
            }
            __syncthreads();
}

// z[row][d] of a wave's 16-row tile <- sum_k M[d][k] a[row][k] on the float64 matrix cores: lane (c, q) supplies A = a[row c][4 s + q] and B = M[column tile][4 s + q], the
// accumulator tiles (C / D layout: rows q + 4 r, coordinates c and 16 + c) go to the wave's slice TW.  The fma chain of matvec_rows_mfma (ascending k from +0).
template <int NP, int D, class TWP>
__device__ __forceinline__ void lde_map_tile(const double* __restrict__ M, const double (&a)[(D + 3) / 4], TWP TW, int wave, int c, int q)
{
    constexpr int KS = (D + 3) / 4;
    constexpr bool TWO_TILES = D > 16;
    const int c0 = c < D ? c : D - 1;                              // (D < 16: the columns past the dimension are clamped and masked)
    const int c1 = 16 + c < D ? 16 + c : D - 1;                    // second column tile, clamped
    f64x4 y0 = {0., 0., 0., 0.}, y1 = {0., 0., 0., 0.};
    double bm0[KS], bm1[TWO_TILES ? KS : 1];
#pragma unroll
    for (int s = 0; s < KS; ++s) bm0[s] = M[c0 * D + (4 * s + q < D ? 4 * s + q : D - 1)];
    if constexpr (TWO_TILES) {
#pragma unroll
        for (int s = 0; s < KS; ++s) bm1[s] = M[c1 * D + (4 * s + q < D ? 4 * s + q : D - 1)];
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], (4 * s + q < D && c < D) ? bm0[s] : 0., y0, 0, 0, 0);
    if constexpr (TWO_TILES) {
#pragma unroll
        for (int s = 0; s < KS; ++s) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], (4 * s + q < D && 16 + c < D) ? bm1[s] : 0., y1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lr = q + 4 * r;
        if (16 * wave + lr < NP) {
            if (c < D) TW[lr * D + c] = y0[r];
            if constexpr (TWO_TILES) { if (16 + c < D) TW[lr * D + 16 + c] = y1[r]; }
        }
    }
}

// Everything element-wise between the first linear map and the row sums (phases E1, C, E2 of eval_rows), IN PLACE in the wave's slice, one element per lane at a
// time.  Out of line like the Gallagher search: the generation body (policy, mutation, ranking) is register-allocated without it, and the kinds that have
// nothing element-wise never call it.
template <int NP, int D, int KIND = 0, bool TA = lde_run_two_arrays(D)>
__device__ MBX_LDE_GALL_ATTR void lde_tile_transforms(double* TW_, double* A2_, ConstProblem* Pp, int kind_, double* TW2_ = nullptr)
{
    typedef __attribute__((address_space(3))) double lds_f64;
    lds_f64* TW = (lds_f64*)TW_;
    lds_f64* A2 = (lds_f64*)A2_;
    lds_f64* TW2 = (lds_f64*)TW2_;                                  // the wave's slice of the second tile array (D <= 16 only)
    constexpr int KS = (D + 3) / 4;
    constexpr bool TWO = TA && KIND == 0;                          // kinds 3, 4, 5, 15, 20, 24 live in the any-kind loop of the instantiations that carry the second tile array
    constexpr int MM = D > 16 ? 8 : 4;                             // elements per lane: 4 rows x one or two column tiles
    const uint64_t pu_ = (uint64_t)(uintptr_t)Pp;
    const uint64_t pu = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pu_) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pu_ >> 32)) << 32);
    ConstProblem& P = *(ConstProblem*)(uintptr_t)pu;
    const int kind = KIND ? KIND : __builtin_amdgcn_readfirstlane(kind_);      // KIND: the caller's compile-time kind (lde_run_generations)
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool rvalid = 16 * wave + c < NP;
    int kq[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kq[s] = 4 * s + q < D ? 4 * s + q : D - 1;
        // element m of a lane: m = 4 ct + r -> (local row q + 4 r, coordinate 16 ct + c)
        auto elem = [&](int m, int& lr, int& d) -> bool { lr = q + 4 * (m & 3); d = 16 * (m >> 2) + c; return 16 * wave + lr < NP && d < D; };
        if (kind == 7 && c == 0) {                                 // F7 keeps |z_hat_0| (bbob.py: the max() of the step ellipsoid)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const int lr = q + 4 * r; if (16 * wave + lr < NP) A2[16 * wave + lr] = fabs(TW[lr * D]); }
        }
        {                                                          // phase E1 of eval_rows
            const double s0 = P.s[0];
            const double kats_exp = kind == 23 ? 10 / m_pow((double)D, 1.2) : 0.;
#pragma unroll 1
            for (int m = 0; m < ((TWO && kind == 20) ? 0 : MM); ++m) {       // (Schwefel has its own two passes below)
                int lr, d;
                if (!elem(m, lr, d)) continue;
                lds_f64* pz = TW + lr * D + d;
                const double z = *pz;
                double t;
                if constexpr (TWO) {
                    if (kind == 3 || kind == 4) {                  // Rastrigin / Bueche-Rastrigin (eval_rows phase E1): z stays, cos(2 pi z) goes to the second array
                        double zz;
                        if (kind == 3) zz = P.v0[d] * asy1(osc1(z), P.v1[d]);
                        else { double o = osc1(z); if ((d & 1) == 0 && o > 0.) o *= 10.; zz = o * P.v0[d]; }
                        *pz = zz;
                        TW2[lr * D + d] = m_cos(kTwoPi * zz);
                        continue;
                    }
                    if (kind == 15) { *pz = asy1(osc1(z), P.v1[d]); continue; }
                    if (kind == 24) { *pz = m_cos(kTwoPi * z); continue; }      // (the slice holds M1 (x_hat - mu0): phase E2)
                }
                switch (kind) {
                case 2: case 10: { const double o = osc1(z); t = P.v0[d] * (o * o); break; }
                case 6: { double zi = z; if (zi * P.dshift[d] > 0.) zi *= 100.; t = zi * zi; break; }
                case 7: t = fabs(z) > 0.5 ? floor(0.5 + z) : floor(0.5 + 10. * z) / 10.; break;
                case 8: t = s0 * z + 1; break;
                case 9: case 19: t = z + 0.5; break;
                case 11: { const double o = osc1(z); t = o * o; break; }
                case 12: case 17: case 18: t = asy1(z, P.v1[d]); break;
                case 14: t = m_pow(fabs(z), P.v0[d]); break;
                case 16: t = osc1(z); break;
                case 23: {
                    double temp = 0., p2 = 1., ip2 = 1.;
                    for (int j = 1; j <= 32; ++j) {
                        p2 *= 2.; ip2 *= 0.5;
                        const double a = p2 * z;
                        temp += fabs(a - floor(a + 0.5)) * ip2;
                    }
                    t = m_pow(1 + (d + 1) * temp, kats_exp);
                    break;
                }
                default: t = z; break;
                }
                *pz = t;
            }
        }
        if constexpr (TWO) {
            if (kind == 20) {                                      // Schwefel (bbob.py:754-756; eval_rows phases E1 / E2): T0 = v2 x is in the second array
                wave_lds_fence();
#pragma unroll 1
                for (int m = 0; m < MM; ++m) {
                    int lr, d;
                    if (!elem(m, lr, d)) continue;
                    double zi = TW2[lr * D + d];
                    if (d > 0) zi += 0.25 * (TW2[lr * D + d - 1] - P.v1[d - 1]);
                    TW[lr * D + d] = 100. * (P.v0[d] * (zi - P.v1[d]) + P.v1[d]);
                }
                wave_lds_fence();                                  // every T0 has been read: the second array takes the penalty terms
#pragma unroll 1
                for (int m = 0; m < MM; ++m) {
                    int lr, d;
                    if (!elem(m, lr, d)) continue;
                    const double z = TW[lr * D + d];
                    const double qq = fmax(0., fabs(z / 100) - P.ub);
                    TW2[lr * D + d] = qq * qq;
                    TW[lr * D + d] = z * m_sin(sqrt(fabs(z)));
                }
            }
        }
        if ((kind == 7 || kind == 12 || (TWO && kind == 15) || (kind >= 16 && kind <= 18))) {    // second linear map (F7, F15-F18: M2; F12: M1 again) of the tile just written
            wave_lds_fence();
            double av2[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) { const double v = TW[(rvalid ? c : 0) * D + kq[s]]; av2[s] = (rvalid && 4 * s + q < D) ? v : 0.; }
            wave_lds_fence();                                      // every lane holds its A fragment: the slice can take the product
            lde_map_tile<NP, D>(kind == 12 ? P.m1 : P.m2, av2, TW, wave, c, q);
            if constexpr (TWO) {
                if (kind == 15) {                                  // Rastrigin F15: z = M2 (...) stays, cos(2 pi z) goes to the second array
#pragma unroll 1
                    for (int m = 0; m < MM; ++m) {
                        int lr, d;
                        if (!elem(m, lr, d)) continue;
                        TW2[lr * D + d] = m_cos(kTwoPi * TW[lr * D + d]);
                    }
                }
            }
            if (kind == 16) {                                      // Weierstrass series by angle tripling (eval_rows, phase E2)
#pragma unroll 1
                for (int m = 0; m < MM; ++m) {
                    int lr, d;
                    if (!elem(m, lr, d)) continue;
                    lds_f64* pz = TW + lr * D + d;
                    const double base = kTwoPi * (*pz + 0.5);
                    double cc = m_cos(base), ss = m_sin(base), sum = cc, ak = 1.;
#pragma unroll
                    for (int k = 1; k < 12; ++k) {
                        const double c2 = cc * cc, s2 = ss * ss;
                        cc = cc * __builtin_fma(-3., s2, c2);
                        ss = ss * __builtin_fma(3., c2, -s2);
                        ak *= 0.5;
                        sum = __builtin_fma(ak, cc, sum);
                    }
                    *pz = sum;
                }
            }
        }
        if ((kind == 17 || kind == 18 || kind == 19)) {              // terms of neighbouring coordinates (bbob.py:642-643, 702-703)
            // element (row, d) needs (row, d + 1), which another lane owns and overwrites with ITS term: the wave walks m in step -- all lanes read,
            // then all lanes write -- and in ascending m, so the one neighbour that belongs to a later step (d = 15 -> 16) is still untouched
#pragma unroll 1
            for (int m = 0; m < MM; ++m) {
                int lr, d;
                const bool ok = elem(m, lr, d) && d < D - 1;       // coordinate D - 1 has no term
                wave_lds_fence();
                const double a = ok ? TW[lr * D + d] : 0., nb = ok ? TW[lr * D + d + 1] : 0.;
                wave_lds_fence();
                if (ok) {
                    double t;
                    if (kind == 19) {
                        const double aa = a * a - nb;
                        const double bb = 1. - a;
                        const double sq = 100. * (aa * aa) + bb * bb;
                        t = sq / 4000. - m_cos(sq);
                    } else {
                        const double sq = sqrt(a * a + nb * nb);
                        t = sqrt(sq) * (m_pow(m_sin(50 * m_pow(sq, 0.2)), 2) + 1);
                    }
                    TW[lr * D + d] = t;
                }
            }
        }
}

// timing experiments only (never a shipped build): bit 0 no gate chains, 1 no head chains, 2 cheap hash instead of Philox in the tile phase,
// 4 no noise in the row phase, 5 no ranking pass, 6 parent gathers from the lane's own row (no bank conflicts in the mutation)
#ifndef MBX_LDE_ABL
#define MBX_LDE_ABL 0
#endif
// waves per SIMD the register allocation aims at: 6 (80 VGPRs) -- pop 100: 3 workgroups x 7 waves per CU; pop 50: 6 workgroups x 4 waves (26 880 B of LDS each, the
// rank tables as bytes).  With the first versions' ~30 spilled registers 96 VGPRs / five workgroups was the faster pop-50 build; with the transforms and the Gallagher
// search out of line the 80-register build spills 13 (pop 100) / a handful (pop 50) and the two are level (0.381 / 0.380 ms, same box).
#ifndef MBX_LDE_RUN_WAVES
#define MBX_LDE_RUN_WAVES(NP) 6
#endif

// loop-carried scalars of the instance, in LDS (thread 0 updates them at the end of a generation, every thread reads what it needs at the top of the next)
enum { LR_FES = 0, LR_HCOUNT, LR_BSF, LR_RSUM, LR_RTOT, LR_LOGI, LR_CLEN };

// The generations of ONE launch (everything between the prologue that stages the state block and the epilogue that writes it back).  Out of line, and instantiated once per
// objective kind of the noisy suite besides the any-kind form (KIND = 0: the kind is read from the problem record): a workgroup runs the loop of ITS kind, register-allocated
// and scheduled without the other kinds' code (the any-kind kernel ran a Sphere batch 10 % slower than a build with everything but Sphere compiled out).  One launch for the
// mixed batch all the same: one launch per kind on side streams lost more in the eight tails than the specialisation gained (docs/EXPERIMENTS.md).
// Arguments arrive in VGPRs and are made wave-uniform again (the address of the kernel's argument block among them); the LDS carve-up is rebuilt from the dynamic LDS symbol.  Returns the number of generations executed.
template <int NPC, int DC, int HC_, int KIND, bool TA>
__device__ __noinline__ int lde_run_generations(int b_, int gen0_, int episode_, int n_gens_, uint32_t seed_lo_, uint32_t seed_hi_, uint32_t karg_lo_, uint32_t karg_hi_)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NP = NPC, D = DC, NE = NP * D, H = HC_, IN = NP + 2 * MBX_LDE_BINS, A = 2 * NP, G4 = 4 * H, K1 = IN + H;
    constexpr int TILES = (NP + 15) / 16, KS = (D + 3) / 4, NF = NP + 2 * MBX_LDE_BINS;
    constexpr bool TWO = TA && KIND == 0;                          // the six two-array kinds: any-kind loop of the instantiations that carry the second tile array
    // crossover uniforms of a row: D consecutive elements of the sorted population starting at offset (i D) & 3 in {0, 2} (D even) -> GPR Philox groups
    constexpr int GPR = (D + 2 + 3) / 4, ROWW = 4 * GPR;
    static_assert(D % 2 == 0 && 16 * ROWW * 4 <= 16 * D * 8, "the uniforms of a tile fit the wave's slice");
    static_assert(4 * ((K1 + 3) / 4) - IN <= 64, "the hidden state is staged by one wave");
    const LdeRunLds L = lde_run_carve(smem, NP, D, H, TA);
    const int b = __builtin_amdgcn_readfirstlane(b_), gen0 = __builtin_amdgcn_readfirstlane(gen0_), episode = __builtin_amdgcn_readfirstlane(episode_);
    const int n_gens = __builtin_amdgcn_readfirstlane(n_gens_);
    const uint32_t seed_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)seed_lo_), seed_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)seed_hi_);
    // the kernel's argument block: its address is an argument (in a callee s[8:9] -- what __builtin_amdgcn_kernarg_segment_ptr() reads -- is the IMPLICIT argument pointer)
    const uint32_t karg_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)karg_lo_), karg_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)karg_hi_);
    auto lde_run_args = [&]() -> LdeRunCArgs& {
        uintptr_t p = (uintptr_t)(((uint64_t)karg_hi << 32) | karg_lo);
        asm volatile("" : "+s"(p));                               // (not common with any other use: every field access is a fresh s_load)
        return *(LdeRunCArgs*)p;
    };
    MBX_PHASE_BEGIN
    int executed = 0;

    for (int it = 0; it < n_gens; ++it) {
        // Nothing below may look loop-invariant to the compiler: what it hoists out of a loop this long it spills (the first version of this kernel
        // carried 190 spilled SGPRs and 41 scratch stores in its prologue).  Thread index, argument block and problem record are re-materialised
        // behind empty asm statements at the top of every generation (cf. opaque_tid() and k_rlepso_run).
        const int tid = opaque_tid();
        const int lane = tid & 63, c = lane & 15, q = lane >> 4;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        LdeRunCArgs& ar = lde_run_args();
        uintptr_t pp_ = (uintptr_t)(ar.bp.problems + __builtin_amdgcn_readfirstlane(ar.bp.problem_idx[b]));
        asm volatile("" : "+s"(pp_));
        ConstProblem& P = *(ConstProblem*)pp_;
#ifdef MBX_LDE_ONLY_KIND
        const int kind = MBX_LDE_ONLY_KIND;                        // timing experiment: code size (every other kind compiled out)
#else
        const int kind = KIND ? KIND : P.kind;
#endif
        const double lb = P.lb, ub = P.ub, bias = P.bias;
        const int gen = gen0 + it + 1;
        const Rng rng{seed_lo, seed_hi, (uint32_t)gen, (uint32_t)episode, true};
        const double fes = L.SCAL[LR_FES];
        const uint8_t* RKa = L.RK;                                 // rank of every physical row in the current fitness order
        int* RKb = L.ACC;                                          // accumulator of the next ranking
        constexpr int KB1 = (K1 + 3) / 4, KBH = (H + 3) / 4;
        const float4* WG = (const float4*)ar.net.w;                // k-blocked (lde_run_pack_floats): gates [KB1][4H] float4 | b [4H] | mu [KBH][A] float4 | sigma likewise | bmu | bsg
        const float* bg = ar.net.w + (int64_t)4 * KB1 * G4;
        const float4* WMU = (const float4*)(bg + G4);
        const float4* WSG = WMU + (int64_t)KBH * A;
        const float* bmu = (const float*)(WSG + (int64_t)KBH * A);
        const float* bsg = bmu + A;
        // ================================================================ policy: LSTM cell + heads + sampling (lde_agent.py:8-29)
        // The tile array is idle between the row sums and the next mutation: it holds the cell's inputs [x | h] as float32 (staged by the
        // previous generation's feature phase / the prologue), the gate pre-activations and the sigma head's output.  A chain is one fma per k in
        // ascending k; the weights of LB steps are fetched (L2) before the LB fmas that use them -- few, large batches: the phase is the latency
        // of its dependent L2 round trips.
        float* XS = (float*)L.TB;                                  // [K1]
        float* GT = XS + ((K1 + 3) & ~3);                          // [4H]
        float* SG = GT + G4;                                       // [A] sigma head, pre-activation
        constexpr int LBQ = 10;                                    // float4 (= 4 k-steps) per batch
        if (tid < G4) {
            float acc = bg[tid];
            const float4* wcol = WG + tid;
#pragma unroll 1
            for (int kb0 = 0; kb0 < ((MBX_LDE_ABL & 1) ? 0 : KB1); kb0 += LBQ) {
                float4 wv[LBQ];
#pragma unroll
                for (int j = 0; j < LBQ; ++j) wv[j] = kb0 + j < KB1 ? wcol[(int64_t)(kb0 + j) * G4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < LBQ; ++j) {
                    if (kb0 + j < KB1) {
                        const float4 x = *(const float4*)(XS + 4 * (kb0 + j));      // (XS is zero beyond K1: the padded steps add fma(0, 0, acc) = acc, like the MFMA kernel's)
                        acc = __builtin_fmaf(x.x, wv[j].x, acc); acc = __builtin_fmaf(x.y, wv[j].y, acc);
                        acc = __builtin_fmaf(x.z, wv[j].z, acc); acc = __builtin_fmaf(x.w, wv[j].w, acc);
                    }
                }
            }
            GT[tid] = acc;
        }
        __syncthreads();
        MBX_PHASE(0);                                              // LSTM gates
        if (tid < H) {
            const float gi = sigmoidf_(GT[tid]), gf = sigmoidf_(GT[H + tid]);
            const float gg = tanhf(GT[2 * H + tid]), go = sigmoidf_(GT[3 * H + tid]);
            const float cn = gf * L.HC[H + tid] + gi * gg;
            const float hn = go * tanhf(cn);
            L.HC[H + tid] = cn;
            L.HC[tid] = hn;
        }
        __syncthreads();
        // heads: thread j < A runs the mu chain of component j, thread A + j the sigma chain (when the workgroup has 2 A threads; else both in thread j)
        constexpr bool SPLIT = 2 * A <= 64 * TILES;
        float am = 0.f;
        if (tid < (SPLIT ? 2 * A : A)) {
            const int j = SPLIT && tid >= A ? tid - A : tid;
            const bool sig = SPLIT && tid >= A;
            const float4* W = (sig ? WSG : WMU) + j;
            float acc = sig ? bsg[j] : bmu[j], acc2 = SPLIT ? 0.f : bsg[j];
            float4 wm[KBH], ws[SPLIT ? 1 : KBH];
#pragma unroll
            for (int kb = 0; kb < ((MBX_LDE_ABL & 2) ? 1 : KBH); ++kb) { wm[kb] = W[(int64_t)kb * A]; if (!SPLIT) ws[kb] = WSG[(int64_t)kb * A + j]; }
#pragma unroll
            for (int kb = 0; kb < ((MBX_LDE_ABL & 2) ? 1 : KBH); ++kb) {
                // h[k] for k >= H reads into the cell state behind it (finite) against a zero weight: the padded steps of the MFMA kernel
                const float h0 = L.HC[4 * kb], h1 = L.HC[4 * kb + 1], h2 = L.HC[4 * kb + 2], h3 = L.HC[4 * kb + 3];
                acc = __builtin_fmaf(h0, wm[kb].x, acc); acc = __builtin_fmaf(h1, wm[kb].y, acc); acc = __builtin_fmaf(h2, wm[kb].z, acc); acc = __builtin_fmaf(h3, wm[kb].w, acc);
                if (!SPLIT) { acc2 = __builtin_fmaf(h0, ws[kb].x, acc2); acc2 = __builtin_fmaf(h1, ws[kb].y, acc2); acc2 = __builtin_fmaf(h2, ws[kb].z, acc2); acc2 = __builtin_fmaf(h3, ws[kb].w, acc2); }
            }
            if (sig) SG[j] = acc; else am = acc;
            if (!SPLIT) SG[j] = acc2;
        }
        if (SPLIT) __syncthreads();
        if (tid < A) {
            const float a = sample_action(rng, tid, am, sigmoidf_(SG[tid]), MBX_POLICY_RLEPSO);
            L.ACT[tid] = a;
            if (ar.out.traj_actions) ar.out.traj_actions[((int64_t)it * ar.bp.B + b) * A + tid] = a;
        }
        __syncthreads();
        MBX_PHASE(1);                                              // cell update, heads, sampling

        // ================================================================ the wave's tile: mutation -> first map -> transforms
        // Branch-free on purpose: every load is unconditional with a clamped index and the padding is selected away afterwards -- a conditional
        // load costs an EXEC save / restore and a branch each, serialises the LDS round trips behind it, and its mask lives in an SGPR pair (the first
        // version of this phase ran 16 k cycles per generation alone on a CU, most of them in such chains).
        const double p_rate = (2. / NP - 1) * fes / ar.bp.max_fes + 1;
        const int bound = (int)ceil(NP * fmax(0., p_rate));
        const int my_hist = tid < MBX_LDE_BINS ? lde_unpack_hist(L.HS[MBX_LDE_BINS], tid) : 0;
        const int jrow = 16 * wave + c;                           // the row of this lane's A-layout elements
        const bool rvalid = jrow < NP;
        const int jr_c = rvalid ? jrow : NP - 1;                   // clamped: lanes past the last row compute on a copy of it, nothing of theirs is stored
        double* TW = L.TB + 16 * wave * D;                         // this wave's slice of the tile array: rows 16 wave .. (row-major, D per row)
        double* TW2 = TWO ? L.TB2 + 16 * wave * D : nullptr;       // ... and of the second one
        // the trial vector is parked in the instance's own population block in HBM (free until the end of the launch: the parents live in LDS) and read
        // back by the lanes of the rows that survive -- 16 registers less across the transforms than holding it, and no compiler-placed spill chain
        double* park = ar.bp.state + (int64_t)b * ar.bp.state_stride + MBX_LDE_ST_POP(NP, D);
        const bool gall = kind == 21 || kind == 22;
        int kq[KS];                                                // coordinate 4 s + q of the A / B fragments, clamped
#pragma unroll
        for (int s = 0; s < KS; ++s) kq[s] = 4 * s + q < D ? 4 * s + q : D - 1;
        double av[KS];                                             // A operand of the first map: trial - shift (0 in the padding)
        {
            const int i = RKa[jr_c];                               // position of the row in the fitness order = the individual's index
            // ---- crossover uniforms of the tile: ONE Philox call per group of four consecutive elements of the SORTED population (site LDE_ELEM,
            // index e >> 2), like k_lde_step; a row touches 8 groups, lane m of the wave makes the calls of (row m >> 3, group m & 7)
            uint32_t* UW = (uint32_t*)TW;
#pragma unroll
            for (int h2 = 0; h2 < (16 * GPR + 63) / 64; ++h2) {
                const int m = lane + 64 * h2, rr = m / GPR, g = m - rr * GPR, jr_ = 16 * wave + rr;      // (GPR = 8 at D = 30: m >> 3, m & 7)
                const int ii = RKa[jr_ < NP ? jr_ : NP - 1];
                const int t = ((ii * D) >> 2) + g;
                const U4 w = (MBX_LDE_ABL & 4) ? U4{(uint32_t)t * 2654435761u, (uint32_t)t * 40503u + 7u, (uint32_t)t ^ 0x9E3779B9u, (uint32_t)t * 69069u} : rng.draw((uint32_t)t, MBX_SITE_LDE_ELEM);
                if (rr < 16 && jr_ < NP && 4 * t < NE) *(uint4*)(UW + rr * ROWW + 4 * g) = make_uint4(w.x, w.y, w.z, w.w);
            }
            // ---- per-individual draws (:101-105, 88-99)
            const U4 w = (MBX_LDE_ABL & 4) ? U4{(uint32_t)i * 2654435761u, (uint32_t)i * 40503u + 7u, (uint32_t)i ^ 0x9E3779B9u, (uint32_t)i * 69069u} : rng.draw((uint32_t)i, MBX_SITE_LDE_PART);
            const int pidx = (int)__umulhi(w.x, (uint32_t)bound);
            int r0 = (int)__umulhi(w.y, (uint32_t)(NP - 1)); r0 += r0 >= i;
            int r1 = (int)__umulhi(w.z, (uint32_t)(NP - 2));
            { const int lo = i < r0 ? i : r0, hi = i < r0 ? r0 : i; r1 += r1 >= lo; r1 += r1 >= hi; }
            const int jr = (int)__umulhi(w.w, (uint32_t)D);
            const float sf32 = L.ACT[i];
            const double sf = (double)sf32, cr = (double)L.ACT[NP + i], om = (double)(1.f - sf32);
            // (MBX_LDE_ABL bit 6: the three parent gathers read the lane's own row instead of random rows -- what the bank conflicts of this phase cost)
            const double* rowP = L.P + ((MBX_LDE_ABL & 64) ? jr_c : L.ORDER[pidx]) * D;
            const double* row0 = L.P + ((MBX_LDE_ABL & 64) ? jr_c : L.ORDER[r0]) * D;
            const double* row1 = L.P + ((MBX_LDE_ABL & 64) ? jr_c : L.ORDER[r1]) * D;
            const double* rowI = L.P + jr_c * D;
            const uint32_t* urow = UW + (jr_c - 16 * wave) * ROWW + ((i * D) & 3);   // (the clamped lanes read the uniforms of the row they are clamped to)
            double* pk = park + jr_c * D;
            const bool self = pidx == i;
            wave_lds_fence();
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += 2) {                   // two coordinates at a time: their 12 LDS reads in flight, then their arithmetic
                double xi[2], xp[2], x0[2], x1[2], sh[2];
                uint32_t uw[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (s0 + j >= KS) break;                      // (KS is odd at D = 10)
                    const int d = kq[s0 + j];
                    xi[j] = rowI[d]; xp[j] = rowP[d]; x0[j] = row0[d]; x1[j] = row1[d]; uw[j] = urow[d]; sh[j] = L.DSH[d];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (s0 + j >= KS) break;
                    const int s = s0 + j, d = kq[s];
                    const double u = d == jr ? 0. : u32d(uw[j]);
                    // :88-99: x_i + F (x_pbest - x_i) + F (x_r0 - x_r1), written like the reference's tensor arithmetic: the two orders of the first sum
                    // (pidx < i / pidx > i in k_lde_step) are the same addition, pidx == i leaves x_i
                    const double blend = sf * xp[j] + om * xi[j];
                    double m = self ? xi[j] : blend;
                    m = m + sf * (x0[j] - x1[j]);
                    double cval = u <= cr ? m : xi[j];
                    const double rl = (xi[j] + lb) / 2., ru = (xi[j] + ub) / 2.;
                    cval = cval < lb ? rl : (cval > ub ? ru : cval);
                    pk[d] = cval;                                  // unconditional: a lane in the padding computes the element it is clamped to, bit for bit what its owner stores
                    double aop = cval - sh[j];
                    if constexpr (TWO) {
                        // the kinds without a first map read the candidate itself (eval_rows phase E1): their first array goes straight to the SECOND slice (the first
                        // still holds the uniforms), in [row][coordinate] order
                        if (kind == 24) aop = P.v0[d] * cval - P.s[0];      // x_hat - mu0: the operand of the M1 map (phase C)
                        if (rvalid && 4 * s + q < D && (kind == 5 || kind == 20 || kind == 24)) {
                            double tv = cval;                                  // F24: the row sums need x itself
                            if (kind == 5) {
                                double zi = cval;
                                if (cval * sh[j] > ub * ub) zi = (zi > 0. ? 1. : (zi < 0. ? -1. : 0.)) * ub;
                                tv = P.v1[d] - zi * P.v0[d];
                            } else if (kind == 20) tv = P.v2[d] * cval;
                            TW2[(jrow - 16 * wave) * D + d] = tv;
                        }
                    }
                    av[s] = (rvalid && 4 * s + q < D) ? aop : 0.;
                }
            }
            wave_lds_fence();                                      // the uniforms are consumed: the slice is free for the transforms' output
        }
        MBX_PHASE(7);                                              // (instrumented builds: draws + mutation of wave 0)
        // ---- z = M1 (x - dshift) on the float64 matrix cores (Gallagher: M1 x, the peaks are pre-rotated): the fma chain of matvec_rows_mfma.
        // The result goes straight to the wave's slice (C / D layout: lane (c, q) holds rows q + 4 r, coordinates c and 16 + c); everything
        // element-wise then happens IN PLACE there, one element per lane at a time (loops that are not unrolled: the transforms are out-of-line
        // calls, and the fewer values are alive across a call the fewer are spilled around it).
        if (!(TWO && (kind == 5 || kind == 20))) lde_map_tile<NP, D>(P.m1, av, TW, wave, c, q);      // (linear slope, Schwefel: no map)
        MBX_PHASE(8);                                              // (first linear map of wave 0)
        if (!(kind == 1 || kind == 13 || gall || (TWO && kind == 5))) lde_tile_transforms<NP, D, KIND, TA>(TW, L.A2, &P, kind, TW2);      // (Sphere-like kinds and Gallagher: nothing element-wise)
        __syncthreads();
        MBX_PHASE(2);                                              // mutation, linear maps, transforms (wave-local)

        // ================================================================ Gallagher: winning peak of every row (block-wide: lane = row, wave = peaks)
        if (gall) lde_gallagher_search<NP, D>(L.TB, (unsigned long long*)L.A1, (int*)L.A2, &P);
        MBX_PHASE(3);                                              // Gallagher peak search
        // ================================================================ row sums + noise -> trial costs; selection (:55-59)
        int surv_mine = 0;
        if (tid < NP) {
            const int j = tid;
            const double* z = L.TB + j * D;
            const double* t = z;
            const double bh = 0.;                                  // boundary penalty of an in-range candidate: exactly +0 (see the header)
            double f;
            switch (kind) {
            case 1: { double s = 0.; for (int d = 0; d < D; ++d) s += z[d] * z[d]; f = s + bias + bh; break; }
            case 2: case 10: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = s + bias + bh; break; }
            case 6: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = m_pow(osc1(s), 0.9) + bias; break; }
            case 7: {
                double s = 0.;
                for (int d = 0; d < D; ++d) s += P.v0[d] * (z[d] * z[d]);
                f = 0.1 * fmax(L.A2[j] / 1e4, s) + bh + bias;
                break;
            }
            case 8: case 9: {
                double s = 0.;
                for (int d = 0; d < D - 1; ++d) {
                    const double a = z[d] * z[d] - z[d + 1];
                    const double bb = z[d] - 1;
                    s += 100 * (a * a) + bb * bb;
                }
                f = s + bias + bh;
                break;
            }
            case 11: { double s = 0.; for (int d = 1; d < D; ++d) s += t[d]; f = 1000000 * t[0] + s + bias; break; }
            case 12: { double s = 0.; for (int d = 1; d < D; ++d) s += 1000000 * (z[d] * z[d]); f = z[0] * z[0] + s + bias; break; }
            case 13: { double s = 0.; for (int d = 1; d < D; ++d) s += z[d] * z[d]; f = z[0] * z[0] + 100. * sqrt(s) + bias; break; }
            case 14: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = m_pow(s, 0.5) + bias + bh; break; }
            case 16: {
                double acc = 0.; for (int d = 0; d < D; ++d) acc += t[d];
                f = 10 * m_pow(acc / D - P.s[0], 3) + 10. / D * bh + bias;
                break;
            }
            case 17: case 18: {
                double acc = 0.; for (int d = 0; d < D - 1; ++d) acc += t[d];
                f = m_pow(1. / (D - 1) * acc, 2) + bh + bias;
                break;
            }
            case 19: {
                double acc = 0.; for (int d = 0; d < D - 1; ++d) acc += t[d];
                f = P.s[0] + P.s[0] * acc / (D - 1.) + bias + bh;
                break;
            }
            case 21: case 22: {
                const int ks = ((const int*)L.A2)[j];
                const double* __restrict__ ry = P.pyr + ks * D;
                const double* __restrict__ ck = P.pc + ks * D;
                double acc = 0.;
                for (int d = 0; d < D; ++d) { const double zd = z[d] - ry[d]; acc += ck[d] * (zd * zd); }
                const double best = P.pw[ks] * m_exp((-0.5 / D) * acc);
                const double o = osc1(10 - best);
                f = o * o + bias + bh;
                break;
            }
            case 23: {
                double res = 1.; for (int d = 0; d < D; ++d) res *= t[d];
                const double tmp = 10. / D / D;
                f = res * tmp - tmp + bh + bias;
                break;
            }
            default:
                f = NAN;
                if constexpr (TWO) {                               // the two-array kinds (eval_rows' row phase, same order of the sums)
                    const double* t2 = L.TB2 + j * D;
                    if (kind == 3 || kind == 15 || kind == 4) {
                        double sc = 0., sq = 0.;
                        for (int d = 0; d < D; ++d) { sc += t2[d]; sq += z[d] * z[d]; }
                        f = kind == 4 ? 10. * (D - sc) + sq + 100 * bh + bias : 10. * (D - sc) + sq + bias;
                    } else if (kind == 5) {
                        double s = 0.; for (int d = 0; d < D; ++d) s += t2[d];
                        f = s + bias;
                    } else if (kind == 20) {
                        double acc = 0., pen = 0.;
                        for (int d = 0; d < D; ++d) { acc += z[d]; pen += t2[d]; }
                        f = 4.189828872724339 - 0.01 * (acc / D) + 100 * pen + bias;
                    } else if (kind == 24) {
                        const double mu0 = P.s[0], sc_ = P.s[1], mu1 = P.s[2];
                        double a = 0., bq = 0., sc = 0.;
                        for (int d = 0; d < D; ++d) {
                            const double xh = P.v0[d] * t2[d];      // t2 = the candidate
                            a += (xh - mu0) * (xh - mu0);
                            bq += (xh - mu1) * (xh - mu1);
                            sc += t[d];
                        }
                        f = fmin(a, D + sc_ * bq) + 10. * (D - sc) + 1e4 * bh + bias;
                    }
                }
                break;
            }
            const RowPost post{&rng, nullptr, MBX_SITE_NOISE0_A, MBX_SITE_NOISE0_B, NP};
            const double nc = (MBX_LDE_ABL & 16) ? f - P.optimum : row_post(P, post, RKa[j], f);        // the noise draw of an individual is indexed by its position in the order
            surv_mine = nc <= L.FIT[j];
            if (surv_mine) L.FIT[j] = nc;
            RKb[j] = 0;                                            // accumulator of the next ranking
        }
        // the survivors' flags: one ballot per wave, read by the lanes that own the rows (row j's flag is bit j & 63 of wave j >> 6's ballot)
        {
            const unsigned long long bal = __ballot(surv_mine);
            if (lane == 0) ((unsigned long long*)L.RED)[wave] = bal;
        }
        __syncthreads();
        MBX_PHASE(4);                                              // row sums, noise, selection
        // ---- survivors take the trial vector (parked in HBM by the lanes that built it, read back by the same lanes); stable ranking of the new fitness values
        {
            const unsigned long long bal = ((const unsigned long long*)L.RED)[jrow >> 6];
            if (rvalid && ((bal >> (jrow & 63)) & 1ull)) {
                double tv[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) tv[s] = park[jrow * D + kq[s]];        // written by this very lane in the tile phase (and, identically, by the lanes clamped to it)
#pragma unroll
                for (int s = 0; s < KS; ++s) if (4 * s + q < D) L.P[jrow * D + 4 * s + q] = tv[s];
            }
            // rank = number of individuals that are better, or equally good and earlier in the previous order (the stable sort of __order_by_f).
            // Thread (i, part) counts over a slice of j; two equal fitness values are rare (a collapsed population), so the first pass counts
            // `<` and `==` only and the order of the equal ones is looked up when there are any.
            constexpr int parts = (64 * TILES) / NP > 0 ? (64 * TILES) / NP : 1;
            for (int w2 = tid; w2 < ((MBX_LDE_ABL & 32) ? 0 : parts * NP); w2 += MBX_NT) {
                const int part = w2 / NP, i = w2 - part * NP;
                const int j0 = part * NP / parts, j1 = (part + 1) * NP / parts;
                const double fi = L.FIT[i];
                int cnt = 0, eq = 0;
#pragma unroll 5
                for (int j = j0; j < j1; ++j) { const double fj = L.FIT[j]; cnt += fj < fi; eq += fj == fi; }
                if (eq > (i >= j0 && i < j1 ? 1 : 0)) {
                    const int ri = RKa[i];
                    for (int j = j0; j < j1; ++j) cnt += (L.FIT[j] == fi && RKa[j] < ri);
                }
                if (cnt) atomicAdd(&RKb[i], cnt);
            }
            if (tid < MBX_LDE_BINS) { L.HS[tid] += (double)my_hist; L.HIST[tid] = 0; }
        }
        __syncthreads();
        if (tid < NP) { const int r = (MBX_LDE_ABL & 32) ? RKa[tid] : RKb[tid]; L.RK[tid] = (uint8_t)r; L.ORDER[r] = (uint8_t)tid; L.A1[r] = L.FIT[tid]; }
        // the next generation's policy input [x | h] as float32 in the (idle) tile array: h here, the features by the threads that make them
        if (tid >= MBX_NT - 64) { float* XS = (float*)L.TB; const int k = IN + (tid - (MBX_NT - 64)); if (k < 4 * ((K1 + 3) / 4)) XS[k] = k < K1 ? L.HC[k - IN] : 0.f; }
        __syncthreads();
        MBX_PHASE(5);                                              // survivors, ranking, order
        // ---- features of the new state (:145-157) and the bookkeeping of update() (:170-198).  The bookkeeping needs the best fitness only: lane 0 of the LAST wave (no
        // rows of the normalisation there) runs it beside the normalisation instead of behind it.
        if (tid == MBX_NT - 64) {
            const double bsf_next = L.A1[0], bsf_cur = L.SCAL[LR_BSF], fes_next = fes + NP;
            const double reward = (bsf_cur - bsf_next) / bsf_cur;   // :170
            int log_index = (int)L.SCAL[LR_LOGI], cost_len = (int)L.SCAL[LR_CLEN];
            double* sc = ar.bp.state + (int64_t)b * ar.bp.state_stride + MBX_LDE_ST_SCALARS(NP, D);
            bool dn;
            {   // log_and_terminate (mbx_rlepso.hpp) on the argument block's fields
                double* cost = sc + MBX_NSCALAR;
                if (fes_next >= (double)log_index * ar.bp.log_interval) { log_index += 1; cost[cost_len++] = bsf_next; }
                dn = fes_next >= ar.bp.max_fes;
                if (!isnan(P.optimum) && ar.bp.early_stop) dn = dn || bsf_next <= 1e-8;
                if (dn) {
                    if (cost_len >= ar.bp.n_logpoint + 1) cost[cost_len - 1] = bsf_next;
                    else cost[cost_len++] = bsf_next;
                }
            }
            L.SCAL[LR_FES] = fes_next; L.SCAL[LR_HCOUNT] += 1.; L.SCAL[LR_BSF] = bsf_next;      // (fes is in every thread's registers; the new history length is read below, behind the barrier)
            L.SCAL[LR_RSUM] += reward; L.SCAL[LR_RTOT] += reward; L.SCAL[LR_LOGI] = log_index; L.SCAL[LR_CLEN] = cost_len;
            L.FLAG[0] = dn ? 1 : 0;
            if (ar.out.traj_reward) ar.out.traj_reward[(int64_t)it * ar.bp.B + b] = reward;
            if (ar.out.traj_done) ar.out.traj_done[(int64_t)it * ar.bp.B + b] = dn ? 1 : 0;
        }
        lde_norm_hist<true>(L.A1, NP, L.FEAT, L.HIST, (float*)L.TB);      // (HIST cleared before the ranking's barrier)
        if (tid < MBX_LDE_BINS) {
            float* XS = (float*)L.TB;
            const double hcount = L.SCAL[LR_HCOUNT];
            const double f1 = (double)L.HIST[tid], f2 = L.HS[tid] / hcount;
            L.FEAT[NP + tid] = f1; L.FEAT[NP + MBX_LDE_BINS + tid] = f2;
            XS[NP + tid] = (float)f1; XS[NP + MBX_LDE_BINS + tid] = (float)f2;
        }
        if (tid == 64) L.HS[MBX_LDE_BINS] = lde_pack_hist(L.HIST);
        __syncthreads();
        if (ar.out.traj_state) for (int k = tid; k < NF; k += MBX_NT) ar.out.traj_state[((int64_t)it * ar.bp.B + b) * NF + k] = L.FEAT[k];
        MBX_PHASE(6);                                              // features, bookkeeping
        executed = it + 1;
        if (L.FLAG[0] != 0) break;
    }
    return executed;
}

// TA: the instantiation carries the second tile array (and with it the six kinds whose row sums read two arrays or the candidate): always at D <= 16; at D = 30 an
// instantiation of its own for the reference's NP = 50 on plain bbob (38.9 KB of LDS: four workgroups per CU instead of six -- config 3's bbob-noisy batches keep the lean one)
template <int NPC, int DC, int HC_ = 50, bool TA = lde_run_two_arrays(DC)>
__global__ __launch_bounds__(64 * ((NPC + 15) / 16)) __attribute__((amdgpu_waves_per_eu(MBX_LDE_RUN_WAVES(NPC))))
void k_lde_run(LdeRunArgs args_)
{
    (void)args_;                                                   // read through lde_run_args()
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NP = NPC, D = DC, NE = NP * D, H = HC_, IN = NP + 2 * MBX_LDE_BINS, K1 = IN + H, NF = NP + 2 * MBX_LDE_BINS;
    static_assert(D <= 32 && D % 2 == 0, "one or two 16-column tiles, coordinate pairs");
    const LdeRunLds L = lde_run_carve(smem, NP, D, H, TA);
    int b, gen0, episode, n_gens, kind0;
    uint32_t seed_lo, seed_hi;
    {
        LdeRunCArgs& ar = lde_run_args();
        const int tid = threadIdx.x;
        b = __builtin_amdgcn_readfirstlane(ar.bp.order[blockIdx.x]);
        n_gens = ar.n_gens;
        { const uint64_t sd = ar.bp.seeds[b]; seed_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sd); seed_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sd >> 32)); }
        double* S = ar.bp.state + (int64_t)b * ar.bp.state_stride;
        double* sc = S + MBX_LDE_ST_SCALARS(NP, D);
        if (sc[MBX_SC_DONE] != 0.) {                               // finished before this launch: state_out keeps the last features
            if (tid == 0) {
                if (ar.out.reward_out) ar.out.reward_out[b] = 0.;
                if (ar.out.done_out) ar.out.done_out[b] = 1;
                for (int g = 0; g < n_gens; ++g) {
                    if (ar.out.traj_reward) ar.out.traj_reward[(int64_t)g * ar.bp.B + b] = 0.;
                    if (ar.out.traj_done) ar.out.traj_done[(int64_t)g * ar.bp.B + b] = 1;
                }
            }
            return;
        }
        episode = __builtin_amdgcn_readfirstlane((int)sc[MBX_SC_EPISODE]);
        gen0 = __builtin_amdgcn_readfirstlane((int)sc[MBX_SC_GEN]);
        // ---- the state block, once
        const double* gPop = S + MBX_LDE_ST_POP(NP, D);
        for (int e = tid; e < NE; e += MBX_NT) L.P[e] = gPop[e];
        for (int i = tid; i < NP; i += MBX_NT) {
            const double f = S[MBX_LDE_ST_FIT(NP, D) + i];
            L.FIT[i] = f; L.A1[i] = f;                             // sorted in HBM: SORTED == FIT
            L.RK[i] = (uint8_t)i; L.ORDER[i] = (uint8_t)i;
        }
        if (tid < 8) L.HS[tid] = S[MBX_LDE_ST_HSUM(NP, D) + tid];
        {
            const DevProblem* pr = ar.bp.problems + ar.bp.problem_idx[b];
            kind0 = __builtin_amdgcn_readfirstlane(pr->kind);
            const double* dsh = pr->dshift;
            if (tid < D) L.DSH[tid] = (dsh && !(pr->kind == 21 || pr->kind == 22)) ? dsh[tid] : 0.;      // Gallagher: no shift (peaks pre-rotated)
        }
        for (int k = tid; k < NF; k += MBX_NT) L.FEAT[k] = ar.state_in[(int64_t)b * NF + k];
        if (tid < H) { L.HC[tid] = ar.hbuf[(int64_t)b * H + tid]; L.HC[H + tid] = ar.cbuf[(int64_t)b * H + tid]; }
        {   // the policy's input [x | h] as float32 (afterwards the feature phase of a generation stages the next one's)
            float* XS = (float*)L.TB;
            for (int k = tid; k < 4 * ((K1 + 3) / 4); k += MBX_NT) XS[k] = k < IN ? (float)ar.state_in[(int64_t)b * NF + k] : (k < K1 ? ar.hbuf[(int64_t)b * H + (k - IN)] : 0.f);
        }
        if (tid == 0) {
            L.FLAG[0] = 0;
            L.SCAL[LR_FES] = sc[MBX_SC_FES]; L.SCAL[LR_HCOUNT] = sc[MBX_SC_HCOUNT]; L.SCAL[LR_BSF] = S[MBX_LDE_ST_FIT(NP, D)];
            L.SCAL[LR_RSUM] = 0.; L.SCAL[LR_RTOT] = sc[MBX_SC_RETURN]; L.SCAL[LR_LOGI] = sc[MBX_SC_LOG_INDEX]; L.SCAL[LR_CLEN] = sc[MBX_SC_COST_LEN];
        }
    }
    __syncthreads();
    int executed;
    const uint64_t karg = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
#define MBX_LDE_GENS(K) lde_run_generations<NPC, DC, HC_, K, TA>(b, gen0, episode, n_gens, seed_lo, seed_hi, (uint32_t)karg, (uint32_t)(karg >> 32))
    switch (kind0) {                                               // the kinds of the noisy suite (problem/bbob.py: _NOISY); anything else takes the any-kind loop
    case 1: executed = MBX_LDE_GENS(1); break;
    case 7: executed = MBX_LDE_GENS(7); break;
    case 8: executed = MBX_LDE_GENS(8); break;
    case 10: executed = MBX_LDE_GENS(10); break;
    case 14: executed = MBX_LDE_GENS(14); break;
    case 17: executed = MBX_LDE_GENS(17); break;
    case 19: executed = MBX_LDE_GENS(19); break;
    case 21: executed = MBX_LDE_GENS(21); break;
    default: executed = MBX_LDE_GENS(0); break;
    }
#undef MBX_LDE_GENS
    executed = __builtin_amdgcn_readfirstlane(executed);

    // ---- the state block, once: population and fitness in fitness order, like k_lde_step leaves them
    {
        LdeRunCArgs& ar = lde_run_args();
        const int tid = threadIdx.x;
        double* S = ar.bp.state + (int64_t)b * ar.bp.state_stride;
        double* sc = S + MBX_LDE_ST_SCALARS(NP, D);
        const uint8_t* RK = L.RK;
        double* gPop = S + MBX_LDE_ST_POP(NP, D);
        const FastDiv fd(D);
        for (int e = tid; e < NE; e += MBX_NT) { const int j = fd.div(e), d = e - j * D; gPop[RK[j] * D + d] = L.P[e]; }
        for (int i = tid; i < NP; i += MBX_NT) S[MBX_LDE_ST_FIT(NP, D) + i] = L.A1[i];
        if (tid < 8) S[MBX_LDE_ST_HSUM(NP, D) + tid] = L.HS[tid];
        for (int k = tid; k < NF; k += MBX_NT) ar.out.state_out[(int64_t)b * NF + k] = L.FEAT[k];
        if (tid < H) { ar.hbuf[(int64_t)b * H + tid] = L.HC[tid]; ar.cbuf[(int64_t)b * H + tid] = L.HC[H + tid]; }
        if (tid == 0) {
            const bool done = L.FLAG[0] != 0;
            sc[MBX_SC_GBEST] = L.SCAL[LR_BSF]; sc[MBX_SC_FES] = L.SCAL[LR_FES]; sc[MBX_SC_LOG_INDEX] = L.SCAL[LR_LOGI]; sc[MBX_SC_COST_LEN] = L.SCAL[LR_CLEN];
            sc[MBX_SC_DONE] = done ? 1. : 0.; sc[MBX_SC_RETURN] = L.SCAL[LR_RTOT]; sc[MBX_SC_GEN] = gen0 + executed; sc[MBX_SC_HCOUNT] = L.SCAL[LR_HCOUNT];
            if (ar.out.reward_out) ar.out.reward_out[b] = L.SCAL[LR_RSUM];
            if (ar.out.done_out) ar.out.done_out[b] = done ? 1 : 0;
            for (int g = executed; g < n_gens; ++g) {              // generations after the termination: reward 0, done 1 (like k_rlepso_run's records)
                if (ar.out.traj_reward) ar.out.traj_reward[(int64_t)g * ar.bp.B + b] = 0.;
                if (ar.out.traj_done) ar.out.traj_done[(int64_t)g * ar.bp.B + b] = 1;
            }
        }
    }
}

}  // namespace mbx
