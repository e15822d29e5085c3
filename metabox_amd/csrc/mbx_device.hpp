// mbx_device.hpp — device-side building blocks for gfx950 (wave64): Philox4x32-10, block reductions and
// the block-cooperative BBOB evaluator shared by the stand-alone eval kernel and the fused generation
// kernels.  One workgroup evaluates a whole population that is resident in LDS.
//
// Reference semantics being implemented (cited per function): src/problem/bbob.py of GMC-DRL/MetaBox.
// Arithmetic is float64 and the translation unit is built with -ffp-contract=off so that element-wise
// expressions round like numpy's (no implicit FMA).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mbx.h"
#include "../../include/mbx_layout.h"
#include "mbx_math.hpp"

namespace mbx {

#ifdef MBX_NOINLINE_MATH
#define MBX_MATH __device__ __noinline__
#else
#define MBX_MATH __device__ __forceinline__
#endif
#ifdef MBX_LIBM_MATH
MBX_MATH double m_pow(double a, double b) { return pow(a, b); }
MBX_MATH double m_sin(double a) { return sin(a); }
MBX_MATH double m_cos(double a) { return cos(a); }
MBX_MATH double m_exp(double a) { return exp(a); }
MBX_MATH double m_log(double a) { return log(a); }
#else                                   // range-specialised routines of mbx_math.hpp (library fall-through outside their ranges)
MBX_MATH double m_pow(double a, double b) { return fm::pow_fast(a, b); }
MBX_MATH double m_sin(double a) { return fm::sincos_fast<false>(a); }
MBX_MATH double m_cos(double a) { return fm::sincos_fast<true>(a); }
MBX_MATH double m_exp(double a) { return fm::exp_fast(a); }
MBX_MATH double m_log(double a) { return fm::log_fast(a); }
#endif


constexpr int kThreads = 256;          // 4 waves of 64: the default workgroup
// The block-cooperative helpers take the workgroup size from the launch (256, or 512 for the LDS-bound geometries of mbx_rlepso.hpp)
#define MBX_NT ((int)blockDim.x)
#define MBX_NW ((int)blockDim.x >> 6)
constexpr double kTwoPi = 6.283185307179586;

// ------------------------------------------------------------------------------------------------
// Device copy of a problem (pointers are device addresses inside the suite's constant pool).
// ------------------------------------------------------------------------------------------------
struct DevProblem {
    int32_t func_id, kind, dim, n_peaks, noise_kind;
    int32_t n_close;     // protein: the leading pairs of the list that can come within the 9 A cut-off while |x_k| <= ub (mbx_suite_create); 0 = not computed
    double bias, lb, ub, pen_coef, s[4], noise_a, noise_b, optimum;
    const double *dshift, *m1, *m2, *v0, *v1, *v2, *py, *pc, *pw;
    const double* pyr;   // Gallagher: R y_k, precomputed at upload (mbx_suite_create); protein: pair records [n_pairs][4] = sqrt(e) | q | r | 0
    const double* plogw; // Gallagher: log(w_k), precomputed at upload; protein: the pairs' atoms, int32 i | j << 16 (-1: no pair)
};

// The same record read through the constant address space: every field access is a scalar load (s_load_dword*) of memory the compiler knows
// to be invariant, so a long-running kernel (k_rlepso_run) keeps no copy of the 54-dword record in SGPRs -- with the by-value copy the SGPR file
// overflowed into VGPR lanes and every spilled value came back through a v_readlane, a VALU instruction (861 of them in that kernel).
typedef const DevProblem __attribute__((address_space(4))) ConstProblem;

// Division of small non-negative integers by a loop-invariant divisor without the ~35-instruction software divide:
// e / D == umulhi(e, floor((2^32-1)/D) + 1) for e < 2^20, D <= 256 (checked exhaustively on the host).
struct FastDiv {
    uint32_t m; int D;
    __device__ __forceinline__ explicit FastDiv(int D_) : m(0xFFFFFFFFu / (uint32_t)D_ + 1u), D(D_) {}
    __device__ __forceinline__ int div(int e) const { return (int)__umulhi((uint32_t)e, m); }
    __device__ __forceinline__ int mod(int e) const { return e - div(e) * D; }
};

// ------------------------------------------------------------------------------------------------ Philox
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#ifndef MBX_PHILOX_ROUNDS
#define MBX_PHILOX_ROUNDS 10      // anything else is a timing experiment (tools/kbench.py), never a shipped build
#endif
#pragma unroll
    for (int r = 0; r < MBX_PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)MBX_PHILOX_M0 * c0, p1 = (uint64_t)MBX_PHILOX_M1 * c2;   // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#ifdef MBX_PHILOX_XOR2
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
#else
        // gfx950's three-input bit operation (truth table 0x96 = a ^ b ^ c): one instruction per word instead of two v_xor_b32
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
#endif
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += MBX_PHILOX_W0; k1 += MBX_PHILOX_W1;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ double u32d(uint32_t a) { return (double)a * 2.3283064365386963e-10; }      // a / 2^32 in [0, 1)

struct Rng {                      // per-instance stream: key = seed, counter = (index, site, gen, episode)
    uint32_t k0, k1, gen, episode;
    // uniform_fresh: key and counter words are workgroup-uniform and the caller wants their share of the rounds recomputed at every call.
    // Three of the four counter words and the key are uniform, so the compiler folds their part of every round into scalar constants -- per
    // draw site and generation -- and keeps them all alive across the generation body; in k_rlepso_run they overflow the SGPR file and come
    // back one v_readlane (a VALU instruction) at a time.  Behind an empty asm the scalar unit recomputes them where they are used.
    bool uniform_fresh = false;
    __device__ __forceinline__ U4 draw(uint32_t idx, uint32_t site) const
    {
#ifdef MBX_ABLATE_RNG
        return U4{idx * 2654435761u, site + 0x9E3779B9u * idx, gen ^ (idx << 7), episode + idx};   // timing experiments only
#else
        uint32_t a = k0, b = k1;
        if (uniform_fresh) {
            // (readfirstlane: a no-op when the key already sits in SGPRs; instrumented builds load it with vector loads)
            a = (uint32_t)__builtin_amdgcn_readfirstlane((int)a); b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            asm volatile("" : "+s"(a), "+s"(b));
        }
        return philox4x32_10(idx, site, gen, episode, a, b);
#endif
    }
};

__device__ __forceinline__ void box_muller(double ua, double ub, double& n0, double& n1)
{
    const double r = sqrt(-2.0 * m_log(1.0 - ua));
    const double t = kTwoPi * ub;
    n0 = r * m_cos(t);
    n1 = r * m_sin(t);
}

// noise draws (a,b,c) of evaluation row `row` (include/mbx_layout.h §3)
__device__ __forceinline__ void philox_noise(const Rng& rng, uint32_t row, uint32_t siteA, uint32_t siteB, int noise_kind,
                                             double& a, double& b, double& c)
{
    a = b = c = 0.0;
    if (noise_kind == MBX_NOISE_NONE) return;
    U4 w = rng.draw(row, siteA);
    const double ua = u53(w.x, w.y), ub = u53(w.z, w.w);
    if (noise_kind == MBX_NOISE_GAUSS) {
        a = sqrt(-2.0 * m_log(1.0 - ua)) * m_cos(kTwoPi * ub);       // the cosine half of box_muller: the sine half is not used (an out-of-line call the compiler cannot drop)
    } else if (noise_kind == MBX_NOISE_UNIFORM) {
        a = ua; b = ub;
    } else {
        a = ua;
        w = rng.draw(row, siteB);
        box_muller(u53(w.x, w.y), u53(w.z, w.w), b, c);
    }
}

// NoisyProblem.noisy for one value (bbob.py:108-146)
template <class PT>
__device__ __forceinline__ double apply_noise(const PT& P, double ftrue, double a, double b, double c)
{
    if (P.noise_kind == MBX_NOISE_NONE) return ftrue;
    const double fu = ftrue - P.optimum;
    double fn;
    if (P.noise_kind == MBX_NOISE_GAUSS) {
        fn = fu * m_exp(P.noise_a * a);
    } else if (P.noise_kind == MBX_NOISE_UNIFORM) {
        fn = fu * m_pow(a, P.noise_b) * fmax(1., m_pow(1e9 / (fu + 1e-99), P.noise_a * (0.49 + 1. / P.dim) * b));
    } else {
        fn = fu + P.noise_a * fmax(0., 1e3 + (a < P.noise_b ? 1. : 0.) * b / (fabs(c) + 1e-199));
    }
    return fu >= 1e-8 ? fn + P.optimum + 1.01 * 1e-8 : ftrue;
}

// threadIdx.x behind an empty volatile asm: the block-cooperative helpers below take their thread index through this, so that nothing
// derived from it looks loop-invariant to the compiler.  k_rlepso_run calls them once per generation inside its generation loop; with a
// plain threadIdx.x all their index arithmetic was hoisted out of that loop and spilled (96 - 208 B of scratch per thread, re-read every
// generation: 37 KB of HBM traffic per env-step for a kernel whose state never leaves the chip).  In the one-generation kernels it is a no-op.
__device__ __forceinline__ int opaque_tid()
{
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// ------------------------------------------------------------------------------------------------ reductions
// lexicographic (value, index) minimum across a wave: smallest value, lowest index on ties (np.argmin)
__device__ __forceinline__ void wave_argmin(double& v, int& i)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// argmin over a[0..n) in LDS (first index on ties).  All threads call; result broadcast through red[0..1].
__device__ __forceinline__ void block_argmin(const double* a, int n, double* red, double& vmin, int& imin)
{
    const int tid = opaque_tid();
    if (tid < 64) {
        double v = INFINITY; int idx = 0x7fffffff;
        for (int j = tid; j < n; j += 64) {
            const double x = a[j];
            if (x < v) { v = x; idx = j; }
        }
        wave_argmin(v, idx);
        if (idx >= n) { idx = 0; v = a[0]; }          // every entry +inf (or NaN): np.argmin answers 0; never hand back an out-of-range index
        if (tid == 0) { red[0] = v; red[1] = (double)idx; }
    }
    __syncthreads();
    vmin = red[0]; imin = (int)red[1];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ BBOB pieces

// osc_transform, bbob.py:51-67: sign(x) * exp(y + 0.49 (sin(c1 y) + sin(c2 y))) ** 0.1 with y = log|x| / 0.1.
// The reference's exp(t) ** 0.1 is evaluated as exp(0.1 t): the same real number, one transcendental instead of two
// (a general pow is the most expensive libm call on this path); |t| <= ~750 so the re-association costs < 1e-13
// relative, eight orders of magnitude inside the 1e-5 contract (tests pin it at 1e-10 against the reference's KATs).
// L / 0.1 as the reference writes it (0.1 is not 1/10): one Newton correction of L * fl(1 / 0.1) instead of a full fp64 division
__device__ __forceinline__ double div_by_tenth(double L)
{
    const double q = L * 10.0;
    return __builtin_fma(__builtin_fma(-0.1, q, L), 10.0, q);
}

__device__ __forceinline__ double osc1(double x)
{
    // One code path for both signs: the coordinates of a wave have mixed signs, so an if / else over x > 0 / x < 0 executes BOTH sides
    // (log + exp + 2 sin each) under complementary EXEC masks.  Selecting the two frequencies instead halves the cost; 1.0 * y == y
    // exactly, so the positive side is bit-identical to sin(y).
    if (x == 0.) return x;
    const bool pos = x > 0.;
    const double y = div_by_tenth(m_log(fabs(x)));
    const double c1 = pos ? 1.0 : 0.55, c2 = pos ? 0.79 : 0.31;
    const double r = m_exp(0.1 * (y + 0.49 * (m_sin(c1 * y) + m_sin(c2 * y))));
    return pos ? r : -r;
}

// The same transform as ONE out-of-line call per coordinate with its four transcendentals inline inside (same routines, same arithmetic as the
// four m_* calls of osc1: each call costs argument moves, a jump pair and a full s_waitcnt at the callee's entry), and the Schaffers term
// (bbob.py:642-643) likewise.  A kernel's register count is the maximum over everything it can call, and these two need more than the leaf
// routines (round 2: with them k_lde_step<512, 50, 30> lost its third resident workgroup), so only the evaluator instantiation of the headline
// geometry (eval_rows<10>: k_rlepso_step / k_rlepso_run<256, 100, 10, 5>, 96 VGPRs either way) refers to them: 122.5 -> 119.5 us per generation
// (A/B on one box).  Measured and dropped: the Weierstrass series and T_asy(T_osz(.)) as composites of their own (119.9 -> 120.1 us).
#if defined(MBX_NOINLINE_MATH) && !defined(MBX_LIBM_MATH)
__device__ __noinline__ double osc1_composite(double x)
{
    if (x == 0.) return x;
    const bool pos = x > 0.;
    const double y = div_by_tenth(fm::log_fast(fabs(x)));
    const double c1 = pos ? 1.0 : 0.55, c2 = pos ? 0.79 : 0.31;
    const double r = fm::exp_fast(0.1 * (y + 0.49 * (fm::sincos_fast<false>(c1 * y) + fm::sincos_fast<false>(c2 * y))));
    return pos ? r : -r;
}
__device__ __noinline__ double schaffers_composite(double s)
{
    const double w = fm::sincos_fast<false>(50 * fm::pow_fast(s, 0.2));
    return sqrt(s) * (w * w + 1);                                 // pow(w, 2) is the correctly rounded square in numpy, libm and pow_fast alike
}
#else
__device__ __forceinline__ double osc1_composite(double x) { return osc1(x); }
__device__ __forceinline__ double schaffers_composite(double s) { return sqrt(s) * (m_pow(m_sin(50 * m_pow(s, 0.2)), 2) + 1); }
#endif
#ifndef MBX_COMPOSITE_DC
#define MBX_COMPOSITE_DC 10
#endif
template <bool CMP> __device__ __forceinline__ double osc_t(double x) { if constexpr (CMP) return osc1_composite(x); else return osc1(x); }

__device__ __forceinline__ double asy1(double x, double beta_lin)  // asy_transform, bbob.py:70-82
{
    return x > 0. ? m_pow(x, 1. + beta_lin * sqrt(x)) : x;
}

__device__ __forceinline__ double pen_row(const double* x, int D, double ub)   // pen_func, bbob.py:85-93
{
    double s = 0.;
    for (int d = 0; d < D; ++d) {
        const double t = fmax(0., fabs(x[d]) - ub);
        s += t * t;
    }
    return s;
}

// LDS working set of the block-cooperative evaluator.
// size of the evaluator's T scratch for n rows of dimension D (doubles): n*D, but never less than what the Gallagher
// path needs for one chunk of peak tables (>= 8 peaks) and its per-wave partial results (8 per row)
__host__ __device__ inline long long eval_t_doubles(int n, int D)
{
    long long t = (long long)n * D;
    if (t < 17ll * D) t = 17ll * D;
    if (t < 8ll * n) t = 8ll * n;
    return (t + 1) & ~1ll;
}

struct EvalLds {
    const double* X;       // [n*D]  candidate positions (input)
    double *Z, *T;         // [max(n*D, 512)], [eval_t_doubles(n, D)]  scratch
    double *M1T, *M2T;     // [D*D]  linear maps, transposed: MT[k*D+d] = M[d][k]
    double *DSH, *V0, *V1, *V2;   // [D] per-problem vectors
    double* F;             // [n]    objective values (output)
    int z_doubles = 0;     // capacity of Z when the caller says so (0: unknown -- only the guaranteed max(n*D, 512) is used); eval_rows_protein's wave-per-row mode needs it
};

// Stage the per-problem constants (two D x D maps, transposed so that lanes differing in d read consecutive
// LDS words, and the four D-vectors) from global memory into LDS.  Caller synchronises afterwards.
// DC: the dimension as a compile-time constant (0 = P.dim), for the compile-time-geometry instantiations of the generation kernels
#ifndef MBX_EVAL_DC_MAX
#define MBX_EVAL_DC_MAX 16
#endif
// (only small dimensions are handed down: with D = 30 / 40 the fully unrolled D-loops of the evaluator spill registers)
constexpr int eval_dc(int dc) { return dc <= MBX_EVAL_DC_MAX ? dc : 0; }
// MD: the dimension handed to the scalar-operand matvec (matvec_rows_scalar) -- the large compile-time dimensions; 0 = LDS tile route.
#ifndef MBX_EVAL_MD_MIN
#define MBX_EVAL_MD_MIN 16
#endif
constexpr int eval_md(int dc) { return dc >= MBX_EVAL_MD_MIN ? dc : 0; }
// MAPS = false: the kernel's matvec reads the maps from global memory (MD > 0), nothing to stage for them
template <int DC = 0, bool MAPS = true, class PT = DevProblem>
__device__ __forceinline__ void stage_problem(const PT& P, const EvalLds& L)
{
    const int D = DC ? DC : P.dim;
    const FastDiv fd(D);
    for (int t = threadIdx.x; MAPS && t < D * D; t += MBX_NT) {
        const int d = fd.div(t), k = t - d * D;
        if (P.m1) L.M1T[k * D + d] = P.m1[t];
        if (P.m2) L.M2T[k * D + d] = P.m2[t];
    }
    for (int d = threadIdx.x; d < D; d += MBX_NT) {
        L.DSH[d] = P.dshift ? P.dshift[d] : 0.;
        L.V0[d] = P.v0 ? P.v0[d] : 0.;
        L.V1[d] = P.v1 ? P.v1[d] : 0.;
        L.V2[d] = P.v2 ? P.v2[d] : 0.;
    }
}

// Out[i][d] = sum_k M[d][k] * (In[i][k] - sub[k])   (sr_func, bbob.py:6-8: shift, then the matmul) as ONE fused multiply-add per term:
// acc = +0; acc = fma(M[d][k], y_k, acc) for k ascending.  Every matvec variant below (and the CPU oracle) evaluates exactly this chain, so
// they stay bit-identical to each other; it is also, bit for bit, what v_mfma_f64_16x16x4_f64 computes (tools/ubench/mfma_f64_probe.hip:
// 76 800 outputs, K = 12 chained, none differs), and the reference's np.matmul is FMA-based BLAS itself.  Half the instructions of the
// separate multiply + add of rounds 1-2.
// sub == nullptr: plain matvec.  Subtracting inside the product loop instead of in a pass of its own saves a barrier interval per
// evaluation (in the short phases of the generation kernels an interval costs more than the NP*D subtractions repeated per output pair).
template <bool SUB>
__device__ __forceinline__ void matvec_rows_impl(const double* MT, const double* In, const double* sub, int n, int D, double* Out)
{
    const int tid0 = opaque_tid();
    if ((D & 1) == 0) {
        // Register tile of 2 rows x 2 dimensions per thread: the plain loop below reads 16 bytes of LDS per multiply-add (M[d][k] and
        // In[i][k]) and is LDS-bandwidth bound (128 B/clk per CU feed 8 MAC/clk, the four SIMDs could issue 32); the tile reads one
        // 16-byte pair of M and two row values per four multiply-adds, 8 bytes each.  Every output keeps its own k-ascending fma chain.
        const int HD = D >> 1, tiles = ((n + 1) >> 1) * HD;
        const FastDiv fh(HD);
        for (int t = tid0; t < tiles; t += MBX_NT) {
            const int rp = fh.div(t), d = 2 * (t - rp * HD), i0 = 2 * rp;
            const bool two = i0 + 1 < n;
            const double* r0 = In + i0 * D;
            const double* r1 = two ? r0 + D : r0;
            const double* col = MT + d;
            double s00 = 0., s01 = 0., s10 = 0., s11 = 0.;
#pragma unroll 5
            for (int k = 0; k < D; ++k) {
                const double m0 = col[k * D], m1 = col[k * D + 1];
                double y0 = r0[k], y1 = r1[k];
                if (SUB) { const double sh = sub[k]; y0 = y0 - sh; y1 = y1 - sh; }
                s00 = __builtin_fma(m0, y0, s00); s01 = __builtin_fma(m1, y0, s01); s10 = __builtin_fma(m0, y1, s10); s11 = __builtin_fma(m1, y1, s11);
            }
            Out[i0 * D + d] = s00; Out[i0 * D + d + 1] = s01;
            if (two) { Out[(i0 + 1) * D + d] = s10; Out[(i0 + 1) * D + d + 1] = s11; }
        }
        return;
    }
    const int NE = n * D;
    const FastDiv fd(D);
    for (int e = tid0; e < NE; e += MBX_NT) {
        const int i = fd.div(e), d = e - i * D;
        const double* row = In + i * D;
        const double* col = MT + d;
        double s = 0.;
#pragma unroll 5
        for (int k = 0; k < D; ++k) s = __builtin_fma(col[k * D], SUB ? row[k] - sub[k] : row[k], s);
        Out[e] = s;
    }
}
__device__ __forceinline__ void matvec_rows(const double* MT, const double* In, int n, int D, double* Out) { matvec_rows_impl<false>(MT, In, nullptr, n, D, Out); }
__device__ __forceinline__ void matvec_rows_shifted(const double* MT, const double* In, const double* sub, int n, int D, double* Out) { matvec_rows_impl<true>(MT, In, sub, n, D, Out); }

// The same product for a compile-time dimension MD >= 16, organised for the large-D geometries (config 3: D = 30, config 5: D = 40) whose
// tile loop above is LDS-bandwidth bound (48 bytes of LDS per 4 multiply-adds at 128 B/clk per CU: 12.8 k cycles per map at NP = 128, D = 40
// against 6.4 k of VALU issue):  lane = row, the row's MD values live in registers, a wave owns a block of consecutive outputs d of one
// 64-row group, and M[d][k] -- wave-uniform -- is read straight from the problem's row-major map in global memory through the constant
// address space, i.e. as scalar loads into SGPRs that v_mul_f64 takes as an operand.  No LDS read in the inner loop, no LDS copy of the
// maps at all.  Each output keeps its own k-ascending fma chain: bit-identical to matvec_rows_impl.
template <int MD, bool SUB>
__device__ __forceinline__ void matvec_rows_scalar(const double* __restrict__ Mg, const double* In, const double* sub, int n, double* Out)
{
    typedef const double __attribute__((address_space(4)))* kptr;
    const kptr M = (kptr)Mg;
    const int tid0 = opaque_tid(), lane = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), nw = __builtin_amdgcn_readfirstlane(MBX_NW);
    const int units = ((n + 63) >> 6) * MD;                      // unit u = (row group u / MD, output u % MD)
    const int per = (units + nw - 1) / nw;
    const int u0 = wave * per, u1 = u0 + per < units ? u0 + per : units;
    double y[MD];
    int loaded = -1;
    for (int u = u0; u < u1; ++u) {
        const int g = u / MD, d = u - g * MD;
        const int i = g * 64 + lane;
        if (g != loaded) {
            loaded = g;
            if (i < n) {
#pragma unroll
                for (int k = 0; k < MD; ++k) { const double v = In[i * MD + k]; y[k] = SUB ? v - sub[k] : v; }
            }
        }
        const kptr row = M + d * MD;
        double acc = 0.;
#pragma unroll
        for (int k = 0; k < MD; ++k) acc = __builtin_fma(row[k], y[k], acc);
        if (i < n) Out[i * MD + d] = acc;
    }
}

// The same scheme for kernels with a tight register budget (k_lde_step<512, 50, 30>: 80 VGPRs): the row is taken in chunks of KC values and up to
// UMAX outputs of the wave are accumulated side by side, so 2 KC + 2 UMAX registers replace the 2 MD of matvec_rows_scalar.  Every output still runs
// its fma chain in ascending k: bit-identical.
#ifndef MBX_UMAX40
#define MBX_UMAX40 5      // outputs accumulated side by side at D = 40 (A/B on one box with KC = 8: 4 / 5 / 8 -> see DESIGN.md)
#endif
template <int MD, bool SUB, int KC, int UMAX>
__device__ __forceinline__ void matvec_rows_scalar_kc(const double* __restrict__ Mg, const double* In, const double* sub, int n, double* Out)
{
    static_assert(MD % KC == 0, "the chunk must divide the dimension");
    typedef const double __attribute__((address_space(4)))* kptr;
    const kptr M = (kptr)Mg;
    const int tid0 = opaque_tid(), lane = tid0 & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), nw = __builtin_amdgcn_readfirstlane(MBX_NW);
    const int units = ((n + 63) >> 6) * MD;
    const int per = (units + nw - 1) / nw;
    const int u0 = wave * per, u1 = u0 + per < units ? u0 + per : units;
    for (int ub = u0; ub < u1;) {
        const int g = ub / MD, ge = (g + 1) * MD, ue = u1 < ge ? u1 : ge;      // this wave's units inside row group g: [ub, ue)
        const int i = g * 64 + lane;
        for (int us = ub; us < ue; us += UMAX) {
            const int cnt = ue - us < UMAX ? ue - us : UMAX, d0 = us - g * MD;
            double acc[UMAX];
#pragma unroll
            for (int j = 0; j < UMAX; ++j) acc[j] = 0.;
#pragma unroll 1
            for (int c = 0; c < MD; c += KC) {
                double y[KC];
                if (i < n) {
#pragma unroll
                    for (int k = 0; k < KC; ++k) { const double v = In[i * MD + c + k]; y[k] = SUB ? v - sub[c + k] : v; }
                }
#pragma unroll
                for (int j = 0; j < UMAX; ++j) {
                    if (j < cnt) {
                        const kptr row = M + (d0 + j) * MD + c;
#pragma unroll
                        for (int k = 0; k < KC; ++k) acc[j] = __builtin_fma(row[k], y[k], acc[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < UMAX; ++j)
                if (j < cnt && i < n) Out[i * MD + d0 + j] = acc[j];
        }
        ub = ue;
    }
}

// The same product on the float64 matrix cores: v_mfma_f64_16x16x4_f64, rows = population rows (A: lane l holds In[16 rt + (l & 15)][4 s + (l >> 4)] - sub),
// columns = outputs (B: lane l holds M[16 ct + (l & 15)][4 s + (l >> 4)], straight from the problem's row-major map in global memory / L2), the
// accumulator tile starts at +0 and is chained over the ceil(MD / 4) k-steps.  The instruction IS the fma chain in ascending k, bit for bit
// (tools/ubench/mfma_f64_probe.hip), and the zero padding of k >= MD adds fma(0, 0, acc) = acc: results are identical to matvec_rows_scalar.
// On MI355X the float64 matrix peak equals the float64 vector peak (78.6 TFLOP/s: 64 cycles per instruction = the 16 v_fma_f64 it replaces), so
// this is not a faster multiplier but a SECOND pipe: the ~40 VALU instructions per 16 x 16 tile (fragment loads, shift, stores) replace ~300, and
// the matrix pipe works while the other resident workgroups' waves keep the vector pipe busy.  Tile (rt, ct) -> wave (rt CT + ct) mod waves.
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int MD, bool SUB>
__device__ __forceinline__ void matvec_rows_mfma(const double* __restrict__ Mg, const double* In, const double* sub, int n, double* Out)
{
    constexpr int KS = (MD + 3) / 4, CT = (MD + 15) / 16;
    const int tid0 = opaque_tid(), lane = tid0 & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), nw = __builtin_amdgcn_readfirstlane(MBX_NW);
    const int units = ((n + 15) >> 4) * CT;
    double sh[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) { const int k = 4 * s + q; sh[s] = (SUB && k < MD) ? sub[k] : 0.; }
    for (int u = wave; u < units; u += nw) {
        const int rt = u / CT, ct = u - rt * CT;
        const int row = 16 * rt + c, d = 16 * ct + c;
        double a[KS], b[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + q;
            b[s] = (d < MD && k < MD) ? Mg[d * MD + k] : 0.;
            a[s] = (row < n && k < MD) ? In[row * MD + k] - sh[s] : 0.;
        }
        f64x4 acc = {0., 0., 0., 0.};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
        if (d < MD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int orow = 16 * rt + q + 4 * r;              // C / D layout of the float64 shape: row (l >> 4) + 4 r, column l & 15
                if (orow < n) Out[orow * MD + d] = acc[r];
            }
        }
    }
}

// sum of v over the block (all threads call; result to every thread).  red: >= 16 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (MBX_NW == 1) return v;                                       // a single wave: every lane already holds the sum, nothing to exchange
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    double s;
    if (MBX_NW < 4) {                                                // two or three waves (k_dq_step's 128-thread workgroups)
        s = red[0];
        for (int g = 1; g < MBX_NW; ++g) s = s + red[g];
    } else {
        s = (red[0] + red[1]) + (red[2] + red[3]);                   // 4 waves: the association the oracle-pinned protein energy uses
        for (int g = 4; g < MBX_NW; g += 4) s = s + ((red[g] + red[g + 1]) + (red[g + 2] + red[g + 3]));
    }
    __syncthreads();
    return s;
}

// Protein-docking energy (src/problem/protein_docking.py:28-48) of n individuals X[n][D] -> F[n].  The individuals
// are processed one after the other; for each, the 3n displaced coordinates are built cooperatively in LDS, then the atom pairs are
// spread over the block and the energy mean_j sum_i term_ij = (sum of all terms) / n is reduced with wave shuffles.
// P.v0 = 1/sqrt(eigval), P.py = basis [D,3n], P.pc = coor_init [3n], P.pw = sqrt(e) | q | r tables.
//
// Only the pairs i < j are visited.  The three tables are symmetric by construction (protein_docking.py:175-181: q = q^T q, e = sqrt(e^T e),
// r = (r + r^T) / 2) and so is the distance, hence term_ij = term_ji; on the diagonal the distance is sqrt(0.01) = 0.1, outside both
// distance windows (0.11, 7) and (7, 9), so term_ii = 0 exactly.  The sum over all n^2 pairs is therefore 2 x the sum over the n (n - 1) / 2
// pairs above the diagonal (the doubling is exact) -- 4950 instead of 10^4 pair terms at n = 100.  Pair index t of the folded rectangle
// ceil(n / 2) x (n - 1): row a holds the n - 1 - a pairs of atom a followed by the a pairs of atom n - 1 - a.
// sqrt and the two divisions by the distance (the reference's `r / pair_dis` and `q / (4 pair_dis)`) come from ONE v_rsq_f64 estimate:
// two coupled Goldschmidt steps give sqrt(s) and 1 / (2 sqrt(s)), one residual correction each brings both to <= 1 ulp -- 13 instructions
// instead of a library sqrt and two IEEE divisions (~70).  The energy is pinned at 1e-9 relative against the reference's own outputs.
// pairs in flight per lane in eval_rows_protein: 4 for the D = 12 instantiation (k_dq_step<100, 12>, two waves per instance), the plain loop elsewhere
// (the generation kernels that can meet a protein problem run under a 96-register cap)
#ifndef MBX_PROTEIN_PF
#define MBX_PROTEIN_PF 4
#endif
// Gallagher peak search with a block of peaks as accumulators (eval_rows): the instantiations that have the registers for it -- under the 80-VGPR
// cap of k_lde_step<512, 50, 30> it spilled so badly that the whole kernel ran 6x slower (615 -> 3680 us per generation), and k_lde_step<512, 100, 30>
// (128 VGPRs, two workgroups per CU) lost 17 % (1123 -> 1314 us), and config 5's resident kernel (velocities / pbest positions in registers next to it)
// 33 % (1.87 -> 2.50 ms): only config 5's one-generation kernel k_rlepso_step<1024, 128, 40, 5> takes it (Gallagher-101 evaluation 203 k -> 83 k cycles)
constexpr bool gallagher_blocked(int md, int kc) { return md == 40 && kc == 0; }
#ifndef MBX_GALLAGHER_LEAN
#define MBX_GALLAGHER_LEAN 2
#endif
constexpr bool gallagher_lean(int md, int kc) { return MBX_GALLAGHER_LEAN && (md == 30 || (MBX_GALLAGHER_LEAN >= 2 && md == 40 && kc > 0)); }

// which compile-time geometries take the matrix-core matvec (matvec_rows_mfma): MBX_MFMA_MATVEC = 0 none, 1 the D = 30 kernels (LDE, RLEPSO --dim 30),
// 2 also D = 40 (config 5)
#ifndef MBX_MFMA_MATVEC
#define MBX_MFMA_MATVEC 1
#endif
constexpr bool mfma_matvec(int md, int kc) { (void)kc; return (MBX_MFMA_MATVEC >= 1 && md == 30) || (MBX_MFMA_MATVEC >= 2 && md == 40); }

constexpr int protein_prefetch(int dc) { return dc == 12 ? MBX_PROTEIN_PF : 1; }

// PF: atom pairs per lane and loop iteration.  PF = 1 is the plain loop.  With PF > 1 the body fetches PF pairs and walks their PF independent
// dependency chains side by side, one arithmetic step at a time over all of them: k_dq_step runs one or two waves per instance with 2-4 waves per
// SIMD, and a pair term is a chain of ~40 dependent float64 operations (rsq, Goldschmidt steps, r^12) that a single wave issues at the
// chain's latency -- 11-13 cycles per instruction (instrumented build: 1070 cycles per pair, the energy 93 k of the step's 152 k cycles).
// Same pairs per lane in the same order: every PF gives the same sums.
// The pair terms of one candidate whose atom records are in ATOM, for the caller's share of the pair list: pairs first, first + stride, ... (stride = the number of lanes that
// share the candidate: the workgroup, or one wave), PF of them in flight; returns the caller's partial sum.
template <int PF>
__device__ __forceinline__ double protein_pair_sum(const double* __restrict__ ATOM, const double2* __restrict__ rec, const int32_t* __restrict__ pij, int n_pairs,
                                                   int first, int stride)
{
    double acc = 0.;
    for (int t0 = first; t0 < n_pairs; t0 += PF * stride) {
        double se[PF], q[PF], rr[PF], s[PF], g[PF], h[PF], e[PF], inv[PF], tv[PF];
        bool ok[PF];
        // ---- squared distance: one 4-byte index word per pair (lane t reads word t), two 32-byte atom records
        int tcl[PF];
        bool close = false;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = t0 + u * stride;
            const int tc = t < n_pairs ? t : n_pairs - 1;      // clamped: the loads are unconditional
            const int w = pij[tc];
            tcl[u] = tc;
            ok[u] = t < n_pairs && w >= 0;
            const int i = w < 0 ? 0 : (w & 0xffff), j = w < 0 ? 1 : (w >> 16);
            const double2 ai0 = *(const double2*)(ATOM + 4 * i), ai1 = *(const double2*)(ATOM + 4 * i + 2);      // (x, y), (z, |.|^2) of atom i
            const double2 aj0 = *(const double2*)(ATOM + 4 * j), aj1 = *(const double2*)(ATOM + 4 * j + 2);
            const double p3 = ai0.x * aj0.x + ai0.y * aj0.y + ai1.x * aj1.x;
            s[u] = ai1.y - 2 * p3 + aj1.y + 0.01;              // >= 0.01 up to rounding: always a positive normal number
            close = close || (ok[u] && s[u] < 81.01);
        }
        // The pairs come in ascending order of their distance in coor_init (mbx_suite_create) and a move displaces a pair by a fraction of an
        // Angstrom, so the tail of the list -- ~60 % of the pairs -- is beyond the 9 A cut-off in every candidate.  sqrt(81.01) = 9.0006: at
        // s >= 81.01 both distance windows are closed and the term is exactly 0, so a wave that sees no closer pair adds nothing and moves on.
        if (!__any(close)) continue;
#pragma unroll
        for (int u = 0; u < PF; ++u) {                         // one 32-byte record per pair (lane t reads record t)
            const double2 a = rec[2 * tcl[u]], b = rec[2 * tcl[u] + 1];
            se[u] = a.x; q[u] = a.y; rr[u] = b.x;
        }
        // ---- pd = sqrt(s) and 1 / pd from one v_rsq_f64 estimate: two coupled Goldschmidt steps, one residual correction each
#pragma unroll
        for (int u = 0; u < PF; ++u) { const double y = __builtin_amdgcn_rsq(s[u]); g[u] = s[u] * y; h[u] = 0.5 * y; }
#pragma unroll
        for (int u = 0; u < PF; ++u) e[u] = __builtin_fma(-h[u], g[u], 0.5);
#pragma unroll
        for (int u = 0; u < PF; ++u) { g[u] = __builtin_fma(g[u], e[u], g[u]); h[u] = __builtin_fma(h[u], e[u], h[u]); }
#pragma unroll
        for (int u = 0; u < PF; ++u) e[u] = __builtin_fma(-h[u], g[u], 0.5);
#pragma unroll
        for (int u = 0; u < PF; ++u) { g[u] = __builtin_fma(g[u], e[u], g[u]); h[u] = __builtin_fma(h[u], e[u], h[u]); }
#pragma unroll
        for (int u = 0; u < PF; ++u) g[u] = __builtin_fma(__builtin_fma(-g[u], g[u], s[u]), h[u], g[u]);    // pd: sqrt(s) + (s - g^2) / (2 sqrt(s))
#pragma unroll
        for (int u = 0; u < PF; ++u) inv[u] = h[u] + h[u];
#pragma unroll
        for (int u = 0; u < PF; ++u) inv[u] = __builtin_fma(inv[u], __builtin_fma(-g[u], inv[u], 1.0), inv[u]);   // 1 / pd
        // ---- Coulomb + Lennard-Jones with the distance windows (protein_docking.py:40-46)
#pragma unroll
        for (int u = 0; u < PF; ++u) rr[u] = rr[u] * inv[u];
#pragma unroll
        for (int u = 0; u < PF; ++u) { const double r2 = rr[u] * rr[u]; rr[u] = r2 * r2 * r2; }                 // (r / pd)^6
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const double pd = g[u], r6 = rr[u];
            const double coeff = q[u] * (0.25 * inv[u]) + se[u] * (r6 * r6 - r6);
            const bool near = pd > 0.11 && pd < 7.0, far = pd > 7.0 && pd < 9.0;
            const double sw = (9 - pd) * (9 - pd) * (-12 + 2 * pd) * 0.125;
            const double c10 = 10 * coeff;
            tv[u] = near ? c10 : (far ? c10 * sw : 0.);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) if (ok[u]) acc += tv[u];
    }
    return acc;
}

// pairs in flight per lane in the wave-per-row mode of eval_rows_protein (the generation kernels run under a 96-register cap)
#ifndef MBX_PROTEIN_PFW
#define MBX_PROTEIN_PFW 2
#endif

template <int DC = 0, class PT = DevProblem, int PF = 1>
__device__ void eval_rows_protein(const PT& P, const EvalLds& L, int n_rows)
{
    const int D = DC ? DC : P.dim, n = P.n_peaks, m3 = 3 * n, tid = threadIdx.x;
    // one 32-byte record per atom: x | y | z | x^2 + y^2 + z^2.  A pair gathers its two atoms with TWO 16-byte-aligned ds_read_b128 pairs instead of eight scattered
    // ds_read_b64 (coordinates at a 24-byte stride + the squared norms in an array of their own: 43 % of the LDS-active cycles of k_dq_step were bank conflicts,
    // profiles/r05_dq_step_pmc.json).  Same LDS footprint (4 n doubles), same arithmetic.
    const double2* __restrict__ rec = (const double2*)P.pyr;       // [n_pairs][2]: (sqrt(e), q), (r, 0) in pair order (mbx_suite_create)
    const int32_t* __restrict__ pij = (const int32_t*)P.plogw;
    const int n_pairs_all = ((n + 1) >> 1) * (n - 1);
    const int n_close = P.n_close;
    const int NW = MBX_NT >> 6;
    // ---- a POPULATION of candidates (every population-based optimizer on protein docking: RLEPSO, GLEET, the resets): ONE WAVE PER ROW.  Each wave keeps its own atom table
    // and walks the pair list with its 64 lanes; no workgroup barrier between rows (the block form pays four per row -- 400 per generation at NP = 100 -- and runs the rows'
    // ~40-deep float64 chains one row at a time: 2.57 ms per generation of 2240 RLEPSO instances, 2.7x its issue bound).  Needs NW atom tables in Z (the caller says how large
    // Z is: EvalLds::z_doubles); a workgroup always takes the same form for the same (rows, threads), so the two routes of a kernel family stay bit-identical.
    if (n_rows >= NW && L.z_doubles >= NW * 4 * n) {
        const int lane = tid & 63, wave = tid >> 6;
        double* ATOM = L.Z + wave * 4 * n;
        for (int r = wave; r < n_rows; r += NW) {
            const double* x = L.X + r * D;
            bool inside = n_close > 0;
            for (int k = 0; k < D; ++k) inside = inside && fabs(x[k]) <= P.ub;
            const int n_pairs = inside ? n_close : n_pairs_all;
            for (int m = lane; m < m3; m += 64) {
                double s = 0.;
                for (int k = 0; k < D; ++k) s += (x[k] * L.V0[k]) * P.py[(size_t)k * m3 + m];
                const int i = m / 3;
                ATOM[4 * i + (m - 3 * i)] = s + P.pc[m];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int i = lane; i < n; i += 64)
                ATOM[4 * i + 3] = ATOM[4 * i] * ATOM[4 * i] + ATOM[4 * i + 1] * ATOM[4 * i + 1] + ATOM[4 * i + 2] * ATOM[4 * i + 2];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double acc = protein_pair_sum<MBX_PROTEIN_PFW>(ATOM, rec, pij, n_pairs, lane, 64);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);      // fixed butterfly: every lane ends with the same sum
            if (lane == 0) L.F[r] = (2 * acc) / n;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // the next row overwrites ATOM
        }
        __syncthreads();
        return;
    }
    double* ATOM = L.Z;
    double* RED = ATOM + 4 * n;
    for (int r = 0; r < n_rows; ++r) {
        const double* x = L.X + r * D;
        // The pair list is ordered by the SMALLEST distance a pair can reach while the candidate stays inside the box (|x_k| <= ub: every optimizer clamps / repairs
        // to it): the pairs behind the first n_close can never come within the 9 A cut-off, their terms are exactly 0 and adding them changes nothing -- the loop
        // stops there (bit-identical sums, ~half of the 4950 distance tests).  A row outside the box (mbx_eval takes any x) walks the whole list.
        bool inside = n_close > 0;
        for (int k = 0; k < D; ++k) inside = inside && fabs(x[k]) <= P.ub;
        const int n_pairs = inside ? n_close : n_pairs_all;
        for (int m = tid; m < m3; m += MBX_NT) {
            double s = 0.;
            for (int k = 0; k < D; ++k) s += (x[k] * L.V0[k]) * P.py[(size_t)k * m3 + m];
            const int i = m / 3;
            ATOM[4 * i + (m - 3 * i)] = s + P.pc[m];
        }
        __syncthreads();
        for (int i = tid; i < n; i += MBX_NT)
            ATOM[4 * i + 3] = ATOM[4 * i] * ATOM[4 * i] + ATOM[4 * i + 1] * ATOM[4 * i + 1] + ATOM[4 * i + 2] * ATOM[4 * i + 2];
        __syncthreads();
        const double acc = protein_pair_sum<PF>(ATOM, rec, pij, n_pairs, tid, MBX_NT);
        const double total = block_sum(acc, RED);
        if (tid == 0) L.F[r] = (2 * total) / n;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// Block-cooperative objective: F[i] = func(X[i,:]) for i < n, bias and boundary penalty included
// (the value F*.func returns).  All arrays of `L` live in LDS; stage_problem() must have been called.
// Must be called by every thread of the block.
// ------------------------------------------------------------------------------------------------
// What __get_costs adds on top of func() (noise, then `- optimum`), applied by the row's own thread at the end of the row phase when the
// caller passes it (population_costs): one barrier interval less than a pass of its own.
struct RowPost {
    const Rng* rng; const double* tape_noise; uint32_t siteA, siteB; int n_total;
};

// NOISE: what the caller knows at compile time -- -1 nothing (the problem record decides), 0 a noise-free function (the noise models are compiled out), 1 a noisy one
template <class PT, int NOISE = -1>
__device__ __forceinline__ double row_post(const PT& P, const RowPost& rp, int i, double f)
{
    if (NOISE != 0 && (NOISE == 1 || P.noise_kind != MBX_NOISE_NONE)) {
        double a, b, c;
        if (rp.tape_noise) { a = rp.tape_noise[i]; b = rp.tape_noise[rp.n_total + i]; c = rp.tape_noise[2 * rp.n_total + i]; }
        else philox_noise(*rp.rng, (uint32_t)i, rp.siteA, rp.siteB, P.noise_kind, a, b, c);
        f = apply_noise(P, f, a, b, c);
    }
    return isnan(P.optimum) ? f : f - P.optimum;
}

// KIND: the caller knows the function kind at compile time (k_rlepso_run's per-kind bodies): every other kind's code is compiled out.  NOISE: see row_post.
template <int DC = 0, int MD = 0, class PT = DevProblem, int KC = 0, int KIND = 0, int NOISE = -1>
__device__ void eval_rows(const PT& P, const EvalLds& L, int n, const RowPost* post = nullptr)
{
    if constexpr (KIND == 0) {
        if (P.kind == MBX_KIND_PROTEIN) {
            // (four pairs in flight only where the caller has the registers: k_dq_step's DevProblem route; the RLEPSO / LDE generation kernels hand a
            //  ConstProblem and run under a 96-register cap next to their resident state -- the plain loop, same sums)
            eval_rows_protein<DC, PT, (std::is_same<PT, DevProblem>::value ? protein_prefetch(DC) : 1)>(P, L, n);
            if (post) { for (int i = threadIdx.x; i < n; i += MBX_NT) L.F[i] = row_post(P, *post, i, L.F[i]); __syncthreads(); }
            return;
        }
    }
    constexpr bool CMP = DC == MBX_COMPOSITE_DC;                   // composite out-of-line transforms (osc1_composite): the headline geometry only
    const int D = DC ? DC : P.dim, NE = n * D, kind = KIND ? KIND : P.kind, tid = opaque_tid();
    const double ub = P.ub, bias = P.bias;
    const double* X = L.X;
    double* Z = L.Z;
    double* T = L.T;
    double* F = L.F;
    const double* M1T = L.M1T;
    const double* M2T = L.M2T;
    const double* v0 = L.V0;
    const double* v1 = L.V1;
    const double* v2 = L.V2;
    const double* dsh = L.DSH;
    const FastDiv fd(D);

    // ---- phase A: first linear map  z = M1 (x - dshift)   (Gallagher: M1 x, the peaks are pre-rotated)
    // Barriers only where a phase exists for this function (the kind is workgroup-uniform, so every thread takes the same path).
    const bool first_map = !(kind == 5 || kind == 20 || kind == 24);
    if (first_map) {
        if constexpr (mfma_matvec(MD, KC)) {
            if (kind == 21 || kind == 22) matvec_rows_mfma<MD, false>(P.m1, X, nullptr, n, Z);
            else matvec_rows_mfma<MD, true>(P.m1, X, dsh, n, Z);
        } else if constexpr (MD > 0 && KC > 0) {
            if (kind == 21 || kind == 22) matvec_rows_scalar_kc<MD, false, KC, (MD == 40 ? MBX_UMAX40 : 4)>(P.m1, X, nullptr, n, Z);
            else matvec_rows_scalar_kc<MD, true, KC, (MD == 40 ? MBX_UMAX40 : 4)>(P.m1, X, dsh, n, Z);
        } else if constexpr (MD > 0) {
            if (kind == 21 || kind == 22) matvec_rows_scalar<MD, false>(P.m1, X, nullptr, n, Z);
            else matvec_rows_scalar<MD, true>(P.m1, X, dsh, n, Z);
        } else {
            if (kind == 21 || kind == 22) matvec_rows(M1T, X, n, D, Z);
            else matvec_rows_shifted(M1T, X, dsh, n, D, Z);
        }
        __syncthreads();
    }

    // ---- phase E1: element-wise transforms
    if (kind == 21 || kind == 22) {
        // Gallagher (bbob.py:796-800): max_k w_k exp(-1/(2D) sum_d C_kd z_kd^2), z_k = R (x - y_k) = R x - R y_k.
        // exp is monotone, so the winning peak of a row is found on key_k = log(w_k) - s_k/(2D) (log w_k pre-computed at
        // upload) and the reference's expression is evaluated for that peak only: one exp per row instead of n_peaks.
        // Wave w takes the peaks k = w (mod number of waves) -- uniform per wave, so the pre-rotated peaks, C and log w come through scalar
        // loads -- and lane l the rows l, l + 64, ..; the running maximum of a (row, wave) pair lives in registers.  Peaks that tie to within rounding are both
        // "the maximum" to 1 ulp; the lower peak index is kept.  The search key accumulates with one fused multiply-add per
        // coordinate (3 instead of 4 instructions in the (row, peak, coordinate) loop); the value of the winning peak is
        // evaluated in the row phase with the reference's expression.
        const int npk = P.n_peaks, lane = tid & 63, wave = tid >> 6;
        const double cexp = -0.5 / D;
        double bkey[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bk[4] = {0, 0, 0, 0};
        if constexpr (DC > 0 && DC <= 16) {
            // compile-time dimension: a row of z stays in registers and the peak tables never touch LDS -- the peak index is
            // wave-uniform, so y_k, C_k and log w_k are read from the constant address space, i.e. as scalar loads into SGPRs
            // that the vector ALU takes as operands (the LDS route above spends 10 broadcast ds_read_b128 per (row, peak) and is
            // LDS-bandwidth bound).  Same visiting order, same arithmetic, same tie rule.
            typedef const double __attribute__((address_space(4)))* kptr;
            const kptr py = (kptr)P.pyr, pcc = (kptr)P.pc, plw = (kptr)P.plogw;
            const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = lane + 64 * q;
                if (i < n) {
                    double rx[DC > 0 ? DC : 1];
#pragma unroll
                    for (int d = 0; d < DC; ++d) rx[d] = Z[i * DC + d];
                    for (int kk = wv; kk < npk; kk += MBX_NW) {
                        double acc = 0.;
#pragma unroll
                        for (int d = 0; d < DC; ++d) { const double zd = rx[d] - py[kk * DC + d]; acc = __builtin_fma(pcc[kk * DC + d], zd * zd, acc); }
                        const double key = plw[kk] + cexp * acc;
                        if (key > bkey[q]) { bkey[q] = key; bk[q] = kk; }
                    }
                }
            }
        } else if constexpr (gallagher_lean(MD, KC)) {
            // D = 30 (LDE, RLEPSO --dim 30) and config 5's resident kernel (D = 40): the same idea as the blocked route below with a footprint that fits these
            // kernels' register caps -- four peaks as accumulators, the row in chunks of six (five at D = 40) coordinates, one row group at a time: one LDS read of the row per FOUR peaks instead of per peak.
            // LDE pop 50: the three Gallagher functions 919 -> 825 us per generation of 16 384 instances, all 30 noisy functions 539 -> 529 us; config 3
            // 0.566 -> 0.556 ms (pop 50), 1.141 -> 1.130 ms (pop 100); config 5's resident kernel on a batch of F21 / F128 only 3.86 -> 2.64 ms per generation
            // (A/B on one box).
            typedef const double __attribute__((address_space(4)))* kptr;
            const kptr py = (kptr)P.pyr, pcc = (kptr)P.pc, plw = (kptr)P.plogw;
            const int wv = __builtin_amdgcn_readfirstlane(wave), nw = __builtin_amdgcn_readfirstlane(MBX_NW);
            constexpr int PB = 4, GC = MD == 40 ? 5 : 6;
            static_assert(MD % GC == 0, "whole chunks only");
            const int mine = wv < npk ? (npk - wv + nw - 1) / nw : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (64 * q >= n) break;                                      // wave-uniform
                const int i = lane + 64 * q;
                const double* rx = Z + (i < n ? i : n - 1) * MD;
                for (int j0 = 0; j0 < mine; j0 += PB) {
                    double acc[PB];
#pragma unroll
                    for (int j = 0; j < PB; ++j) acc[j] = 0.;
#pragma unroll 1
                    for (int c0 = 0; c0 < MD; c0 += GC) {
                        double y[GC];
#pragma unroll
                        for (int k = 0; k < GC; ++k) y[k] = rx[c0 + k];
#pragma unroll
                        for (int j = 0; j < PB; ++j) {
                            if (j0 + j < mine) {                             // wave-uniform
                                const int kk = wv + (j0 + j) * nw;
                                const kptr ry = py + (int64_t)kk * MD + c0;
                                const kptr ck = pcc + (int64_t)kk * MD + c0;
#pragma unroll
                                for (int k = 0; k < GC; ++k) { const double zd = y[k] - ry[k]; acc[j] = __builtin_fma(ck[k], zd * zd, acc[j]); }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        if (j0 + j < mine) {
                            const int kk = wv + (j0 + j) * nw;
                            const double key = plw[kk] + cexp * acc[j];
                            if (key > bkey[q]) { bkey[q] = key; bk[q] = kk; }
                        }
                    }
                }
            }
        } else if constexpr (gallagher_blocked(MD, KC)) {
            // config 5 (D = 40, NP = 128, one 1024-thread workgroup per CU).  The row does not fit in registers, and re-reading it
            // from LDS for every peak made this loop LDS-bound: one 512-byte ds_read per (row group, peak, coordinate) next to three VALU
            // instructions -- the 101-peak functions took 203 k cycles per evaluation at NP = 128, D = 40 on 16 waves against an issue bound of
            // 24 k (instrumented build, round 3), with the peak tables streamed through LDS chunks on top.  Here a wave keeps a block of up to
            // PB of its peaks as accumulators and walks the row in chunks of GC coordinates held in registers: one LDS read of the row per
            // PB peaks, y_k / C_k / log w_k -- wave-uniform -- through scalar loads as in the small-dimension route.  Each key is still the
            // fma chain over d ascending, each (row, wave) still visits its peaks k = wave, wave + waves, .. in ascending order with the
            // strict `>`: bit-identical to the routes this replaces.
            typedef const double __attribute__((address_space(4)))* kptr;
            const kptr py = (kptr)P.pyr, pcc = (kptr)P.pc, plw = (kptr)P.plogw;
            const int wv = __builtin_amdgcn_readfirstlane(wave), nw = __builtin_amdgcn_readfirstlane(MBX_NW);
            constexpr int PB = 8, GC = 10;
            const int mine = wv < npk ? (npk - wv + nw - 1) / nw : 0;       // peaks of this wave: wv + j nw, j < mine
            // two row groups (lanes l and l + 64) share every scalar load of y_k / C_k: half the scalar traffic and two independent chains
#pragma unroll
            for (int q0 = 0; q0 < 4; q0 += 2) {
                if (64 * q0 >= n) break;                                     // wave-uniform
                const int i0 = lane + 64 * q0, i1 = i0 + 64;
                const double* rx0 = Z + (i0 < n ? i0 : n - 1) * D;          // lanes past the last row repeat it (results unused)
                const double* rx1 = Z + (i1 < n ? i1 : n - 1) * D;
                for (int j0 = 0; j0 < mine; j0 += PB) {
                    double acc0[PB], acc1[PB];
#pragma unroll
                    for (int j = 0; j < PB; ++j) { acc0[j] = 0.; acc1[j] = 0.; }
                    for (int c0 = 0; c0 < D; c0 += GC) {
                        double y0[GC], y1[GC];
                        const int kn = D - c0 < GC ? D - c0 : GC;
                        if (kn == GC) {                                      // a whole chunk (every chunk at D = 10, 20, 30, 40)
#pragma unroll
                            for (int k = 0; k < GC; ++k) { y0[k] = rx0[c0 + k]; y1[k] = rx1[c0 + k]; }
#pragma unroll
                            for (int j = 0; j < PB; ++j) {
                                if (j0 + j < mine) {                         // wave-uniform
                                    const int kk = wv + (j0 + j) * nw;
                                    const kptr ry = py + (int64_t)kk * D + c0;
                                    const kptr ck = pcc + (int64_t)kk * D + c0;
#pragma unroll
                                    for (int k = 0; k < GC; ++k) {
                                        const double z0 = y0[k] - ry[k], z1 = y1[k] - ry[k];
                                        acc0[j] = __builtin_fma(ck[k], z0 * z0, acc0[j]);
                                        acc1[j] = __builtin_fma(ck[k], z1 * z1, acc1[j]);
                                    }
                                }
                            }
                        } else {
                            for (int k = 0; k < kn; ++k) { y0[k] = rx0[c0 + k]; y1[k] = rx1[c0 + k]; }
#pragma unroll
                            for (int j = 0; j < PB; ++j) {
                                if (j0 + j < mine) {
                                    const int kk = wv + (j0 + j) * nw;
                                    const kptr ry = py + (int64_t)kk * D + c0;
                                    const kptr ck = pcc + (int64_t)kk * D + c0;
                                    for (int k = 0; k < kn; ++k) {
                                        const double z0 = y0[k] - ry[k], z1 = y1[k] - ry[k];
                                        acc0[j] = __builtin_fma(ck[k], z0 * z0, acc0[j]);
                                        acc1[j] = __builtin_fma(ck[k], z1 * z1, acc1[j]);
                                    }
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        if (j0 + j < mine) {
                            const int kk = wv + (j0 + j) * nw;
                            const double k0 = plw[kk] + cexp * acc0[j], k1 = plw[kk] + cexp * acc1[j];
                            if (k0 > bkey[q0]) { bkey[q0] = k0; bk[q0] = kk; }
                            if (k1 > bkey[q0 + 1]) { bkey[q0 + 1] = k1; bk[q0 + 1] = kk; }
                        }
                    }
                }
            }
        } else if (MBX_NW <= 8) {
            // run-time (or large) dimension, up to 8 waves: the row stays in LDS (one read per coordinate, lanes = rows), the peak tables
            // still come through scalar loads instead of LDS chunks (three LDS reads per (row, peak, coordinate)): LDE's Gallagher functions
            // at D = 30 1290 -> 990 us per 16 384 instances.  (With 16 waves, config 5, a wave visits only 6-7 peaks and the chunk
            // route below is 3 % faster: it stays for 1024-thread workgroups.)
            typedef const double __attribute__((address_space(4)))* kptr;
            const kptr py = (kptr)P.pyr, pcc = (kptr)P.pc, plw = (kptr)P.plogw;
            const int wv = __builtin_amdgcn_readfirstlane(wave), nw = __builtin_amdgcn_readfirstlane(MBX_NW);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = lane + 64 * q;
                if (i < n) {
                    const double* rx = Z + i * D;
                    for (int kk = wv; kk < npk; kk += nw) {
                        const kptr ry = py + (int64_t)kk * D;
                        const kptr ck = pcc + (int64_t)kk * D;
                        double acc = 0.;
#pragma unroll 8
                        for (int d = 0; d < D; ++d) { const double zd = rx[d] - ry[d]; acc = __builtin_fma(ck[d], zd * zd, acc); }
                        const double key = plw[kk] + cexp * acc;
                        if (key > bkey[q]) { bkey[q] = key; bk[q] = kk; }
                    }
                }
            }
        } else {
            const int CH = (int)(eval_t_doubles(n, D) / (2 * D + 1));       // peaks per chunk
            double* TY = T; double* TC = T + CH * D; double* TW = T + 2 * CH * D;
            for (int c0 = 0; c0 < npk; c0 += CH) {
                const int cn = npk - c0 < CH ? npk - c0 : CH;
                for (int t = tid; t < cn * D; t += MBX_NT) { TY[t] = P.pyr[c0 * D + t]; TC[t] = P.pc[c0 * D + t]; }
                for (int t = tid; t < cn; t += MBX_NT) TW[t] = P.plogw[c0 + t];
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int i = lane + 64 * q;
                    if (i < n) {
                        const double* rx = Z + i * D;
                        for (int kk = wave; kk < cn; kk += MBX_NW) {
                            const double* ry = TY + kk * D;
                            const double* ck = TC + kk * D;
                            double acc = 0.;
                            for (int d = 0; d < D; ++d) { const double zd = rx[d] - ry[d]; acc = __builtin_fma(ck[d], zd * zd, acc); }
                            const double key = TW[kk] + cexp * acc;
                            if (key > bkey[q]) { bkey[q] = key; bk[q] = c0 + kk; }
                        }
                    }
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = lane + 64 * q;
            if (i < n) { T[(wave * n + i) * 2] = bkey[q]; T[(wave * n + i) * 2 + 1] = (double)bk[q]; }
        }
    } else {
        const double kats_exp = kind == 23 ? 10 / m_pow((double)D, 1.2) : 0.;
        for (int e = tid; e < NE; e += MBX_NT) {
            const int d = fd.mod(e);
            switch (kind) {
            case 2: case 10: { const double o = osc_t<CMP>(Z[e]); T[e] = v0[d] * (o * o); break; }
            case 3: { const double z = v0[d] * asy1(osc_t<CMP>(Z[e]), v1[d]); Z[e] = z; T[e] = m_cos(kTwoPi * z); break; }
            case 4: {
                double o = osc_t<CMP>(Z[e]);
                if ((d & 1) == 0 && o > 0.) o *= 10.;
                const double z = o * v0[d];
                Z[e] = z; T[e] = m_cos(kTwoPi * z);
                break;
            }
            case 5: {
                const double x = X[e];
                double zi = x;
                if (x * dsh[d] > ub * ub) zi = (zi > 0. ? 1. : (zi < 0. ? -1. : 0.)) * ub;
                T[e] = v1[d] - zi * v0[d];
                break;
            }
            case 6: { double zi = Z[e]; if (zi * dsh[d] > 0.) zi *= 100.; T[e] = zi * zi; break; }
            case 7: {
                const double zh = Z[e];
                T[e] = fabs(zh) > 0.5 ? floor(0.5 + zh) : floor(0.5 + 10. * zh) / 10.;
                break;
            }
            case 8: Z[e] = P.s[0] * Z[e] + 1; break;
            case 9: case 19: Z[e] = Z[e] + 0.5; break;
            case 11: { const double o = osc_t<CMP>(Z[e]); T[e] = o * o; break; }
            case 12: case 17: case 18: T[e] = asy1(Z[e], v1[d]); break;
            case 14: T[e] = m_pow(fabs(Z[e]), v0[d]); break;
            case 15: T[e] = asy1(osc_t<CMP>(Z[e]), v1[d]); break;
            case 16: T[e] = osc_t<CMP>(Z[e]); break;
            case 20: T[e] = v2[d] * X[e]; break;
            case 23: {                                              // Katsuura inner series, bbob.py:858-863
                const double z = Z[e];
                double temp = 0., p2 = 1., ip2 = 1.;
                for (int j = 1; j <= 32; ++j) {
                    p2 *= 2.; ip2 *= 0.5;                          // x / 2^j == x * 2^-j bit for bit
                    const double a = p2 * z;
                    temp += fabs(a - floor(a + 0.5)) * ip2;
                }
                T[e] = m_pow(1 + (d + 1) * temp, kats_exp);
                break;
            }
            case 24: T[e] = v0[d] * X[e] - P.s[0]; break;         // x_hat - mu0
            default: break;                                         // 1, 13: nothing element-wise
            }
        }
    }
    if (!(kind == 1 || kind == 13)) __syncthreads();

    // ---- phase C: second linear map (F7 keeps |z_hat_0| in F first)
    if (kind == 7) {
        for (int i = tid; i < n; i += MBX_NT) F[i] = fabs(Z[i * D]);
        __syncthreads();
        if constexpr (mfma_matvec(MD, KC)) matvec_rows_mfma<MD, false>(P.m2, T, nullptr, n, Z); else if constexpr (MD > 0 && KC > 0) matvec_rows_scalar_kc<MD, false, KC, (MD == 40 ? MBX_UMAX40 : 4)>(P.m2, T, nullptr, n, Z); else if constexpr (MD > 0) matvec_rows_scalar<MD, false>(P.m2, T, nullptr, n, Z); else matvec_rows(M2T, T, n, D, Z);
        __syncthreads();
    } else if (kind == 12 || kind == 24) {
        if constexpr (mfma_matvec(MD, KC)) matvec_rows_mfma<MD, false>(P.m1, T, nullptr, n, Z); else if constexpr (MD > 0 && KC > 0) matvec_rows_scalar_kc<MD, false, KC, (MD == 40 ? MBX_UMAX40 : 4)>(P.m1, T, nullptr, n, Z); else if constexpr (MD > 0) matvec_rows_scalar<MD, false>(P.m1, T, nullptr, n, Z); else matvec_rows(M1T, T, n, D, Z);
        __syncthreads();
    } else if (kind == 15 || kind == 16 || kind == 17 || kind == 18) {
        if constexpr (mfma_matvec(MD, KC)) matvec_rows_mfma<MD, false>(P.m2, T, nullptr, n, Z); else if constexpr (MD > 0 && KC > 0) matvec_rows_scalar_kc<MD, false, KC, (MD == 40 ? MBX_UMAX40 : 4)>(P.m2, T, nullptr, n, Z); else if constexpr (MD > 0) matvec_rows_scalar<MD, false>(P.m2, T, nullptr, n, Z); else matvec_rows(M2T, T, n, D, Z);
        __syncthreads();
    }

    // ---- phase E2: element-wise terms after the second map
    if (kind == 15 || kind == 24) {
        for (int e = tid; e < NE; e += MBX_NT) T[e] = m_cos(kTwoPi * Z[e]);
    } else if (kind == 16) {                                        // Weierstrass series, bbob.py:623
        // sum_k 0.5^k cos(3^k b), b = 2 pi (z + 0.5): the twelve cosines are one (cos b, sin b) pair and eleven complex cubes
        //   (c + i s)^3:  c' = c (c^2 - 3 s^2),  s' = s (3 c^2 - s^2)
        // -- 7 instructions per term instead of a range reduction + 21st-order polynomial each (~35).  The cube triples the angle error and the
        // radius error of the pair per step (the one-square forms cos 3a = cos a (1 - 4 sin^2 a) amplify a radius error 9x per step and end at
        // 2e-10): 3^11 x 1e-16 = 2e-11 at the last term, whose weight is 0.5^11 -- 3e-14 absolute in the sum against an extended-precision
        // evaluation.  The reference rounds the ARGUMENT 3^k b of each cosine (up to 5.5e6 rad at k = 11, half an ulp = 5e-10 rad): its own sum is
        // 5.6e-13 off the same exact value, which is the distance between the two routes; the KATs pin the function at 1e-10 relative.
        for (int e = tid; e < NE; e += MBX_NT) {
            const double base = kTwoPi * (Z[e] + 0.5);
            double c = m_cos(base), s = m_sin(base), sum = c, ak = 1.;
#pragma unroll
            for (int k = 1; k < 12; ++k) {
                const double c2 = c * c, s2 = s * s;
                c = c * __builtin_fma(-3., s2, c2);
                s = s * __builtin_fma(3., c2, -s2);
                ak *= 0.5;                                          // compile-time constant after unrolling
                sum = __builtin_fma(ak, c, sum);                    // 0.5^k c is exact: the same value as sum + ak * c
            }
            T[e] = sum;
        }
    } else if (kind == 17 || kind == 18) {                          // Schaffers, bbob.py:642-643
        for (int e = tid; e < NE; e += MBX_NT) {
            const int d = fd.mod(e);
            if (d < D - 1) {
                const double s = sqrt(Z[e] * Z[e] + Z[e + 1] * Z[e + 1]);
                if constexpr (CMP) T[e] = schaffers_composite(s);
                else T[e] = sqrt(s) * (m_pow(m_sin(50 * m_pow(s, 0.2)), 2) + 1);
            }
        }
    } else if (kind == 19) {                                        // Griewank-Rosenbrock, bbob.py:702-703
        for (int e = tid; e < NE; e += MBX_NT) {
            const int d = fd.mod(e);
            if (d < D - 1) {
                const double a = Z[e] * Z[e] - Z[e + 1];
                const double b = 1. - Z[e];
                const double s = 100. * (a * a) + b * b;
                T[e] = s / 4000. - m_cos(s);
            }
        }
    } else if (kind == 20) {                                        // Schwefel, bbob.py:754-756
        for (int e = tid; e < NE; e += MBX_NT) {
            const int d = fd.mod(e);
            double zi = T[e];
            if (d > 0) zi += 0.25 * (T[e - 1] - v1[d - 1]);
            Z[e] = 100. * (v0[d] * (zi - v1[d]) + v1[d]);
        }
        __syncthreads();
        for (int e = tid; e < NE; e += MBX_NT) {
            const double z = Z[e];
            const double q = fmax(0., fabs(z / 100) - ub);
            T[e] = q * q;
            Z[e] = z * m_sin(sqrt(fabs(z)));
        }
    }
    if (kind == 15 || kind == 24 || (kind >= 16 && kind <= 20)) __syncthreads();

    // ---- row phase: sequential sums over d in ascending order (one thread per row)
    for (int i = tid; i < n; i += MBX_NT) {
        const double* x = X + i * D;
        const double* z = Z + i * D;
        const double* t = T + i * D;
        const double bh = P.pen_coef != 0. ? P.pen_coef * pen_row(x, D, ub) : 0.;
        double f;
        switch (kind) {
        case 1: { double s = 0.; for (int d = 0; d < D; ++d) s += z[d] * z[d]; f = s + bias + bh; break; }
        case 2: case 10: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = s + bias + bh; break; }
        case 3: case 15: {
            double sc = 0., sq = 0.;
            for (int d = 0; d < D; ++d) { sc += t[d]; sq += z[d] * z[d]; }
            f = 10. * (D - sc) + sq + bias;
            break;
        }
        case 4: {
            double sc = 0., sq = 0.;
            for (int d = 0; d < D; ++d) { sc += t[d]; sq += z[d] * z[d]; }
            f = 10. * (D - sc) + sq + 100 * pen_row(x, D, ub) + bias;
            break;
        }
        case 5: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = s + bias; break; }
        case 6: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = m_pow(osc_t<CMP>(s), 0.9) + bias; break; }
        case 7: {
            double s = 0.;
            for (int d = 0; d < D; ++d) s += v0[d] * (z[d] * z[d]);
            f = 0.1 * fmax(F[i] / 1e4, s) + bh + bias;
            break;
        }
        case 8: case 9: {
            double s = 0.;
            for (int d = 0; d < D - 1; ++d) {
                const double a = z[d] * z[d] - z[d + 1];
                const double b = z[d] - 1;
                s += 100 * (a * a) + b * b;
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
            f = 10 * m_pow(acc / D - P.s[0], 3) + 10. / D * pen_row(x, D, ub) + bias;
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
        case 20: {
            double acc = 0., pen = 0.;
            for (int d = 0; d < D; ++d) { acc += z[d]; pen += t[d]; }
            f = 4.189828872724339 - 0.01 * (acc / D) + 100 * pen + bias;
            break;
        }
        case 21: case 22: {
            double key = T[i * 2]; int ks = (int)T[i * 2 + 1];                 // combine the waves' partial maxima
            for (int w = 1; w < MBX_NW; ++w) {
                const double kw = T[(w * n + i) * 2]; const int kq = (int)T[(w * n + i) * 2 + 1];
                if (kw > key || (kw == key && kq < ks)) { key = kw; ks = kq; }
            }
            const double* __restrict__ ry = P.pyr + ks * D;
            const double* __restrict__ ck = P.pc + ks * D;
            double acc = 0.;
            for (int d = 0; d < D; ++d) { const double zd = z[d] - ry[d]; acc += ck[d] * (zd * zd); }
            const double best = P.pw[ks] * m_exp((-0.5 / D) * acc);
            const double o = osc_t<CMP>(10 - best);
            f = o * o + bias + bh;
            break;
        }
        case 23: {
            double res = 1.; for (int d = 0; d < D; ++d) res *= t[d];
            const double tmp = 10. / D / D;
            f = res * tmp - tmp + pen_row(x, D, ub) + bias;
            break;
        }
        case 24: {
            const double mu0 = P.s[0], sc_ = P.s[1], mu1 = P.s[2];
            double a = 0., b = 0., sc = 0.;
            for (int d = 0; d < D; ++d) {
                const double xh = v0[d] * x[d];
                a += (xh - mu0) * (xh - mu0);
                b += (xh - mu1) * (xh - mu1);
                sc += t[d];
            }
            f = fmin(a, D + sc_ * b) + 10. * (D - sc) + 1e4 * pen_row(x, D, ub) + bias;
            break;
        }
        default: f = NAN; break;
        }
        F[i] = post ? row_post<PT, NOISE>(P, *post, i, f) : f;
    }
    __syncthreads();
}


// cost_i = problem.eval(x_i) [- optimum] for the n rows staged in L.X (the __get_costs of every optimizer, e.g.
// rlepso_optimizer.py:68-74): objective, then NoisyProblem's noise with draws from the replay tape ([3, n] rows) or from Philox
// (sites siteA / siteB, row index = draw index), then the optimum.  Results in L.F; ends with a barrier.  All threads call.
template <int DC = 0, int MD = 0, class PT = DevProblem, int KC = 0, int KIND = 0, int NOISE = -1>
__device__ __forceinline__ void population_costs(const PT& P, const EvalLds& L, int n, const Rng& rng, const double* tape_noise,
                                                 uint32_t siteA, uint32_t siteB)
{
    const RowPost post{&rng, tape_noise, siteA, siteB, n};
#ifdef MBX_ABLATE_EVAL
    for (int i = threadIdx.x; i < n; i += MBX_NT) L.F[i] = row_post(P, post, i, L.X[i * P.dim] * L.X[i * P.dim] + P.bias);
    __syncthreads();
#else
    eval_rows<DC, MD, PT, KC, KIND, NOISE>(P, L, n, &post);
#endif
}

}  // namespace mbx
