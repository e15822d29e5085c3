// mbx_device.hpp — device-side building blocks for gfx950 (wave64): Philox4x32-10, block reductions and
// the block-cooperative BBOB evaluator shared by the stand-alone eval kernel and the fused generation
// kernels.  One workgroup evaluates a whole population that is resident in LDS.
//
// Reference semantics being implemented (cited per function): src/problem/bbob.py of GMC-DRL/MetaBox.
// Arithmetic is float64 and the translation unit is built with -ffp-contract=off so that element-wise
// expressions round like numpy's (no implicit FMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mbx.h"
#include "../../include/mbx_layout.h"

namespace mbx {

constexpr int kThreads = 256;          // 4 waves of 64
constexpr double kTwoPi = 6.283185307179586;

// ------------------------------------------------------------------------------------------------
// Device copy of a problem (pointers are device addresses inside the suite's constant pool).
// ------------------------------------------------------------------------------------------------
struct DevProblem {
    int32_t func_id, kind, dim, n_peaks, noise_kind, pad;
    double bias, lb, ub, pen_coef, s[4], noise_a, noise_b, optimum;
    const double *dshift, *m1, *m2, *v0, *v1, *v2, *py, *pc, *pw;
};

// ------------------------------------------------------------------------------------------------ Philox
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(MBX_PHILOX_M0, c0), lo0 = MBX_PHILOX_M0 * c0;
        const uint32_t hi1 = __umulhi(MBX_PHILOX_M1, c2), lo1 = MBX_PHILOX_M1 * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += MBX_PHILOX_W0; k1 += MBX_PHILOX_W1;
    }
    return U4{c0, c1, c2, c3};
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

struct Rng {                      // per-instance stream: key = seed, counter = (index, site, gen, episode)
    uint32_t k0, k1, gen, episode;
    __device__ __forceinline__ U4 draw(uint32_t idx, uint32_t site) const
    {
        return philox4x32_10(idx, site, gen, episode, k0, k1);
    }
};

__device__ __forceinline__ void box_muller(double ua, double ub, double& n0, double& n1)
{
    const double r = sqrt(-2.0 * log(1.0 - ua));
    const double t = kTwoPi * ub;
    n0 = r * cos(t);
    n1 = r * sin(t);
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
        double n1;
        box_muller(ua, ub, a, n1);
    } else if (noise_kind == MBX_NOISE_UNIFORM) {
        a = ua; b = ub;
    } else {
        a = ua;
        w = rng.draw(row, siteB);
        box_muller(u53(w.x, w.y), u53(w.z, w.w), b, c);
    }
}

// NoisyProblem.noisy for one value (bbob.py:108-146)
__device__ __forceinline__ double apply_noise(const DevProblem& P, double ftrue, double a, double b, double c)
{
    if (P.noise_kind == MBX_NOISE_NONE) return ftrue;
    const double fu = ftrue - P.optimum;
    double fn;
    if (P.noise_kind == MBX_NOISE_GAUSS) {
        fn = fu * exp(P.noise_a * a);
    } else if (P.noise_kind == MBX_NOISE_UNIFORM) {
        fn = fu * pow(a, P.noise_b) * fmax(1., pow(1e9 / (fu + 1e-99), P.noise_a * (0.49 + 1. / P.dim) * b));
    } else {
        fn = fu + P.noise_a * fmax(0., 1e3 + (a < P.noise_b ? 1. : 0.) * b / (fabs(c) + 1e-199));
    }
    return fu >= 1e-8 ? fn + P.optimum + 1.01 * 1e-8 : ftrue;
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
    const int tid = threadIdx.x;
    if (tid < 64) {
        double v = INFINITY; int idx = 0x7fffffff;
        for (int j = tid; j < n; j += 64) {
            const double x = a[j];
            if (x < v) { v = x; idx = j; }
        }
        wave_argmin(v, idx);
        if (tid == 0) { red[0] = v; red[1] = (double)idx; }
    }
    __syncthreads();
    vmin = red[0]; imin = (int)red[1];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ BBOB pieces
__device__ __forceinline__ double osc1(double x)                 // osc_transform, bbob.py:51-67
{
    if (x > 0.) {
        const double y = log(x) / 0.1;
        return pow(exp(y + 0.49 * (sin(y) + sin(0.79 * y))), 0.1);
    }
    if (x < 0.) {
        const double y = log(-x) / 0.1;
        return -pow(exp(y + 0.49 * (sin(0.55 * y) + sin(0.31 * y))), 0.1);
    }
    return x;
}

__device__ __forceinline__ double asy1(double x, double beta_lin)  // asy_transform, bbob.py:70-82
{
    return x > 0. ? pow(x, 1. + beta_lin * sqrt(x)) : x;
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

// Stage a D x D row-major matrix from global memory into LDS, transposed (MT[k*D+d] = M[d*D+k]) so that
// lanes that differ in d read consecutive LDS words.
__device__ __forceinline__ void stage_transposed(const double* __restrict__ M, int D, double* MT)
{
    if (M == nullptr) return;
    for (int t = threadIdx.x; t < D * D; t += kThreads) {
        const int d = t / D, k = t - d * D;
        MT[k * D + d] = M[t];
    }
}

// Out[i][d] = sum_k M[d][k] * (In[i][k] - sub[k])   (sr_func, bbob.py:6-8; sub == nullptr -> plain product)
__device__ __forceinline__ void matvec_rows(const double* MT, const double* In, const double* __restrict__ sub, int n, int D,
                                            double* Out)
{
    const int NE = n * D;
    for (int e = threadIdx.x; e < NE; e += kThreads) {
        const int i = e / D, d = e - i * D;
        const double* row = In + i * D;
        double s = 0.;
        if (sub) {
            for (int k = 0; k < D; ++k) s += MT[k * D + d] * (row[k] - sub[k]);
        } else {
            for (int k = 0; k < D; ++k) s += MT[k * D + d] * row[k];
        }
        Out[e] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Block-cooperative objective: F[i] = func(X[i,:]) for i < n, bias and boundary penalty included
// (the value F*.func returns).  X, Z, T are LDS arrays of n*D doubles, M1T/M2T the transposed linear
// maps in LDS, F an LDS array of n doubles.  Must be called by every thread of the block.
// ------------------------------------------------------------------------------------------------
__device__ void eval_rows(const DevProblem& P, const double* X, int n, double* Z, double* T, const double* M1T,
                          const double* M2T, double* F)
{
    const int D = P.dim, NE = n * D, kind = P.kind, tid = threadIdx.x;
    const double ub = P.ub, bias = P.bias;
    const double* __restrict__ v0 = P.v0;
    const double* __restrict__ v1 = P.v1;
    const double* __restrict__ v2 = P.v2;

    // ---- phase A: first linear map
    const bool first_map = !(kind == 5 || kind == 20 || kind == 21 || kind == 22 || kind == 24);
    if (first_map) matvec_rows(M1T, X, P.dshift, n, D, Z);
    __syncthreads();

    // ---- phase E1: element-wise transforms
    if (kind == 21 || kind == 22) {
        // Gallagher (bbob.py:796-800): S threads share a row and split the peaks; partial maxima go to Z.
        const int S = kThreads / n > 0 ? kThreads / n : 1;
        const int npk = P.n_peaks;
        for (int w = tid; w < n * S; w += kThreads) {
            const int i = w / S, sl = w - i * S;
            const double* x = X + i * D;
            double best = -INFINITY;
            for (int k = sl; k < npk; k += S) {
                const double* __restrict__ yk = P.py + k * D;
                const double* __restrict__ ck = P.pc + k * D;
                double acc = 0.;
                for (int d = 0; d < D; ++d) {
                    double zd = 0.;
                    for (int j = 0; j < D; ++j) zd += M1T[j * D + d] * (x[j] - yk[j]);
                    acc += ck[d] * (zd * zd);
                }
                const double e = P.pw[k] * exp((-0.5 / D) * acc);
                if (e > best) best = e;
            }
            Z[w] = best;
        }
    } else {
        for (int e = tid; e < NE; e += kThreads) {
            const int i = e / D, d = e - i * D;
            switch (kind) {
            case 2: case 10: { const double o = osc1(Z[e]); T[e] = v0[d] * (o * o); break; }
            case 3: { const double z = v0[d] * asy1(osc1(Z[e]), v1[d]); Z[e] = z; T[e] = cos(kTwoPi * z); break; }
            case 4: {
                double o = osc1(Z[e]);
                if ((d & 1) == 0 && o > 0.) o *= 10.;
                const double z = o * v0[d];
                Z[e] = z; T[e] = cos(kTwoPi * z);
                break;
            }
            case 5: {
                const double x = X[e];
                double zi = x;
                if (x * P.dshift[d] > ub * ub) zi = (zi > 0. ? 1. : (zi < 0. ? -1. : 0.)) * ub;
                T[e] = v1[d] - zi * v0[d];
                break;
            }
            case 6: { double zi = Z[e]; if (zi * P.dshift[d] > 0.) zi *= 100.; T[e] = zi * zi; break; }
            case 7: {
                const double zh = Z[e];
                T[e] = fabs(zh) > 0.5 ? floor(0.5 + zh) : floor(0.5 + 10. * zh) / 10.;
                break;
            }
            case 8: Z[e] = P.s[0] * Z[e] + 1; break;
            case 9: case 19: Z[e] = Z[e] + 0.5; break;
            case 11: { const double o = osc1(Z[e]); T[e] = o * o; break; }
            case 12: case 17: case 18: T[e] = asy1(Z[e], v1[d]); break;
            case 14: T[e] = pow(fabs(Z[e]), v0[d]); break;
            case 15: T[e] = asy1(osc1(Z[e]), v1[d]); break;
            case 16: T[e] = osc1(Z[e]); break;
            case 20: T[e] = v2[d] * X[e]; break;
            case 23: {                                              // Katsuura inner series, bbob.py:858-863
                const double z = Z[e];
                double temp = 0., p2 = 1.;
                for (int j = 1; j <= 32; ++j) {
                    p2 *= 2.;
                    const double a = p2 * z;
                    temp += fabs(a - floor(a + 0.5)) / p2;
                }
                T[e] = pow(1 + (d + 1) * temp, 10 / pow((double)D, 1.2));
                break;
            }
            case 24: T[e] = v0[d] * X[e] - P.s[0]; break;         // x_hat - mu0
            default: break;                                         // 1, 13: nothing element-wise
            }
        }
    }
    __syncthreads();

    // ---- phase C: second linear map (F7 keeps |z_hat_0| in F first)
    if (kind == 7) {
        for (int i = tid; i < n; i += kThreads) F[i] = fabs(Z[i * D]);
        __syncthreads();
        matvec_rows(M2T, T, nullptr, n, D, Z);
    } else if (kind == 12 || kind == 24) {
        matvec_rows(M1T, T, nullptr, n, D, Z);
    } else if (kind == 15 || kind == 16 || kind == 17 || kind == 18) {
        matvec_rows(M2T, T, nullptr, n, D, Z);
    }
    __syncthreads();

    // ---- phase E2: element-wise terms after the second map
    if (kind == 15 || kind == 24) {
        for (int e = tid; e < NE; e += kThreads) T[e] = cos(kTwoPi * Z[e]);
    } else if (kind == 16) {                                        // Weierstrass series, bbob.py:623
        for (int e = tid; e < NE; e += kThreads) {
            const double base = kTwoPi * (Z[e] + 0.5);
            double s = 0., ak = 1., bk = 1.;
            for (int k = 0; k < 12; ++k) { s += ak * cos(base * bk); ak *= 0.5; bk *= 3.; }
            T[e] = s;
        }
    } else if (kind == 17 || kind == 18) {                          // Schaffers, bbob.py:642-643
        for (int e = tid; e < NE; e += kThreads) {
            const int i = e / D, d = e - i * D;
            if (d < D - 1) {
                const double s = sqrt(Z[e] * Z[e] + Z[e + 1] * Z[e + 1]);
                T[e] = sqrt(s) * (pow(sin(50 * pow(s, 0.2)), 2) + 1);
            }
        }
    } else if (kind == 19) {                                        // Griewank-Rosenbrock, bbob.py:702-703
        for (int e = tid; e < NE; e += kThreads) {
            const int i = e / D, d = e - i * D;
            if (d < D - 1) {
                const double a = Z[e] * Z[e] - Z[e + 1];
                const double b = 1. - Z[e];
                const double s = 100. * (a * a) + b * b;
                T[e] = s / 4000. - cos(s);
            }
        }
    } else if (kind == 20) {                                        // Schwefel, bbob.py:754-756
        for (int e = tid; e < NE; e += kThreads) {
            const int i = e / D, d = e - i * D;
            double zi = T[e];
            if (d > 0) zi += 0.25 * (T[e - 1] - v1[d - 1]);
            Z[e] = 100. * (v0[d] * (zi - v1[d]) + v1[d]);
        }
        __syncthreads();
        for (int e = tid; e < NE; e += kThreads) {
            const double z = Z[e];
            const double q = fmax(0., fabs(z / 100) - ub);
            T[e] = q * q;
            Z[e] = z * sin(sqrt(fabs(z)));
        }
    }
    __syncthreads();

    // ---- row phase: sequential sums over d in ascending order (one thread per row)
    for (int i = tid; i < n; i += kThreads) {
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
        case 6: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = pow(osc1(s), 0.9) + bias; break; }
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
        case 14: { double s = 0.; for (int d = 0; d < D; ++d) s += t[d]; f = pow(s, 0.5) + bias + bh; break; }
        case 16: {
            double acc = 0.; for (int d = 0; d < D; ++d) acc += t[d];
            f = 10 * pow(acc / D - P.s[0], 3) + 10. / D * pen_row(x, D, ub) + bias;
            break;
        }
        case 17: case 18: {
            double acc = 0.; for (int d = 0; d < D - 1; ++d) acc += t[d];
            f = pow(1. / (D - 1) * acc, 2) + bh + bias;
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
            const int S = kThreads / n > 0 ? kThreads / n : 1;
            double best = -INFINITY;
            for (int sl = 0; sl < S; ++sl) { const double v = Z[i * S + sl]; if (v > best) best = v; }
            const double o = osc1(10 - best);
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
        F[i] = f;
    }
    __syncthreads();
}

}  // namespace mbx
