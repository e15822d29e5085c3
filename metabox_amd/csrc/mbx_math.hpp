// mbx_math.hpp — float64 log / exp / sin / cos / pow for the objective functions, written for the argument ranges they see.
//
// The BBOB transforms are transcendental-bound on this GPU: T_osz is log + 2 sin + exp per coordinate, T_asy a general pow,
// Rastrigin / Weierstrass / Griewank cosines -- a third of k_rlepso_step's instructions and most of the D = 30 kernels.  The ROCm
// device library routines are built for every input (Payne-Hanek reduction, double-double pow, denormals); here
//   * x > 0 normal for log, |x| <= 700 for exp, |x| < 2^26 for sin / cos, x > 0 and |y log x| <= 700 for pow
// are the fast paths (FMA Cody-Waite reductions, series coefficients 1/n! and 2/(2n+1) evaluated by Horner with FMA, reciprocal by
// v_rcp_f64 + two Newton steps), everything else falls through to the library.  Accuracy: <= 2 ulp for log / exp / sin / cos,
// <= (2 + |y log x|) ulp for pow -- measured on the device against numpy by tests/test_gpu_bbob.py::test_device_math_accuracy.
// The reference computes these with the C library (<= 1 ulp); the 1e-10 KAT tolerance and the 1e-5 trajectory contract are seven
// orders of magnitude looser.  -DMBX_LIBM_MATH restores the library calls.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbx {
namespace fm {

__device__ __forceinline__ double from_bits(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t to_bits(double d) { return (uint64_t)__double_as_longlong(d); }

// Horner step p z + c with the 64-bit coefficient c as the SGPR-pair operand of v_fma_f64.  Written with the builtin, clang emits
// v_mov_b32 x 2 (the literal into a VGPR pair) + v_fmac_f64 for every coefficient -- three VALU instructions per step; the coefficient
// belongs on the scalar unit (s_mov_b32 x 2, issued beside the vector pipe), which leaves ONE VALU instruction per step.  Same fused
// multiply-add, same rounding.
__device__ __forceinline__ double fma_k(double p, double z, double c)
{
#ifdef MBX_PLAIN_HORNER
    return __builtin_fma(p, z, c);
#else
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(z), "s"(c));
    return r;
#endif
}

// 1 / a for normal a: hardware estimate + two Newton steps (quadratic: 2^-13 -> 2^-26 -> 2^-52)
__device__ __forceinline__ double recip(double a)
{
    double r = __builtin_amdgcn_rcp(a);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-a, r, 1.0), r);
    return r;
}

// n / d, correctly rounded in all but rare cases
__device__ __forceinline__ double divide(double n, double d)
{
    const double r = recip(d);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

constexpr double kLn2Hi = 6.93147180369123816490e-01;     // 0x3fe62e42fee00000: 32 significant bits, k * kLn2Hi is exact
constexpr double kLn2Lo = 1.90821492927058770002e-10;     // ln 2 - kLn2Hi

// log(x), x normal and positive:  x = 2^k m, m in [sqrt(1/2), sqrt(2));  s = (m - 1) / (m + 1);
// log m = 2 s (1 + s^2/3 + s^4/5 + ...), |s| <= 0.1716 so 11 terms reach 6e-19
__device__ __forceinline__ double log_pos(double x)
{
    uint64_t u = to_bits(x);
    int k = (int)(u >> 52) - 1023;
    u = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;          // m in [1, 2)
    double m = from_bits(u);
    if (m > 1.4142135623730951) { m *= 0.5; k += 1; }
    const double f = m - 1.0;
    const double s = divide(f, 2.0 + f);
    const double z = s * s;
    double p = 2.0 / 23.0;
    p = fma_k(p, z, 2.0 / 21.0); p = fma_k(p, z, 2.0 / 19.0); p = fma_k(p, z, 2.0 / 17.0);
    p = fma_k(p, z, 2.0 / 15.0); p = fma_k(p, z, 2.0 / 13.0); p = fma_k(p, z, 2.0 / 11.0);
    p = fma_k(p, z, 2.0 / 9.0);  p = fma_k(p, z, 2.0 / 7.0);  p = fma_k(p, z, 2.0 / 5.0);
    p = fma_k(p, z, 2.0 / 3.0);
    const double dk = (double)k;
    // 2 s + s z p + k ln2: the small terms first
    const double tail = __builtin_fma(s * z, p, dk * kLn2Lo);
    return __builtin_fma(dk, kLn2Hi, __builtin_fma(2.0, s, tail));
}

__device__ __forceinline__ double log_fast(double x)
{
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log(x);      // 0, denormal, negative, inf, nan
    return log_pos(x);
}

// exp(x), |x| <= 700:  x = k ln2 + r, |r| <= 0.3466;  exp r by its series to r^13 / 13! (next term 4e-18 relative)
__device__ __forceinline__ double exp_mid(double x)
{
    const double kd = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-kd, kLn2Hi, x);
    r = __builtin_fma(-kd, kLn2Lo, r);
    double p = 1.0 / 6227020800.0;
    p = fma_k(p, r, 1.0 / 479001600.0); p = fma_k(p, r, 1.0 / 39916800.0); p = fma_k(p, r, 1.0 / 3628800.0);
    p = fma_k(p, r, 1.0 / 362880.0);    p = fma_k(p, r, 1.0 / 40320.0);    p = fma_k(p, r, 1.0 / 5040.0);
    p = fma_k(p, r, 1.0 / 720.0);       p = fma_k(p, r, 1.0 / 120.0);      p = fma_k(p, r, 1.0 / 24.0);
    p = fma_k(p, r, 1.0 / 6.0);         p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p * r, r, r);                                    // r + r^2 (1/2 + ...)
    return ldexp(1.0 + p, (int)kd);
}

__device__ __forceinline__ double exp_fast(double x)
{
    if (!(fabs(x) <= 700.0)) return exp(x);
    return exp_mid(x);
}

// sin / cos.  Reduction: r = x - m pi/2 with pi/2 split in two doubles and FMA: the first subtraction is exact for |m| < 2^27, the second
// rounds once; the neglected third part contributes < m * 1e-33.
constexpr double kPio2Hi = 1.5707963267948966;             // fl(pi/2)
constexpr double kPio2Lo = 6.123233995736766e-17;          // pi/2 - kPio2Hi
constexpr double kTwoOverPi = 0.6366197723675814;

// sin x = (-1)^n sin r with x = n pi + r;  cos x = (-1)^((m + 1) / 2) sin r with x = m pi/2 + r, m odd;  |r| <= pi/2 in both cases, so ONE odd
// polynomial (Taylor to r^21: the next term is 1.3e-18 at pi/2) serves both -- the quadrant scheme above evaluates the sine AND the cosine
// polynomial on |r| <= pi/4 and selects, 33 instructions against 21 here.  The reduction is the same two-constant FMA Cody-Waite with pi/2 as the
// unit (m = 2 n for the sine): m pi/2_hi is exact for |m| < 2^27.
template <bool COS>
__device__ __forceinline__ double sincos_fast(double x)
{
    if (!(fabs(x) < 67108864.0)) return COS ? cos(x) : sin(x);        // |x| >= 2^26, inf, nan: library (Payne-Hanek)
    // number of half-turns n (sine) or n + 1/2 (cosine), as the integer m = 2 n (+ 1) of quarter-turns
    const double h = COS ? __builtin_fma(x, kTwoOverPi * 0.5, -0.5) : x * (kTwoOverPi * 0.5);
    const double nd = __builtin_rint(h);
    const double md = COS ? __builtin_fma(2.0, nd, 1.0) : 2.0 * nd;
    double r = __builtin_fma(-md, kPio2Hi, x);
    r = __builtin_fma(-md, kPio2Lo, r);
    const double z = r * r;
    double p = -1.0 / 51090942171709440000.0;                                                               // -1/21!
    p = fma_k(p, z, 1.0 / 121645100408832000.0); p = fma_k(p, z, -1.0 / 355687428096000.0);    // 1/19!, -1/17!
    p = fma_k(p, z, 1.0 / 1307674368000.0);      p = fma_k(p, z, -1.0 / 6227020800.0);         // 1/15!, -1/13!
    p = fma_k(p, z, 1.0 / 39916800.0);           p = fma_k(p, z, -1.0 / 362880.0);             // 1/11!, -1/9!
    p = fma_k(p, z, 1.0 / 5040.0);               p = fma_k(p, z, -1.0 / 120.0);                // 1/7!, -1/5!
    p = fma_k(p, z, 1.0 / 6.0);                                                                // 1/3!
    const double s = __builtin_fma(-(r * z), p, r);                   // r - r^3 (1/3! - r^2/5! + ...)
    // sign: (-1)^n for the sine, (-1)^(n + 1) for the cosine -- the parity bit of n moved into the sign bit
    const uint64_t flip = (uint64_t)(((uint32_t)(int)nd + (COS ? 1u : 0u)) & 1u) << 63;
    return from_bits(to_bits(s) ^ flip);
}

// x^y for x > 0 normal: exp(y log x) with the rounding error of the product y * log x carried into the result
__device__ __forceinline__ double pow_fast(double x, double y)
{
    if (y == 2.0) return x * x;                                        // numpy and libm both return the correctly rounded square
    if (y == 0.5 && x >= 0.0) return sqrt(x);
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return pow(x, y);
    const double L = log_pos(x);
    const double p = y * L;
    if (!(fabs(p) <= 700.0)) return pow(x, y);
    const double e = __builtin_fma(y, L, -p);
    const double r = exp_mid(p);
    return __builtin_fma(r, e, r);
}

}  // namespace fm
}  // namespace mbx
