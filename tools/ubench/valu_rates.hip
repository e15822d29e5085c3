// valu_rates.hip — issue cost (shader cycles per wave-instruction per SIMD) of the VALU / LDS instructions k_rlepso_step is made of.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip ;  run on the GPU box: ./valu_rates
// Each test: W waves on every SIMD of one CU (block = 256*W threads, 1 block), each wave runs ITER x 16 independent
// instructions; cycles = s_memtime delta of wave 0; reported = cycles * 1 / (ITER * 16 * W)  (per SIMD: W waves share it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 8192

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, DECL, BODY, SINK)                                                            \
    __global__ void NAME(unsigned long long* out, double seed)                                     \
    {                                                                                              \
        DECL                                                                                       \
        __syncthreads();                                                                           \
        unsigned long long t0 = __builtin_readcyclecounter();                                      \
        for (int it = 0; it < ITER; ++it) { BODY }                                                 \
        unsigned long long t1 = __builtin_readcyclecounter();                                      \
        SINK                                                                                       \
        if (threadIdx.x == 0) out[0] = t1 - t0;                                                    \
    }

// 16 independent double accumulators
#define DECL_D double a[16]; double x = seed + threadIdx.x * 1e-9, y = seed * 0.5; _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed + i;
#define SINK_D { double s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i]; if (s == 12345.678) out[1] = (unsigned long long)s; }
#define DECL_F float a[16]; float x = (float)seed + threadIdx.x * 1e-6f, y = (float)seed * 0.5f; _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = (float)seed + i;
#define SINK_F { float s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i]; if (s == 12345.678f) out[1] = (unsigned long long)s; }
#define DECL_U uint32_t a[16]; uint32_t x = (uint32_t)seed + threadIdx.x, y = 0x9E3779B9u; _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = (uint32_t)seed + i;
#define SINK_U { uint32_t s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i]; if (s == 12345u) out[1] = s; }

#define A1(op) asm volatile(op " %0, %0, %1" : "+v"(a[i_]) : "v"(x));
#define OPD(name, op) KERNEL(name, DECL_D, _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { asm volatile(op : "+v"(a[i_]) : "v"(x), "v"(y)); }, SINK_D)
#define OPF(name, op) KERNEL(name, DECL_F, _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { asm volatile(op : "+v"(a[i_]) : "v"(x), "v"(y)); }, SINK_F)
#define OPU(name, op) KERNEL(name, DECL_U, _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { asm volatile(op : "+v"(a[i_]) : "v"(x), "v"(y)); }, SINK_U)

OPD(k_add_f64, "v_add_f64 %0, %0, %1")
OPD(k_mul_f64, "v_mul_f64 %0, %0, %1")
OPD(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
OPD(k_min_f64, "v_min_f64 %0, %0, %1")
OPD(k_mov_b64, "v_mov_b64 %0, %1")
OPD(k_rcp_f64, "v_rcp_f64 %0, %0")
OPD(k_rsq_f64, "v_rsq_f64 %0, %0")
OPD(k_sqrt_f64, "v_sqrt_f64 %0, %0")
OPD(k_fract_f64, "v_fract_f64 %0, %0")
OPD(k_rndne_f64, "v_rndne_f64 %0, %0")
OPD(k_floor_f64, "v_floor_f64 %0, %0")
OPD(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3")
OPD(k_frexp_mant_f64, "v_frexp_mant_f64 %0, %0")
OPD(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
OPD(k_cmpx_f64, "v_cmpx_lt_f64 %0, %1\n s_mov_b64 exec, -1")
OPD(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %0")
OPD(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
OPD(k_trig_preop_f64, "v_trig_preop_f64 %0, %0, 1")
OPF(k_add_f32, "v_add_f32 %0, %0, %1")
OPF(k_mul_f32, "v_mul_f32 %0, %0, %1")
OPF(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OPF(k_rcp_f32, "v_rcp_f32 %0, %0")
OPF(k_exp_f32, "v_exp_f32 %0, %0")
OPF(k_log_f32, "v_log_f32 %0, %0")
OPF(k_sqrt_f32, "v_sqrt_f32 %0, %0")
OPF(k_sin_f32, "v_sin_f32 %0, %0")
OPF(k_min_f32, "v_min_f32 %0, %0, %1")
OPF(k_med3_f32, "v_med3_f32 %0, %0, %1, %2")
OPF(k_addabs_f32, "v_add_f32 %0, |%0|, %1")
OPU(k_mov_b32, "v_mov_b32 %0, %1")
OPU(k_add_u32, "v_add_u32 %0, %0, %1")
OPU(k_xor_b32, "v_xor_b32 %0, %0, %1")
OPU(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
OPU(k_min_u32, "v_min_u32 %0, %0, %1")
OPU(k_med3_u32, "v_med3_u32 %0, %0, %1, %2")
OPU(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
OPU(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
OPU(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
OPU(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
OPU(k_bperm, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
OPU(k_bperm_nowait, "ds_bpermute_b32 %0, %1, %0")
OPU(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
OPU(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")


OPF(k_cmp_lt_f32, "v_cmp_lt_f32 s[10:11], %0, %1")
OPU(k_cmp_lt_u32, "v_cmp_lt_u32 s[10:11], %0, %1")
OPU(k_cndmask_sgpr, "v_cndmask_b32 %0, %0, %1, s[10:11]")
OPF(k_cmp_cndmask_f32, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")
OPF(k_max_f32, "v_max_f32 %0, %0, %1")
OPF(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")
OPF(k_sub_f32, "v_sub_f32 %0, %0, %1")
OPU(k_and_b32, "v_and_b32 %0, %0, %1")
OPU(k_or_b32, "v_or_b32 %0, %0, %1")
OPU(k_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
OPU(k_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
OPU(k_sub_u32, "v_sub_u32 %0, %0, %1")
OPU(k_max_u32, "v_max_u32 %0, %0, %1")
OPU(k_max_i32, "v_max_i32 %0, %0, %1")
OPU(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OPU(k_add_dpp, "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf")

// 64-bit products: v_mad_u64_u32 writes a pair
__global__ void k_mad_u64_u32(unsigned long long* out, double seed)
{
    unsigned long long a[16]; uint32_t x = (uint32_t)seed + threadIdx.x, y = 0xD2511F53u;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = (unsigned long long)seed + i;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345ull) out[1] = s;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
// conversions
__global__ void k_cvt_f32_f64(unsigned long long* out, double seed)
{
    double a[16]; float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; r[i] = 0; }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(a[i]));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 12345.f) out[1] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_cvt_f64_f32(unsigned long long* out, double seed)
{
    float a[16]; double r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = (float)seed + i + threadIdx.x; r[i] = 0; }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    if (s == 12345.) out[1] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_pk_mul_f32(unsigned long long* out, double seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[16]; f2 x = {(float)seed, (float)seed + threadIdx.x};
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{(float)seed + i, 1.f};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    if (s == 12345.f) out[1] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_pk_add_f32(unsigned long long* out, double seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[16]; f2 x = {(float)seed, (float)seed + threadIdx.x};
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{(float)seed + i, 1.f};
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    if (s == 12345.f) out[1] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
// LDS reads: broadcast b64 / b128, strided
__global__ void k_ds_read_b64_bcast(unsigned long long* out, double seed)
{
    __shared__ double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = seed + i;
    __syncthreads();
    double acc = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = sm[(it * 16 + i) & 4095];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(v[i]));
        acc += v[0];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 12345.) out[1] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_ds_read_b128_5addr(unsigned long long* out, double seed)
{
    __shared__ __attribute__((aligned(16))) double sm[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = seed + i;
    __syncthreads();
    double acc = 0;
    const int off = (threadIdx.x % 5) * 2;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        double2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *(const double2*)&sm[(((it * 16 + i) * 10) & 4080) + off];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(v[i].x), "v"(v[i].y));
        acc += v[0].x;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 12345.) out[1] = 1;
    if (threadIdx.x == 0) out[0] = t1 - t0;
}

struct Test { const char* name; void (*fn)(unsigned long long*, double); int ninstr; };
#define T(n) {#n, n, 1}

int main()
{
    std::vector<Test> tests = {
        T(k_add_f64), T(k_mul_f64), T(k_fma_f64), T(k_min_f64), T(k_mov_b64), T(k_rcp_f64), T(k_rsq_f64), T(k_sqrt_f64), T(k_fract_f64),
        T(k_rndne_f64), T(k_floor_f64), T(k_ldexp_f64), T(k_frexp_mant_f64), T(k_cmp_f64), {"k_cmpx_f64+s_mov_exec", k_cmpx_f64, 2}, T(k_div_scale_f64), T(k_div_fixup_f64),
        T(k_trig_preop_f64), T(k_cvt_f32_f64), T(k_cvt_f64_f32),
        T(k_add_f32), T(k_mul_f32), T(k_fma_f32), T(k_pk_mul_f32), T(k_pk_add_f32), T(k_rcp_f32), T(k_exp_f32), T(k_log_f32), T(k_sqrt_f32), T(k_sin_f32),
        T(k_min_f32), T(k_med3_f32), T(k_addabs_f32),
        T(k_mov_b32), T(k_add_u32), T(k_xor_b32), T(k_and_or_b32), T(k_min_u32), T(k_med3_u32), T(k_mul_lo_u32), T(k_mul_hi_u32), T(k_mul_u24),
        T(k_mad_u64_u32), T(k_cmp_lt_f32), T(k_cmp_lt_u32), T(k_cndmask_sgpr), {"k_cmp+cndmask_f32", k_cmp_cndmask_f32, 2}, T(k_max_f32), T(k_min3_f32), T(k_sub_f32), T(k_and_b32), T(k_or_b32), T(k_bfi_b32), T(k_lshlrev_b32), T(k_sub_u32), T(k_max_u32), T(k_max_i32), T(k_mov_dpp), T(k_add_dpp), T(k_lshl_add), T(k_alignbit), T(k_bperm_nowait), T(k_ds_read_b64_bcast), T(k_ds_read_b128_5addr),
    };
    unsigned long long* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // whole chip, W waves per SIMD: 256 CUs x (W/4) blocks of 1024 threads (W = 4, 8) or 256 blocks of 256*W threads (W = 1, 2); wall time by HIP
    // events -> ns per wave-instruction per SIMD; x 2.4 = cycles at the nominal clock (the column "ticks" is wave 0's s_memtime delta / instr)
    printf("%-24s %8s %8s %8s %8s | %8s   ns per wave-instruction per SIMD by waves per SIMD (x2.4 = cycles @2.4 GHz); ticks/instr of one wave alone\n", "instr", "1w", "2w", "4w", "8w", "ticks1w");
    for (auto& t : tests) {
        double r[4], ticks = 0;
        for (int wi = 0; wi < 4; ++wi) {
            const int w = 1 << wi;
            const int threads = w >= 4 ? 1024 : 256 * w;
            const int blocks = 256 * (w >= 4 ? w / 4 : 1);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, d, 1.25);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(threads), 0, 0, d, 1.25);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            r[wi] = (double)ms * 1e6 / ((double)ITER * 16 * w * t.ninstr);
            if (wi == 0) { unsigned long long h = 0; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); ticks = (double)h / ((double)ITER * 16 * t.ninstr); }
        }
        printf("%-24s %8.3f %8.3f %8.3f %8.3f | %8.2f\n", t.name, r[0], r[1], r[2], r[3], ticks);
    }
    return 0;
}
