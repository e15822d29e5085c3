// mfma_f64_probe.hip -- what v_mfma_f64_16x16x4_f64 computes on gfx950, bit for bit (run on the GPU box: ./tools/ubench/mfma_f64_probe).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mfma_f64_probe tools/ubench/mfma_f64_probe.hip
// Question: is D = A B + C, chained over K = 12 (three instructions), the same as acc = fma(a_k, b_k, acc) for k = 0 .. 11 in ascending
// order (one rounding per product-accumulate)?  If so the CPU oracle can mirror the MFMA matvec with fma() and stay bit-identical.
// Layout used (guides/cdna_hip_programming.md): A lane l -> A[l & 15][l >> 4], B lane l -> B[l >> 4][l & 15], D reg r of lane l -> D[(l >> 4) + 4 r][l & 15].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void k_probe(const double* A, const double* B, const double* C0, double* D, int K)
{   // A [16][K], B [K][16], C0 / D [16][16]
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    d4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C0[(q + 4 * r) * 16 + i];
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = A[i * K + k0 + q], b = B[(k0 + q) * 16 + i];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(q + 4 * r) * 16 + i] = acc[r];
}

static double rnd(int mode)
{
    double u = (double)rand() / RAND_MAX * 2 - 1;
    if (mode == 1) u *= pow(10., (rand() % 13) - 6);
    if (mode == 2) u = ldexp(u, (rand() % 80) - 40);
    return u;
}

int main()
{
    const int K = 12;
    double hA[16 * K], hB[K * 16], hC[256], hD[256];
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    long bad_chain = 0, bad_mul_add = 0, total = 0; double worst = 0;
    for (int trial = 0; trial < 300; ++trial) {
        const int mode = trial % 3;
        for (int t = 0; t < 16 * K; ++t) { hA[t] = rnd(mode); hB[t] = rnd(mode); }
        for (int t = 0; t < 256; ++t) hC[t] = trial < 150 ? 0. : rnd(mode);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice); hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        k_probe<<<1, 64>>>(dA, dB, dC, dD, K);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double f = hC[i * 16 + j], s = hC[i * 16 + j];
            for (int k = 0; k < K; ++k) { f = fma(hA[i * K + k], hB[k * 16 + j], f); volatile double p = hA[i * K + k] * hB[k * 16 + j]; s = s + p; }
            const double g = hD[i * 16 + j];
            ++total;
            if (memcmp(&f, &g, 8)) { ++bad_chain; const double e = fabs(f - g) / (fabs(f) + 1e-300); if (e > worst) worst = e; }
            if (memcmp(&s, &g, 8)) ++bad_mul_add;
        }
    }
    printf("{\"outputs\": %ld, \"differ_from_fma_chain_k_ascending\": %ld, \"worst_rel\": %.3g, \"differ_from_mul_then_add\": %ld}\n", total, bad_chain, worst, bad_mul_add);
    return 0;
}
