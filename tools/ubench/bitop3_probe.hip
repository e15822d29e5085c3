// bitop3_probe.hip -- gfx950 has a three-input bit operation (v_bitop3_b32, truth table in the modifier; 0x96 = a ^ b ^ c) that gfx942 lacks;
// Philox's round function uses it through __builtin_amdgcn_bitop3_b32 (mbx_device.hpp).  hipcc --offload-arch=gfx950 -O2 -o tools/ubench/bitop3_probe tools/ubench/bitop3_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) { const unsigned a = p[threadIdx.x], b = p[threadIdx.x + 64], c = p[threadIdx.x + 128]; p[threadIdx.x + 192] = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
int main()
{
    unsigned h[256], *d;
    for (int i = 0; i < 256; ++i) h[i] = i * 2654435761u;
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[192 + i] != (h[i] ^ h[64 + i] ^ h[128 + i]);
    printf("{\"v_bitop3_b32_xor3_mismatches\": %d}\n", bad);
    return bad != 0;
}
