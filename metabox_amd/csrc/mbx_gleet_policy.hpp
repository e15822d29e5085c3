// mbx_gleet_policy.hpp — GLEET's attention actor as ONE kernel launch per generation (reference: src/agent/gleet_agent.py:314-444,
// blocks from src/agent/networks.py:47-365; architecture constants :31-45: E = 16, 4 heads of 4, FF 16, node_dim 9, heads 32-8-1).
//
// Per swarm of ps particles with 27 state values each:
//   h = Embed(features);  h = EncoderLayer(h)                      self-attention + residual + swarm-wide normalisation, FF + same
//   q = Embed2([Embed(exploration memory) | Embed(exploitation memory)]);  z = EncoderLayer(h, queries = q)
//   mu, sigma = MLP_16-32-8-1(z) x 2, squashed;  action = clamp(Normal(mu, sigma), 0, 1)
// As PyTorch ops this is ~60 launches over [4, B, ps, ps] attention maps (5 ms per generation of 4096 swarms, 30x the generation
// kernel).  Here one workgroup owns a swarm and one thread a particle: its row of every activation stays in registers, keys /
// values of the swarm sit in LDS (every lane reads the same key at the same time: LDS broadcast), softmax is evaluated online
// (running maximum / sum), the swarm-wide normalisations are two-pass block reductions, and the 5426 weights are read with
// wave-uniform, compile-time indices straight from global memory, i.e. as scalar loads through the constant cache: they cost
// neither LDS nor vector registers.  13 KB of LDS per swarm (the state block, later overwritten by the keys / values).  float32 throughout, like the reference's modules; summation orders differ from the library GEMMs,
// so agreement with PyTorch is to float32 round-off (tests: 2e-4 on mu / sigma), not bitwise.
#pragma once
#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"   // BatchParams, Rng, sample_action

namespace mbx {

// The policy is float32 and has no bit-exact counterpart (PyTorch's GEMMs sum in another order), so this file lets the compiler fuse
// multiply-adds; the optimizers' float64 arithmetic elsewhere keeps -ffp-contract=off.
#pragma clang fp contract(fast)

constexpr int kGpE = 16, kGpH = 4, kGpDk = 4, kGpFF = 16, kGpNode = 9, kGpH1 = 32, kGpH2 = 8;
#ifndef MBX_GP_OCC
#define MBX_GP_OCC __attribute__((amdgpu_waves_per_eu(3)))      // 177 -> 168 VGPRs: three waves per SIMD, measured 6 % faster
#endif
constexpr int kGpThreads = 128;          // 2 waves: ps <= 128 particles, one per thread

// packed float32 weights (GLEET_Agent actor state_dict order, tensors flattened row-major as PyTorch stores them)
struct GpOff {
    static constexpr int we = 0;                                     // embedder.embedder.weight [E, 9]
    static constexpr int attn = kGpH * kGpE * kGpDk;                 // one projection tensor [H, E, dk] (W_out is [H, dk, E]: same size)
    static constexpr int layer = 4 * attn + kGpFF * kGpE + kGpFF + kGpE * kGpFF + kGpE;   // W_query W_key W_val W_out FF.0.w FF.0.b FF.2.w FF.2.b
    static constexpr int enc = we + kGpE * kGpNode;
    static constexpr int wd = enc + layer;                           // embedder_for_decoder.embedder.weight [E, 2E]
    static constexpr int dec = wd + kGpE * 2 * kGpE;
    static constexpr int head = kGpH1 * kGpE + kGpH1 + kGpH2 * kGpH1 + kGpH2 + kGpH2 + 1;
    static constexpr int mu = dec + layer;
    static constexpr int sigma = mu + head;
    static constexpr int total = sigma + head;                       // 5426
};

struct GleetActor { const float* w; float min_sigma, max_sigma; };

__host__ __device__ inline size_t gleet_policy_lds_bytes(int NP)
{
    // X [NP, 27], later K [NP, E] | V [NP, E] in the same storage; reduction scratch
    const int region = NP * 27 > 2 * NP * kGpE ? NP * 27 : 2 * NP * kGpE;
    return sizeof(float) * (size_t)(region + 64);
}

// sum over the workgroup of one value per thread (inactive threads pass 0); every thread gets the total
__device__ __forceinline__ float gp_block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                                 // previous use of red is over
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1];
}

// 'layer' normalisation of the reference (networks.py:69-72): statistics over the whole [ps, E] block, unbiased variance
__device__ __forceinline__ void gp_swarm_norm(float (&x)[kGpE], bool live, int NP, float* red)
{
    float s = 0.f;
    if (live) for (int k = 0; k < kGpE; ++k) s += x[k];
    const float n = (float)(NP * kGpE);
    const float mean = gp_block_sum(s, red) / n;
    float q = 0.f;
    if (live) for (int k = 0; k < kGpE; ++k) { const float t = x[k] - mean; q += t * t; }
    const float var = gp_block_sum(q, red) / (n - 1.f);
    const float inv = 1.f / sqrtf(var + 1e-05f);
    for (int k = 0; k < kGpE; ++k) x[k] = (x[k] - mean) * inv;
}

// MultiHeadEncoder (networks.py:256-340): x <- norm(MHA(keys/values from x, queries from qsrc) + x); x <- norm(FF(x) + x).
// W points at this layer's weights in LDS; KL / VL are the swarm's key / value rows in LDS.
__device__ __forceinline__ void gp_encoder_layer(float (&x)[kGpE], const float (&qsrc)[kGpE], const float* __restrict__ W, float* KL, float* VL,
                                                 bool live, int NP, int i, float* red)
{
    const float* Wq = W; const float* Wk = W + GpOff::attn; const float* Wv = W + 2 * GpOff::attn; const float* Wo = W + 3 * GpOff::attn;
    const float* F1 = W + 4 * GpOff::attn; const float* B1 = F1 + kGpFF * kGpE; const float* F2 = B1 + kGpFF; const float* B2 = F2 + kGpE * kGpFF;
    float Q[kGpH * kGpDk];
    if (live) {
#pragma unroll
        for (int h = 0; h < kGpH; ++h)
#pragma unroll
            for (int c = 0; c < kGpDk; ++c) {
                float q = 0.f, k = 0.f, v = 0.f;
#pragma unroll
                for (int e = 0; e < kGpE; ++e) {
                    const int w = (h * kGpE + e) * kGpDk + c;
                    q += qsrc[e] * Wq[w]; k += x[e] * Wk[w]; v += x[e] * Wv[w];
                }
                Q[h * kGpDk + c] = q * (0.5f * 1.4426950408889634f); KL[i * kGpE + h * kGpDk + c] = k; VL[i * kGpE + h * kGpDk + c] = v;
            }
    }
    __syncthreads();
    float out[kGpE];
#pragma unroll
    for (int e = 0; e < kGpE; ++e) out[e] = 0.f;
    if (live) {
#pragma unroll
        for (int h = 0; h < kGpH; ++h) {
            float m = -INFINITY, l = 0.f, acc[kGpDk] = {0.f, 0.f, 0.f, 0.f};
            // online softmax over the keys, four at a time: one rescale of the running sums per chunk instead of per key.
            // Scores are in units of log2 e (Q was pre-scaled by norm_factor * log2 e), so exp is the bare v_exp_f32.
            for (int j0 = 0; j0 < NP; j0 += 4) {
                float sc4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* kj = KL + (j0 + u < NP ? j0 + u : NP - 1) * kGpE + h * kGpDk;
                    const float d = Q[h * kGpDk] * kj[0] + Q[h * kGpDk + 1] * kj[1] + Q[h * kGpDk + 2] * kj[2] + Q[h * kGpDk + 3] * kj[3];
                    sc4[u] = j0 + u < NP ? d : -INFINITY;
                }
                const float mn = fmaxf(fmaxf(m, fmaxf(sc4[0], sc4[1])), fmaxf(sc4[2], sc4[3]));
                const float scale = __builtin_amdgcn_exp2f(m - mn);
                l *= scale;
#pragma unroll
                for (int c = 0; c < kGpDk; ++c) acc[c] *= scale;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* vj = VL + (j0 + u < NP ? j0 + u : NP - 1) * kGpE + h * kGpDk;
                    const float p = __builtin_amdgcn_exp2f(sc4[u] - mn);     // exp2(-inf) = 0 for the padded tail
                    l += p;
#pragma unroll
                    for (int c = 0; c < kGpDk; ++c) acc[c] += p * vj[c];
                }
                m = mn;
            }
            const float inv = 1.f / l;
#pragma unroll
            for (int c = 0; c < kGpDk; ++c) {
                const float hv = acc[c] * inv;
#pragma unroll
                for (int e = 0; e < kGpE; ++e) out[e] += hv * Wo[(h * kGpDk + c) * kGpE + e];
            }
        }
#pragma unroll
        for (int e = 0; e < kGpE; ++e) x[e] = out[e] + x[e];
    }
    gp_swarm_norm(x, live, NP, red);
    if (live) {
        float hid[kGpFF];
#pragma unroll
        for (int o = 0; o < kGpFF; ++o) {
            float a = B1[o];
#pragma unroll
            for (int e = 0; e < kGpE; ++e) a += x[e] * F1[o * kGpE + e];
            hid[o] = fmaxf(a, 0.f);
        }
#pragma unroll
        for (int e = 0; e < kGpE; ++e) {
            float a = B2[e];
#pragma unroll
            for (int o = 0; o < kGpFF; ++o) a += hid[o] * F2[e * kGpFF + o];
            out[e] = a + x[e];
        }
#pragma unroll
        for (int e = 0; e < kGpE; ++e) x[e] = out[e];
    }
    gp_swarm_norm(x, live, NP, red);
}

// 16 -> 32 -> 8 -> 1 with LeakyReLU(0.01) (gleet_agent.py:365-370)
__device__ __forceinline__ float gp_head(const float (&z)[kGpE], const float* __restrict__ W)
{
    const float* W1 = W; const float* B1 = W1 + kGpH1 * kGpE; const float* W2 = B1 + kGpH1; const float* B2 = W2 + kGpH2 * kGpH1;
    const float* W3 = B2 + kGpH2; const float* B3 = W3 + kGpH2;
    float h1[kGpH1];
#pragma unroll
    for (int o = 0; o < kGpH1; ++o) {
        float a = B1[o];
#pragma unroll
        for (int e = 0; e < kGpE; ++e) a += z[e] * W1[o * kGpE + e];
        h1[o] = a > 0.f ? a : 0.01f * a;
    }
    float out = B3[0];
#pragma unroll
    for (int o = 0; o < kGpH2; ++o) {
        float a = B2[o];
#pragma unroll
        for (int k = 0; k < kGpH1; ++k) a += h1[k] * W2[o * kGpH1 + k];
        a = a > 0.f ? a : 0.01f * a;
        out += a * W3[o];
    }
    return out;
}

__global__ __launch_bounds__(kGpThreads) MBX_GP_OCC void k_gleet_policy(BatchParams bp, GleetActor net, const double* __restrict__ state,
                                                             float* __restrict__ actions, float* __restrict__ mu_sigma)
{
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int b = blockIdx.x, i = threadIdx.x, NP = bp.NP;
    const float* __restrict__ W = net.w;                             // uniform, constant-indexed reads: scalar loads
    float* X = gsm;
    float* KL = gsm;                                                 // overwrites X once the embeddings are in registers
    float* VL = KL + NP * kGpE;
    float* red = gsm + (NP * 27 > 2 * NP * kGpE ? NP * 27 : 2 * NP * kGpE);
    const double* sb = state + (int64_t)b * NP * 27;
    for (int k = i; k < NP * 27; k += kGpThreads) X[k] = (float)sb[k];
    __syncthreads();
    const bool live = i < NP;
    float h[kGpE], dq[kGpE];
    if (live) {
        const float* xi = X + i * 27;
        const float* We = W + GpOff::we;
        float e1[kGpE], e2[kGpE];
#pragma unroll
        for (int o = 0; o < kGpE; ++o) {
            float a = 0.f, p = 0.f, g = 0.f;
#pragma unroll
            for (int k = 0; k < kGpNode; ++k) { const float w = We[o * kGpNode + k]; a += xi[k] * w; p += xi[kGpNode + k] * w; g += xi[2 * kGpNode + k] * w; }
            h[o] = a; e1[o] = p; e2[o] = g;
        }
        const float* Wd = W + GpOff::wd;
#pragma unroll
        for (int o = 0; o < kGpE; ++o) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < kGpE; ++k) a += e1[k] * Wd[o * 2 * kGpE + k];
#pragma unroll
            for (int k = 0; k < kGpE; ++k) a += e2[k] * Wd[o * 2 * kGpE + kGpE + k];
            dq[o] = a;
        }
    }
    __syncthreads();                                                 // X is dead: its storage becomes K / V
    gp_encoder_layer(h, h, W + GpOff::enc, KL, VL, live, NP, i, red);
    __syncthreads();                                                 // every lane is done with the encoder's keys / values
    gp_encoder_layer(h, dq, W + GpOff::dec, KL, VL, live, NP, i, red);
    if (live) {
        const float am = gp_head(h, W + GpOff::mu), as = gp_head(h, W + GpOff::sigma);
        const float mu = (tanhf(am) + 1.f) / 2.f;
        const float sigma = (tanhf(as) + 1.f) / 2.f * (net.max_sigma - net.min_sigma) + net.min_sigma;
        const double* sc = bp.state + (int64_t)b * bp.state_stride + bp.sc_off;
        const uint64_t seed = bp.seeds[b];
        const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)((int)sc[MBX_SC_GEN] + 1), (uint32_t)(int)sc[MBX_SC_EPISODE]};
        actions[(int64_t)b * NP + i] = sample_action(rng, i, mu, sigma, MBX_POLICY_RLEPSO);      // clamp(Normal(mu, sigma), 0, 1)
        if (mu_sigma) { mu_sigma[((int64_t)b * 2) * NP + i] = mu; mu_sigma[((int64_t)b * 2 + 1) * NP + i] = sigma; }
    }
}

#pragma clang fp contract(off)      // back to the translation unit's -ffp-contract=off

}  // namespace mbx
