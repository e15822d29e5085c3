// mbx_policy.hpp — the RLEPSO / RL-PSO actors as one kernel launch per step.
//
// Reference: src/agent/rlepso_agent.py:9-47 (Actor).  Two MLPs in -> H1 -> H2 -> A with ReLU share their input;
// mu = (tanh(.) + 1)/2, sigma = (tanh(.) + 1)/2 * (max_sigma - min_sigma) + min_sigma, action = clamp(N(mu, sigma), 0, 1).
// At H1 = 64, H2 = 32, A = 35 one instance is 6.5 kMAC: far too small for MFMA tiles and, launched as ~15 library kernels,
// pure launch latency (50 us per generation next to a 230 us generation kernel).  Here a wave owns an instance, the
// 27 KB of float32 weights sit in LDS (transposed, so that lanes read consecutive addresses) and the three layers, the
// squashing and the Normal draw (Philox, keyed like every other draw of the instance) happen in registers / LDS.
#pragma once
#include "mbx_rlepso.hpp"

namespace mbx {

constexpr int kPolicyWaves = kThreads / 64;

__host__ __device__ inline size_t gauss_mlp_lds_bytes(int in, int h1, int h2, int A)
{
    return sizeof(float) * (size_t)(2 * gauss_mlp_net_floats(in, h1, h2, A) + kPolicyWaves * 2 * (h1 + h2));
}

// table_rows == 0: one action per instance from `state` [B, in_dim].
// table_rows  > 0: no sampling; row k of mu_sigma receives the actor's (mu, sigma) at the state k / max_fes, i.e. at every value
//                  RLEPSO's scalar state fes/maxFEs can take (rlepso_optimizer.py:170-171) -- the table k_rlepso_step samples from
//                  when the policy is fused into the generation kernel.
__global__ __launch_bounds__(kThreads) void k_gauss_mlp_policy(BatchParams bp, GaussMlp net, const double* __restrict__ state,
                                                               float* __restrict__ actions, float* __restrict__ mu_sigma, int table_rows)
{
    extern __shared__ __attribute__((aligned(16))) float psm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int IN = net.in_dim, H1 = net.h1, H2 = net.h2, A = net.out_dim;
    const int NW = gauss_mlp_net_floats(IN, H1, H2, A);
    for (int k = tid; k < 2 * NW; k += kThreads) psm[k] = net.w[k];
    float* h1v = psm + 2 * NW + wave * 2 * (H1 + H2);        // [2][H1] hidden activations of this wave's instance
    float* h2v = h1v + 2 * H1;                                // [2][H2]
    const int o_b1 = IN * H1, o_w2 = o_b1 + H1, o_b2 = o_w2 + H1 * H2, o_w3 = o_b2 + H2, o_b3 = o_w3 + H2 * A;
    __syncthreads();
    const int rows = table_rows > 0 ? table_rows : bp.B;
    for (int base = blockIdx.x * kPolicyWaves; base < rows; base += gridDim.x * kPolicyWaves) {   // block-uniform trip count
        const int b = base + wave;
        const bool live = b < rows;
        if (live)
            for (int j = lane; j < 2 * H1; j += 64) {
                const int n = j >= H1, o = j - n * H1;
                const float* W = psm + n * NW;
                float acc = W[o_b1 + o];
                for (int k = 0; k < IN; ++k) {
                    const double x = table_rows > 0 ? (double)b / (double)bp.max_fes : state[(int64_t)b * IN + k];
                    acc += (float)x * W[k * H1 + o];
                }
                h1v[j] = fmaxf(acc, 0.f);
            }
        __syncthreads();
        if (live)
            for (int j = lane; j < 2 * H2; j += 64) {
                const int n = j >= H2, o = j - n * H2;
                const float* W = psm + n * NW;
                const float* h = h1v + n * H1;
                float acc = W[o_b2 + o];
#pragma unroll 8
                for (int k = 0; k < H1; ++k) acc += h[k] * W[o_w2 + k * H2 + o];
                h2v[j] = fmaxf(acc, 0.f);
            }
        __syncthreads();
        if (live) {
            Rng rng{0u, 0u, 0u, 0u};
            if (table_rows == 0) {
                const double* sc = bp.state + (int64_t)b * bp.state_stride + bp.sc_off;
                const uint64_t seed = bp.seeds[b];
                // the action drawn here drives generation gen + 1 of the current episode
                rng = Rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)((int)sc[MBX_SC_GEN] + 1), (uint32_t)(int)sc[MBX_SC_EPISODE]};
            }
            for (int j = lane; j < A; j += 64) {
                const float* Wm = psm;
                const float* Ws = psm + NW;
                float am = Wm[o_b3 + j], as = Ws[o_b3 + j];
#pragma unroll 8
                for (int k = 0; k < H2; ++k) { am += h2v[k] * Wm[o_w3 + k * A + j]; as += h2v[H2 + k] * Ws[o_w3 + k * A + j]; }
                float mu, sigma;
                gauss_head(net, am, as, mu, sigma);
                if (table_rows == 0) actions[(int64_t)b * A + j] = sample_action(rng, j, mu, sigma, net.variant);
                if (mu_sigma) { mu_sigma[((int64_t)b * 2) * A + j] = mu; mu_sigma[((int64_t)b * 2 + 1) * A + j] = sigma; }
            }
        }
        __syncthreads();
    }
}

}  // namespace mbx
