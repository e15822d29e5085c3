// mbx_lstm_policy.hpp — LDE's PolicyNet as ONE kernel launch per generation.
//
// Reference: src/agent/lde_agent.py:8-29 (PolicyNet: one LSTM cell NP+10 -> 50, Linear 50 -> 2 NP for mu, Linear 50 -> 2 NP + sigmoid for
// sigma) and :147-163 (rollout: action = clip(Normal(mu, sigma).sample(), 0, 1), (h, c) carried).  As PyTorch ops this is an LSTM call, two
// addmm, a sigmoid, a randn, a clip and their glue, ~0.18 ms of launches next to a 0.63 ms generation kernel (config 3).  The network is
// 32.6 k parameters and 32 kMAC per instance: far below what a library GEMM needs to amortise a launch, and K = 50 / 60 cannot fill an
// MFMA pipeline, so -- like k_gauss_mlp_policy -- it is plain float32 VALU work arranged for coalescing:
//   * a workgroup owns a tile of TI = 16 instances; their inputs [x | h] sit in LDS as XS[k][i] (k-major: one 16-byte LDS read feeds four
//     multiply-adds);
//   * thread u owns output unit u (a gate row of the LSTM, then a row of the mu / sigma heads) for all 16 instances: 16 accumulators in
//     registers, ONE weight per k, read from the TRANSPOSED weight matrix Wt[k][u] so that the lanes of a wave read consecutive words
//     (the 130 KB of weights are shared by every workgroup and stay in L2);
//   * gate nonlinearities, the cell update, both heads, the sigmoid and the Normal draw (Philox, counter (j, MBX_SITE_POLICY, gen + 1,
//     episode) like every other fused policy) happen in the same launch; (h, c) are updated in place.
// Arithmetic is float32 with the reference's association up to the order of the dot products (k ascending; torch's GEMM kernels sum in
// tiles): (mu, sigma, h', c') agree with the recorded reference I/O pairs to 5e-6 (tests/test_policy_io.py).
#pragma once
#include "mbx_rlepso.hpp"

namespace mbx {

struct LstmPolicy {
    const float* w;      // packed, see mbx_lstm_policy in include/mbx.h: WihT [IN][4H] | WhhT [H][4H] | b [4H] | WmuT [H][A] | WsgT [H][A] | bmu [A] | bsg [A]
    int32_t in_dim, hidden, out_dim;
};

constexpr int kLstmTile = 16;

__host__ __device__ inline int64_t lstm_policy_floats(int IN, int H, int A) { return (int64_t)(IN + H) * 4 * H + 4 * H + 2 * (int64_t)H * A + 2 * A; }
__host__ __device__ inline size_t lstm_policy_lds_bytes(int IN, int H) { return sizeof(float) * (size_t)((IN + H) * kLstmTile + 4 * H * kLstmTile + H * kLstmTile); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(kThreads) void k_lstm_policy(BatchParams bp, LstmPolicy net, const double* __restrict__ state, float* __restrict__ hbuf,
                                                          float* __restrict__ cbuf, float* __restrict__ actions, float* __restrict__ mu_sigma)
{
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    constexpr int TI = kLstmTile;
    const int tid = threadIdx.x;
    const int IN = net.in_dim, H = net.hidden, A = net.out_dim, G4 = 4 * H, K1 = IN + H;
    float* XS = lsm;                      // [K1][TI]   inputs: state (IN rows) then h (H rows)
    float* GT = XS + K1 * TI;             // [4H][TI]   gate pre-activations
    float* HN = GT + G4 * TI;             // [H][TI]    new hidden state
    const float* WihT = net.w;
    const float* WhhT = WihT + (int64_t)IN * G4;
    const float* bg = WhhT + (int64_t)H * G4;
    const float* WmuT = bg + G4;
    const float* WsgT = WmuT + (int64_t)H * A;
    const float* bmu = WsgT + (int64_t)H * A;
    const float* bsg = bmu + A;
    const int b0 = blockIdx.x * TI;
    const int nb = bp.B - b0 < TI ? bp.B - b0 : TI;
    // ---- stage [x | h] of the tile, k-major
    for (int t = tid; t < K1 * TI; t += kThreads) {
        const int i = t / K1, k = t - i * K1;                  // consecutive threads read consecutive words of one instance
        float v = 0.f;
        if (i < nb) v = k < IN ? (float)state[(int64_t)(b0 + i) * IN + k] : hbuf[(int64_t)(b0 + i) * H + (k - IN)];
        XS[k * TI + i] = v;
    }
    __syncthreads();
    // ---- LSTM gates: unit u of [i | f | g | o] (torch.nn.LSTM's row order), all TI instances
    for (int u = tid; u < G4; u += kThreads) {
        float acc[TI];
        const float bias = bg[u];
#pragma unroll
        for (int i = 0; i < TI; ++i) acc[i] = bias;
#pragma unroll 4
        for (int k = 0; k < K1; ++k) {
            const float w = k < IN ? WihT[(int64_t)k * G4 + u] : WhhT[(int64_t)(k - IN) * G4 + u];
            const float4* x4 = (const float4*)(XS + k * TI);
#pragma unroll
            for (int q = 0; q < TI / 4; ++q) {
                const float4 x = x4[q];
                acc[4 * q] += w * x.x; acc[4 * q + 1] += w * x.y; acc[4 * q + 2] += w * x.z; acc[4 * q + 3] += w * x.w;
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) GT[u * TI + i] = acc[i];
    }
    __syncthreads();
    // ---- cell update: c' = sigmoid(f) c + sigmoid(i) tanh(g);  h' = sigmoid(o) tanh(c')
    for (int t = tid; t < H * TI; t += kThreads) {
        const int j = t / TI, i = t - j * TI;
        if (i < nb) {
            const float gi = sigmoidf_(GT[j * TI + i]), gf = sigmoidf_(GT[(H + j) * TI + i]);
            const float gg = tanhf(GT[(2 * H + j) * TI + i]), go = sigmoidf_(GT[(3 * H + j) * TI + i]);
            const float cn = gf * cbuf[(int64_t)(b0 + i) * H + j] + gi * gg;
            const float hn = go * tanhf(cn);
            cbuf[(int64_t)(b0 + i) * H + j] = cn;
            hbuf[(int64_t)(b0 + i) * H + j] = hn;
            HN[j * TI + i] = hn;
        } else HN[j * TI + i] = 0.f;
    }
    __syncthreads();
    // ---- heads + sampling: unit j of mu and of sigma for all TI instances
    for (int j = tid; j < A; j += kThreads) {
        float am[TI], as[TI];
        const float b1 = bmu[j], b2 = bsg[j];
#pragma unroll
        for (int i = 0; i < TI; ++i) { am[i] = b1; as[i] = b2; }
#pragma unroll 2
        for (int k = 0; k < H; ++k) {
            const float wm = WmuT[(int64_t)k * A + j], ws = WsgT[(int64_t)k * A + j];
            const float4* x4 = (const float4*)(HN + k * TI);
#pragma unroll
            for (int q = 0; q < TI / 4; ++q) {
                const float4 x = x4[q];
                am[4 * q] += wm * x.x; am[4 * q + 1] += wm * x.y; am[4 * q + 2] += wm * x.z; am[4 * q + 3] += wm * x.w;
                as[4 * q] += ws * x.x; as[4 * q + 1] += ws * x.y; as[4 * q + 2] += ws * x.z; as[4 * q + 3] += ws * x.w;
            }
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (i < nb) {
                const int b = b0 + i;
                const float mu = am[i], sigma = sigmoidf_(as[i]);
                if (mu_sigma) { mu_sigma[((int64_t)b * 2) * A + j] = mu; mu_sigma[((int64_t)b * 2 + 1) * A + j] = sigma; }
                if (actions) {
                    const double* sc = bp.state + (int64_t)b * bp.state_stride + bp.sc_off;
                    const uint64_t seed = bp.seeds[b];
                    // the action drawn here drives generation gen + 1 of the current episode
                    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)((int)sc[MBX_SC_GEN] + 1), (uint32_t)(int)sc[MBX_SC_EPISODE]};
                    actions[(int64_t)b * A + j] = sample_action(rng, j, mu, sigma, MBX_POLICY_RLEPSO);       // clip(N(mu, sigma), 0, 1)
                }
            }
        }
    }
}

}  // namespace mbx
