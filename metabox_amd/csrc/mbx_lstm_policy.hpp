// mbx_lstm_policy.hpp — LDE's PolicyNet as ONE kernel launch per generation.
//
// Reference: src/agent/lde_agent.py:8-29 (PolicyNet: one LSTM cell NP+10 -> 50, Linear 50 -> 2 NP for mu, Linear 50 -> 2 NP + sigmoid for
// sigma) and :147-163 (rollout: action = clip(Normal(mu, sigma).sample(), 0, 1), (h, c) carried).  As PyTorch ops this is an LSTM call, two
// addmm, a sigmoid, a randn, a clip and their glue, ~0.18 ms of launches next to a 0.63 ms generation kernel (config 3).  The network is
// 32.6 k parameters and 32 kMAC per instance: far below what a library GEMM needs to amortise a launch.  Rounds 1-2 ran it as float32 VALU work
// (thread = output unit, 16 accumulators, 62-78 us per generation of 16 384 instances: 10.7 % of the float32 vector peak, bound by the weight-load
// latency of 1024 short workgroups); round 3 puts both GEMM-shaped parts on the float32 matrix cores, 16 x 16 x 4 tiles like mbx_qnet.hpp:
//   * a workgroup owns a tile of TI = 16 instances; their inputs [x | h] sit in LDS as XS[k][i] (k-major: the A fragment of a k-step is 64
//     consecutive-ish words, conflict-free);
//   * a wave owns 16 x 16 output tiles (16 instances x 16 gate rows, then 16 instances x 16 components of the mu AND the sigma head), chained over
//     ceil(K / 4) v_mfma_f32_16x16x4_f32 with the accumulator started at the bias; the B fragment is read from the TRANSPOSED weight matrix
//     Wt[k][u], 16 consecutive words per k (the 130 KB of weights are shared by every workgroup and stay in L2);
//   * gate nonlinearities, the cell update, both heads, the sigmoid and the Normal draw (Philox, counter (j, MBX_SITE_POLICY, gen + 1,
//     episode) like every other fused policy) happen in the same launch; (h, c) are updated in place.
// Arithmetic is float32 with the reference's association up to the order of the dot products (one fma chain per unit, k ascending, which is what
// the instruction computes bit for bit; torch's GEMM kernels sum in tiles): (mu, sigma, h', c') agree with the recorded reference I/O pairs to 5e-6 (tests/test_policy_io.py).
#pragma once
#include "mbx_rlepso.hpp"

namespace mbx {

struct LstmPolicy {
    const float* w;      // packed, see mbx_lstm_policy in include/mbx.h: WihT [IN][4H] | WhhT [H][4H] | b [4H] | WmuT [H][A] | WsgT [H][A] | bmu [A] | bsg [A]
    int32_t in_dim, hidden, out_dim;
};

constexpr int kLstmTile = 16;
typedef float lstm_f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int64_t lstm_policy_floats(int IN, int H, int A) { return (int64_t)(IN + H) * 4 * H + 4 * H + 2 * (int64_t)H * A + 2 * A; }
__host__ __device__ inline size_t lstm_policy_lds_bytes(int IN, int H, int TI = kLstmTile) { return sizeof(float) * (size_t)((IN + H) * TI + 4 * H * TI + H * TI); }
// (TI: instances per workgroup, a multiple of 16.  Measured on one box, 16 384 instances, pop 50 / 100: float32 VALU kernel 63.8 / 81.0 us, this kernel at
// TI = 16 50.9 / 79.5 us (33.5 / 50.8 us with the compile-time dimensions below), at TI = 64 -- four row tiles per B fragment, a quarter of the weight traffic, one workgroup per CU -- 78.6 / 125.1 us: the launch is
// bound by the latency chains of its workgroups, not by weight traffic or matrix work, so the small tile with six workgroups per CU stays.)

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// INC / HC / AC: the network's dimensions at compile time (0 = taken from `net`): with the k-step counts known, ALL B fragments of an output tile are
// loaded before its MFMA chain starts (one L2 round trip per tile instead of one per k-step, as in mbx_qnet.hpp); config 3's two networks get instantiations.
template <int TI, int INC = 0, int HC = 0, int AC = 0>
__global__ __launch_bounds__(kThreads) void k_lstm_policy(BatchParams bp, LstmPolicy net, const double* __restrict__ state, float* __restrict__ hbuf,
                                                          float* __restrict__ cbuf, float* __restrict__ actions, float* __restrict__ mu_sigma, int skip_done)
{
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    constexpr int RT = TI / 16;                                   // 16-instance row tiles per workgroup
    const int tid = threadIdx.x;
    const int IN = INC ? INC : net.in_dim, H = HC ? HC : net.hidden, A = AC ? AC : net.out_dim, G4 = 4 * H, K1 = IN + H;
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
    // ---- LSTM gates: unit u of [i | f | g | o] (torch.nn.LSTM's row order), all TI instances.  One 16 x 16 output tile (16 instances x 16 units) per
    // wave at a time on the float32 matrix cores: A fragment = XS[4 s + (l >> 4)][l & 15] (k-major in LDS: conflict-free), B fragment = the
    // transposed weight Wt[4 s + (l >> 4)][16 t + (l & 15)] (16 consecutive words per k) from L2, accumulator started at the bias.
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), cc = lane & 15, qq = lane >> 4;
    {
        const int tiles = (G4 + 15) >> 4, ks = (K1 + 3) >> 2;
        for (int t = wave; t < tiles; t += kThreads / 64) {
            const int u = 16 * t + cc;
            const float bias = u < G4 ? bg[u] : 0.f;
            lstm_f32x4 acc[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt] = lstm_f32x4{bias, bias, bias, bias};
            if constexpr (INC > 0 && HC > 0) {
                constexpr int KS = (INC + HC + 3) / 4;
                float wv[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int k = 4 * s + qq;
                    wv[s] = (k < INC + HC && u < G4) ? (k < INC ? WihT[k * (4 * HC) + u] : WhhT[(k - INC) * (4 * HC) + u]) : 0.f;
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int k = 4 * s + qq;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float a = k < INC + HC ? XS[k * TI + 16 * rt + cc] : 0.f;
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wv[s], acc[rt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll 2
                for (int s = 0; s < ks; ++s) {
                    const int k = 4 * s + qq;
                    float w = 0.f;
                    if (k < K1 && u < G4) w = k < IN ? WihT[(int64_t)k * G4 + u] : WhhT[(int64_t)(k - IN) * G4 + u];
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float a = k < K1 ? XS[k * TI + 16 * rt + cc] : 0.f;
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w, acc[rt], 0, 0, 0);
                    }
                }
            }
            if (u < G4) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) GT[u * TI + 16 * rt + 4 * qq + r] = acc[rt][r];   // C / D layout: row (instance) 4 (l >> 4) + r, column (unit) l & 15
            }
        }
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
            // skip_done (mbx_lde_rollout's per-generation route): an instance that has terminated keeps its (h, c) -- the agent stops calling its
            // policy once env.step returned done (lde_agent.py:147-163)
            if (!(skip_done && bp.state[(int64_t)(b0 + i) * bp.state_stride + bp.sc_off + MBX_SC_DONE] != 0.)) {
                cbuf[(int64_t)(b0 + i) * H + j] = cn;
                hbuf[(int64_t)(b0 + i) * H + j] = hn;
            }
            HN[j * TI + i] = hn;
        } else HN[j * TI + i] = 0.f;
    }
    __syncthreads();
    // ---- heads + sampling: the mu tile and the sigma tile of 16 instances x 16 action components in the same wave, so that the lane that holds
    // (instance 4 (l >> 4) + r, component 16 t + (l & 15)) of both draws the action
    {
        const int tiles = (A + 15) >> 4, ks = (H + 3) >> 2;
        for (int t = wave; t < tiles; t += kThreads / 64) {
            const int j = 16 * t + cc;
            const float b1 = j < A ? bmu[j] : 0.f, b2 = j < A ? bsg[j] : 0.f;
            lstm_f32x4 am[RT], as[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) { am[rt] = lstm_f32x4{b1, b1, b1, b1}; as[rt] = lstm_f32x4{b2, b2, b2, b2}; }
            if constexpr (HC > 0 && AC > 0) {
                constexpr int KS = (HC + 3) / 4;
                float wmv[KS], wsv[KS];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int k = 4 * s + qq;
                    const bool in = k < HC && j < AC;
                    wmv[s] = in ? WmuT[k * AC + j] : 0.f; wsv[s] = in ? WsgT[k * AC + j] : 0.f;
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int k = 4 * s + qq;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float a = k < HC ? HN[k * TI + 16 * rt + cc] : 0.f;
                        am[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wmv[s], am[rt], 0, 0, 0);
                        as[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wsv[s], as[rt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll 2
                for (int s = 0; s < ks; ++s) {
                    const int k = 4 * s + qq;
                    float wm = 0.f, ws = 0.f;
                    if (k < H && j < A) { wm = WmuT[(int64_t)k * A + j]; ws = WsgT[(int64_t)k * A + j]; }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float a = k < H ? HN[k * TI + 16 * rt + cc] : 0.f;
                        am[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, wm, am[rt], 0, 0, 0);
                        as[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ws, as[rt], 0, 0, 0);
                    }
                }
            }
            if (j < A) {
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * rt + 4 * qq + r;
                        if (i < nb) {
                            const int b = b0 + i;
                            const float mu = am[rt][r], sigma = sigmoidf_(as[rt][r]);
                            if (mu_sigma) { mu_sigma[((int64_t)b * 2) * A + j] = mu; mu_sigma[((int64_t)b * 2 + 1) * A + j] = sigma; }
                            const double* sc = bp.state + (int64_t)b * bp.state_stride + bp.sc_off;
                            if (actions && !(skip_done && sc[MBX_SC_DONE] != 0.)) {
                                const uint64_t seed = bp.seeds[b];
                                // the action drawn here drives generation gen + 1 of the current episode
                                const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)((int)sc[MBX_SC_GEN] + 1), (uint32_t)(int)sc[MBX_SC_EPISODE]};
                                actions[(int64_t)b * A + j] = sample_action(rng, j, mu, sigma, MBX_POLICY_RLEPSO);       // clip(N(mu, sigma), 0, 1)
                            }
                        }
                    }
                }
            }
        }
    }
}

}  // namespace mbx
