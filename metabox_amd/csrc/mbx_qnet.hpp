// mbx_qnet.hpp — DE-DDQN's Q-network (greedy action) for the whole batch as ONE launch on the float32 matrix cores.
//
// Reference: src/agent/de_ddqn_agent.py:59-68, 108-117 -- `action = argmax(Q(state))` with Q = MLP 99 -> 100 -> 100 -> 100 -> 100 -> 4, ReLU
// (src/agent/de_ddqn_agent.py:26-36, networks.py:4-26).  As PyTorch ops this is five hipBLASLt GEMMs at 4.8 % MFMA utilisation plus ~10
// element-wise launches (cast, bias, ReLU, argmax): 56 us of a 174 us step for 2240 instances (round 2).  M = 2240 rows and K = N = 100 cannot
// fill a library tile, and a VALU kernel (round 1) was slower than the library; what the shape does fit is the 16 x 16 x 4 float32 MFMA:
//   * a workgroup (8 waves) owns a tile of 16 instances; their activations sit in LDS as ACT[row][k] (float32, row stride padded to 132 words);
//   * a layer K -> N is ceil(N / 16) column tiles x ceil(K / 4) chained v_mfma_f32_16x16x4_f32; wave w takes the column tiles w, w + 8, .. (one per wave at width 100);
//     A fragment (lane l: row l & 15, k = 4 s + (l >> 4)) is one LDS word, B fragment (k = 4 s + (l >> 4), column l & 15) one word of the
//     TRANSPOSED weight matrix Wt[k][n] -- 16 consecutive words per k, four 64-byte segments per load -- straight from L2 (the 162 KB of
//     weights are shared by all workgroups); the accumulator starts at the bias, so a layer's output is ONE float32 fma chain per unit,
//     k ascending (the instruction is bit-identical to fmaf chains, guides/cdna_hip_programming.md);
//   * the B fragments of the NEXT layer are loaded before the barrier that ends the current one: weights do not depend on activations, so their
//     L2 latency hides behind the current layer's matrix work;
//   * ReLU on the way back to LDS; the last layer's 4 outputs per row are reduced to torch.argmax's answer (first maximum) by the row's lane.
// Only the reference's architecture is instantiated (in 99, width 100, depth 4, 4 actions); other shapes keep the PyTorch route.
#pragma once
#include "mbx_rlepso.hpp"

namespace mbx {

struct QNet {
    const float* w;              // per layer: Wt [in][out] (row-major, i.e. the torch weight transposed) followed by b [out]
    int32_t in_dim, width, depth, n_act;
};

__host__ __device__ inline int64_t qnet_floats(int in, int width, int depth, int n_act)
{
    return (int64_t)in * width + width + (int64_t)(depth - 1) * ((int64_t)width * width + width) + (int64_t)width * n_act + n_act;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kQTile = 16;       // instances per workgroup (one MFMA row tile)
constexpr int kQStride = 132;    // words per activation row in LDS (>= 128, + 4 so that the 16 rows of an A fragment fall into different banks)
#ifndef MBX_QNET_THREADS
#define MBX_QNET_THREADS 512
#endif
constexpr int kQThreads = MBX_QNET_THREADS;   // 8 waves: one 16-column tile of a 100-wide layer per wave (7 tiles), i.e. ONE chain of 25 MFMAs per wave and layer
constexpr int kQWaves = kQThreads / 64;

// B fragments of one layer for this wave: column tiles wave, wave + kQWaves (at most TPW), K / 4 steps each
template <int K, int N, int TPW>
__device__ __forceinline__ void qnet_load_b(const float* __restrict__ Wt, int wave, int lane, float (&bf)[TPW][(K + 3) / 4])
{
    constexpr int KS = (K + 3) / 4;
    const int c = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int n = 16 * (wave + kQWaves * j) + c;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int k = 4 * s + q;
            bf[j][s] = (n < N && k < K) ? Wt[k * N + n] : 0.f;
        }
    }
}

// one layer: OUT[row][n] = act(b[n] + sum_k IN[row][k] Wt[k][n]) for this wave's column tiles
template <int K, int N, int TPW, bool RELU>
__device__ __forceinline__ void qnet_layer(const float* IN, float* OUT, const float* __restrict__ bias, int wave, int lane,
                                           const float (&bf)[TPW][(K + 3) / 4])
{
    constexpr int KS = (K + 3) / 4;
    const int c = lane & 15, q = lane >> 4;
    float a[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) a[s] = IN[c * kQStride + 4 * s + q];          // columns K .. 4 KS - 1 of IN are zero
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int t = wave + kQWaves * j;
        if (16 * t >= N) continue;                                            // wave-uniform
        const int n = 16 * t + c;
        const float b0 = n < N ? bias[n] : 0.f;
        f32x4 acc = {b0, b0, b0, b0};
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bf[j][s], acc, 0, 0, 0);
        if (n < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[r];
                OUT[(4 * q + r) * kQStride + n] = RELU ? fmaxf(v, 0.f) : v;    // C / D layout of the 16 x 16 shapes: row 4 (l >> 4) + r, column l & 15
            }
        }
    }
}

template <int IN, int W, int A>
__global__ __launch_bounds__(kQThreads) void k_qnet_argmax(QNet net, const double* __restrict__ state, int32_t* __restrict__ actions,
                                                          float* __restrict__ q_out, int B)
{
    static_assert(IN <= 128 && W <= 128 && A <= 16, "k_qnet_argmax: one LDS row of 128 words per instance");
    __shared__ __attribute__((aligned(16))) float act0[kQTile * kQStride];
    __shared__ __attribute__((aligned(16))) float act1[kQTile * kQStride];
    constexpr int TPW = ((W + 15) / 16 + kQWaves - 1) / kQWaves;                // column tiles per wave in the hidden layers
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * kQTile;
    const int nb = B - b0 < kQTile ? B - b0 : kQTile;
    const float* W1 = net.w;                          const float* B1 = W1 + IN * W;
    const float* W2 = B1 + W;                         const float* B2 = W2 + W * W;
    const float* W3 = B2 + W;                         const float* B3 = W3 + W * W;
    const float* W4 = B3 + W;                         const float* B4 = W4 + W * W;
    const float* W5 = B4 + W;                         const float* B5 = W5 + W * A;
    // the first layer's weights are on their way while the states are staged
    float bfa[TPW][(IN + 3) / 4];
    qnet_load_b<IN, W, TPW>(W1, wave, lane, bfa);
    for (int t = tid; t < kQTile * kQStride; t += kQThreads) {
        const int i = t / kQStride, k = t - i * kQStride;
        act0[t] = (i < nb && k < IN) ? (float)state[(int64_t)(b0 + i) * IN + k] : 0.f;
        act1[t] = 0.f;                                                        // padding columns of both buffers stay zero: only n < N is ever written
    }
    float bfb[TPW][(W + 3) / 4];
    qnet_load_b<W, W, TPW>(W2, wave, lane, bfb);
    __syncthreads();
    qnet_layer<IN, W, TPW, true>(act0, act1, B1, wave, lane, bfa);
    float bfc[TPW][(W + 3) / 4];
    qnet_load_b<W, W, TPW>(W3, wave, lane, bfc);
    __syncthreads();
    qnet_layer<W, W, TPW, true>(act1, act0, B2, wave, lane, bfb);
    qnet_load_b<W, W, TPW>(W4, wave, lane, bfb);
    __syncthreads();
    qnet_layer<W, W, TPW, true>(act0, act1, B3, wave, lane, bfc);
    float bfl[1][(W + 3) / 4];
    qnet_load_b<W, A, 1>(W5, wave, lane, bfl);                                 // (waves 1 .. 7 load zeros: their column tile is empty)
    __syncthreads();
    qnet_layer<W, W, TPW, true>(act1, act0, B4, wave, lane, bfb);
    __syncthreads();
    qnet_layer<W, A, 1, false>(act0, act1, B5, wave, lane, bfl);               // Q values -> act1[row][0 .. A)
    __syncthreads();
    if (tid < nb) {
        const float* qv = act1 + tid * kQStride;
        int best = 0; float bv = qv[0];
#pragma unroll
        for (int j = 1; j < A; ++j) { const float v = qv[j]; if (v > bv) { bv = v; best = j; } }      // first maximum, like torch.argmax
        actions[b0 + tid] = best;
        if (q_out) for (int j = 0; j < A; ++j) q_out[(int64_t)(b0 + tid) * A + j] = qv[j];
    }
}

}  // namespace mbx
