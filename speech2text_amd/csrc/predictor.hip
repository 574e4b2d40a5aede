// StatelessPredictor's embedding + depthwise context convolution as one gather kernel per pass
// (reference model/predictor/stateless_predictor.py:27-105: Embedding -> Conv1d(D, D, context,
// groups=D, bias=False) on the blank-left-padded label sequence).
//
//   out[b,u,d] = sum_k w[d,k] * E[tok[b,u+k], d]            u < L - K + 1
//
// The tensors are tiny (B x 55 x 512 at C3) but the library path spent 325 us per step on them: a
// generic direct convolution forward, and for the two gradients a naive kernel plus a Winograd
// kernel chosen for a 3x3 image problem, next to the sort-based embedding gradient (~15 launches).
// Here: forward = one gather-multiply pass; backward = a scatter of the embedding gradient
// (fp32 atomics into its zeroed rows) and a split reduction of the K taps per channel.
#include "common.h"
#include <algorithm>

namespace {

constexpr int PRED_MAXK = 8;

__device__ __forceinline__ int clamp_tok(int t, int V) { return t < 0 ? 0 : (t >= V ? V - 1 : t); }

__global__ __launch_bounds__(256) void pred_ctx_fwd_kernel(const int* __restrict__ tok,
                                                           const float* __restrict__ E,
                                                           const float* __restrict__ w, int L,
                                                           int K, int D, int V, int Lo,
                                                           float* __restrict__ out) {
  const int b = blockIdx.x / Lo, u = blockIdx.x % Lo;
  __shared__ int s_tok[PRED_MAXK];
  if (threadIdx.x < K) s_tok[threadIdx.x] = clamp_tok(tok[(long)b * L + u + threadIdx.x], V);
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(w[d * K + k], E[(long)s_tok[k] * D + d], acc);
    out[(long)blockIdx.x * D + d] = acc;
  }
}

// dE[tok[b,j], d] += sum_k [0 <= j-k < Lo] w[d,k] * g[b,j-k,d]: one workgroup per position (b,j)
__global__ __launch_bounds__(256) void pred_ctx_bwd_emb_kernel(const int* __restrict__ tok,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ g, int L,
                                                               int K, int D, int V, int Lo,
                                                               float* __restrict__ dE) {
  const int b = blockIdx.x / L, j = blockIdx.x % L;
  const int t = clamp_tok(tok[blockIdx.x], V);
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const int u = j - k;
      if (u >= 0 && u < Lo) acc = fmaf(w[d * K + k], g[((long)b * Lo + u) * D + d], acc);
    }
    if (acc != 0.f) atomicAdd(&dE[(long)t * D + d], acc);
  }
}

// dw[d,k] += sum_{b,u} g[b,u,d] * E[tok[b,u+k], d]: grid (channel tiles of 64, row splits);
// thread = (channel, row group of 4); the K taps of a channel stay in registers
__global__ __launch_bounds__(256) void pred_ctx_bwd_w_kernel(const int* __restrict__ tok,
                                                             const float* __restrict__ E,
                                                             const float* __restrict__ g, int B,
                                                             int L, int K, int D, int V, int Lo,
                                                             float* __restrict__ dw) {
  __shared__ float s_red[4][PRED_MAXK][64];
  const int d = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const long rows = (long)B * Lo;
  float acc[PRED_MAXK];
#pragma unroll
  for (int k = 0; k < PRED_MAXK; ++k) acc[k] = 0.f;
  if (d < D) {
    for (long r = (long)blockIdx.y * 4 + rg; r < rows; r += (long)gridDim.y * 4) {
      const int b = (int)(r / Lo), u = (int)(r % Lo);
      const float gv = g[r * D + d];
#pragma unroll
      for (int k = 0; k < PRED_MAXK; ++k)
        if (k < K) acc[k] = fmaf(gv, E[(long)clamp_tok(tok[(long)b * L + u + k], V) * D + d], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < PRED_MAXK; ++k) s_red[rg][k][threadIdx.x & 63] = acc[k];
  __syncthreads();
  if (rg == 0 && d < D) {
    for (int k = 0; k < K; ++k) {
      const float v = (s_red[0][k][threadIdx.x] + s_red[1][k][threadIdx.x]) +
                      (s_red[2][k][threadIdx.x] + s_red[3][k][threadIdx.x]);
      if (v != 0.f) atomicAdd(&dw[d * K + k], v);
    }
  }
}

}  // namespace

extern "C" int s2t_predictor_ctx_fwd(const int* tokens, const float* emb, const float* w, int B,
                                     int L, int K, int D, int V, float* out, void* stream) {
  if (B <= 0 || D <= 0) return 0;
  if (K < 1 || K > PRED_MAXK || L < K || V <= 0) return -1;
  const int Lo = L - K + 1;
  hipLaunchKernelGGL(pred_ctx_fwd_kernel, dim3((unsigned)(B * Lo)), dim3(256), 0, (hipStream_t)stream,
                     tokens, emb, w, L, K, D, V, Lo, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_predictor_ctx_bwd(const int* tokens, const float* emb, const float* w,
                                     const float* g, int B, int L, int K, int D, int V,
                                     float* d_emb, float* d_w, void* stream) {
  if (B <= 0 || D <= 0) return 0;
  if (K < 1 || K > PRED_MAXK || L < K || V <= 0) return -1;
  const int Lo = L - K + 1;
  hipStream_t st = (hipStream_t)stream;
  if (d_emb) {
    hipLaunchKernelGGL(pred_ctx_bwd_emb_kernel, dim3((unsigned)(B * L)), dim3(256), 0, st, tokens, w,
                       g, L, K, D, V, Lo, d_emb);
    S2T_CHECK_LAUNCH();
  }
  if (d_w) {
    const long rows = (long)B * Lo;
    const unsigned splits = (unsigned)std::min<long>(64, std::max<long>(1, rows / 32));
    hipLaunchKernelGGL(pred_ctx_bwd_w_kernel, dim3((unsigned)((D + 63) / 64), splits), dim3(256), 0,
                       st, tokens, emb, g, B, L, K, D, V, Lo, d_w);
    S2T_CHECK_LAUNCH();
  }
  return 0;
}
