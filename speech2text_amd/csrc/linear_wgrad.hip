// Weight / bias gradient of the linear layers on the training path:
//   dW[o][i] = sum_n G[n][o] A[n][i]      db[o] = sum_n G[n][o]
// with n = frames x utterances (4k..32k rows) and o, i = a few hundred features: a "tall-skinny
// transposed" GEMM whose output has only a handful of tiles, so the row dimension is split
// across the whole chip and the partial sums meet in a second pass.  Both operands have the
// output index contiguous in memory, which is exactly the lane index of the fp32 MFMA operands
// (v_mfma_f32_32x32x2_f32: lane&31 = m or n, lane>>5 = k), so fragments are loaded straight
// from global memory as float2 (two 32x32 tiles per load) with no LDS staging.
// Reference: every nn.Linear / ScaledLinear of model/encoder/zipformer.py (e.g. 1573-1593,
// 1966-1992) under loss.backward() in task_factory/rnnt_task.py.
#include "common.h"
#include <cstdint>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradPlan {
  int to, ti;     // 64x64 output tiles along o and i
  int gy;         // row-slice groups (4 wave slices each)
  int rows;       // rows per wave slice (even)
};

__host__ WgradPlan make_plan(int R, int N, int M) {
  WgradPlan p;
  p.to = (N + 63) / 64;
  p.ti = (M + 63) / 64;
  const int tiles = p.to * p.ti;
  int gy = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU in total
  const int maxgy = (R + 4 * 32 - 1) / (4 * 32); // at least 32 rows per wave
  if (gy > maxgy) gy = maxgy;
  if (gy < 1) gy = 1;
  int rows = (R + 4 * gy - 1) / (4 * gy);
  rows = (rows + 1) & ~1;
  p.gy = gy;
  p.rows = rows;
  return p;
}

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// part: [gy][N][M] partial dW, bpart: [gy][N] partial db
template <bool BIAS>
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ G, long ldg,
                                                    const float* __restrict__ A, long lda, int R,
                                                    int N, int M, int ti_count, int rows,
                                                    float* __restrict__ part,
                                                    float* __restrict__ bpart) {
  __shared__ float s_red[4][64][65];
  __shared__ float s_b[4][64];
  const int tile = blockIdx.x, to = tile / ti_count, ti = tile % ti_count;
  const int o0 = to * 64, i0 = ti * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
  const int slice = blockIdx.y * 4 + wave;
  const int kb = min(slice * rows, R), ke = min(kb + rows, R);
  // columns of this lane (clamped for the address, zeroed by the flags)
  const int oc = o0 + 2 * lo, ic = i0 + 2 * lo;
  const bool ok_o0 = oc < N, ok_o1 = oc + 1 < N, ok_i0 = ic < M, ok_i1 = ic + 1 < M;
  const float* gp = G + min(oc, N - 2);
  const float* ap = A + min(ic, M - 2);
  f32x16 c00 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 c01 = c00, c10 = c00, c11 = c00;
  float bs0 = 0.f, bs1 = 0.f;
  int n = kb;
  constexpr int U = 8;
  for (; n + 2 * U <= ke; n += 2 * U) {
    float2 g[U], a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long row = n + 2 * u + hi;
      g[u] = *reinterpret_cast<const float2*>(gp + row * ldg);
      a[u] = *reinterpret_cast<const float2*>(ap + row * lda);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float gx = ok_o0 ? g[u].x : 0.f, gy = ok_o1 ? g[u].y : 0.f;
      const float ax = ok_i0 ? a[u].x : 0.f, ay = ok_i1 ? a[u].y : 0.f;
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(gx, ax, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(gx, ay, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(gy, ax, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(gy, ay, c11, 0, 0, 0);
      if (BIAS) { bs0 += gx; bs1 += gy; }
    }
  }
  for (; n < ke; n += 2) {
    const bool rok = n + hi < ke;
    const long row = min(n + hi, ke - 1);
    const float2 g = *reinterpret_cast<const float2*>(gp + row * ldg);
    const float2 a = *reinterpret_cast<const float2*>(ap + row * lda);
    const float gx = (rok && ok_o0) ? g.x : 0.f, gy = (rok && ok_o1) ? g.y : 0.f;
    const float ax = ok_i0 ? a.x : 0.f, ay = ok_i1 ? a.y : 0.f;
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(gx, ax, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(gx, ay, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(gy, ax, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(gy, ay, c11, 0, 0, 0);
    if (BIAS) { bs0 += gx; bs1 += gy; }
  }
  // the 4 wave slices meet in LDS; tile element (m, n) of c[c][c'] is dW[o0+2m+c][i0+2n+c']
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = acc_row(r, hi);
    s_red[wave][2 * m][2 * lo] = c00[r];
    s_red[wave][2 * m][2 * lo + 1] = c01[r];
    s_red[wave][2 * m + 1][2 * lo] = c10[r];
    s_red[wave][2 * m + 1][2 * lo + 1] = c11[r];
  }
  if (BIAS && ti == 0) {
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (hi == 0) {
      s_b[wave][2 * lo] = bs0;
      s_b[wave][2 * lo + 1] = bs1;
    }
  }
  __syncthreads();
  float* pb = part + (long)blockIdx.y * N * M;
  for (int idx = tid; idx < 64 * 64; idx += 256) {
    const int oo = idx >> 6, ii = idx & 63;
    if (o0 + oo < N && i0 + ii < M)
      pb[(long)(o0 + oo) * M + i0 + ii] =
          s_red[0][oo][ii] + s_red[1][oo][ii] + s_red[2][oo][ii] + s_red[3][oo][ii];
  }
  if (BIAS && ti == 0 && tid < 64 && o0 + tid < N)
    bpart[(long)blockIdx.y * N + o0 + tid] = s_b[0][tid] + s_b[1][tid] + s_b[2][tid] + s_b[3][tid];
}

// dW[e] (+)= sum_s part[s][e]  and  db[e - nw] (+)= sum_s bpart[s][e - nw]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, long nw,
                                                           const float* __restrict__ bpart, int nb,
                                                           int S, int accumulate,
                                                           float* __restrict__ dW,
                                                           float* __restrict__ db) {
  long e = (long)blockIdx.x * 256 + threadIdx.x;
  const float* src = part;
  float* dst = dW;
  long n = nw;
  if (e >= nw) {
    e -= nw;
    if (e >= nb) return;
    src = bpart; dst = db; n = nb;
  }
  float acc = accumulate ? dst[e] : 0.f;
  for (int s = 0; s < S; ++s) acc += src[(long)s * n + e];
  dst[e] = acc;
}

}  // namespace

extern "C" long s2t_linear_wgrad_workspace_floats(int R, int N, int M) {
  if (R <= 0 || N <= 0 || M <= 0) return 0;
  const WgradPlan p = make_plan(R, N, M);
  return (long)p.gy * ((long)N * M + N);
}

extern "C" int s2t_linear_wgrad(const float* g, long ldg, const float* a, long lda, int R, int N,
                                int M, float* dW, float* db, int accumulate, float* workspace,
                                void* stream) {
  if (N <= 0 || M <= 0) return 0;
  if (R <= 0) return -1;
  // float2 fragment loads: even feature counts and row strides, 8-byte aligned bases
  if ((N & 1) || (M & 1) || (ldg & 1) || (lda & 1) || N < 2 || M < 2) return -1;
  if ((reinterpret_cast<uintptr_t>(g) & 7) || (reinterpret_cast<uintptr_t>(a) & 7)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const WgradPlan p = make_plan(R, N, M);
  float* part = workspace;
  float* bpart = workspace + (long)p.gy * N * M;
  const dim3 grid(p.to * p.ti, p.gy);
  if (db)
    hipLaunchKernelGGL(wgrad_kernel<true>, grid, dim3(256), 0, st, g, ldg, a, lda, R, N, M, p.ti,
                       p.rows, part, bpart);
  else
    hipLaunchKernelGGL(wgrad_kernel<false>, grid, dim3(256), 0, st, g, ldg, a, lda, R, N, M, p.ti,
                       p.rows, part, bpart);
  S2T_CHECK_LAUNCH();
  const long nw = (long)N * M;
  const int nb = db ? N : 0;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nw + nb + 255) / 256)), dim3(256), 0, st,
                     part, nw, bpart, nb, p.gy, accumulate, dW, db);
  S2T_CHECK_LAUNCH();
  return 0;
}
