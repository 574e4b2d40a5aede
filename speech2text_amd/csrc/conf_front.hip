// First convolution of the conformer `Subsampling` (model/encoder/conformer.py:47-60,114-126:
// Conv2d(1, D, 3, stride 2) + ReLU on the (B, 1, T, 80) feature image) as direct stencil kernels
// for gfx950, channel-last output.
//
// The layer has ONE input channel: as an implicit GEMM it is a 9-deep contraction that a dense
// library runs far below the HBM roofline, and its output (B x D x T/2 x 39 fp32: 637 MB at the C2
// batch) is by far the largest tensor of the step.  Here
//   * forward: one pass, conv + bias + ReLU fused, written channel-last (the layout the second
//     convolution's implicit-GEMM kernels take natively);
//   * weight / bias gradient: ONE pass over the incoming gradient.  The ReLU mask is recomputed
//     from the 3x3 input patch (9 FMAs) instead of reading the saved activation (another 637 MB),
//     so ReLU backward, the bias reduction and the weight-gradient contraction are a single read
//     of the gradient; per-workgroup partial sums, then a fixed-order fold (no atomics).
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

struct C1Args {
  const float* x;      // (B, T, F)
  const float* w;      // (C, 9)   nn.Conv2d weight (C, 1, 3, 3)
  const float* bias;   // (C)
  int B, T, F, C, T1, F1;
  float* out;          // forward: (B, T1, F1, C)
  const float* d;      // backward: gradient w.r.t. out, same layout
  float* partial;      // backward: [nwg][10][C]   (9 taps + bias)
};

// thread = (channel quad c4, position lane); a workgroup walks positions p = (b, t1, f1)
__global__ __launch_bounds__(256) void conv1_relu_fwd_kernel(C1Args a) {
  const int cg = a.C >> 2, c4 = threadIdx.x % cg, pl = threadIdx.x / cg, npl = 256 / cg;
  if (pl >= npl) return;
  float4 w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
    w[k] = make_float4(a.w[(4 * c4) * 9 + k], a.w[(4 * c4 + 1) * 9 + k], a.w[(4 * c4 + 2) * 9 + k],
                       a.w[(4 * c4 + 3) * 9 + k]);
  const float4 bv = reinterpret_cast<const float4*>(a.bias)[c4];
  const long npos = (long)a.B * a.T1 * a.F1;
  for (long p = (long)blockIdx.x * npl + pl; p < npos; p += (long)gridDim.x * npl) {
    const int f1 = (int)(p % a.F1);
    const long q = p / a.F1;
    const int t1 = (int)(q % a.T1), b = (int)(q / a.T1);
    const float* xp = a.x + ((long)b * a.T + 2 * t1) * a.F + 2 * f1;
    float4 acc = bv;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float xv = xp[kh * a.F + kw];
        const float4 wk = w[kh * 3 + kw];
        acc.x = fmaf(wk.x, xv, acc.x); acc.y = fmaf(wk.y, xv, acc.y);
        acc.z = fmaf(wk.z, xv, acc.z); acc.w = fmaf(wk.w, xv, acc.w);
      }
    acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f);
    acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    reinterpret_cast<float4*>(a.out + p * a.C)[c4] = acc;
  }
}

__global__ __launch_bounds__(256) void conv1_relu_wgrad_kernel(C1Args a) {
  __shared__ float4 red[256];
  const int cg = a.C >> 2, c4 = threadIdx.x % cg, pl = threadIdx.x / cg, npl = 256 / cg;
  float4 w[9], acc[10];
#pragma unroll
  for (int k = 0; k < 9; ++k)
    w[k] = make_float4(a.w[(4 * c4) * 9 + k], a.w[(4 * c4 + 1) * 9 + k], a.w[(4 * c4 + 2) * 9 + k],
                       a.w[(4 * c4 + 3) * 9 + k]);
#pragma unroll
  for (int k = 0; k < 10; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 bv = reinterpret_cast<const float4*>(a.bias)[c4];
  const long npos = (long)a.B * a.T1 * a.F1;
  if (pl < npl) {
    for (long p = (long)blockIdx.x * npl + pl; p < npos; p += (long)gridDim.x * npl) {
      const int f1 = (int)(p % a.F1);
      const long q = p / a.F1;
      const int t1 = (int)(q % a.T1), b = (int)(q / a.T1);
      const float* xp = a.x + ((long)b * a.T + 2 * t1) * a.F + 2 * f1;
      float xv[9];
      float4 z = bv;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int k = kh * 3 + kw;
          xv[k] = xp[kh * a.F + kw];
          z.x = fmaf(w[k].x, xv[k], z.x); z.y = fmaf(w[k].y, xv[k], z.y);
          z.z = fmaf(w[k].z, xv[k], z.z); z.w = fmaf(w[k].w, xv[k], z.w);
        }
      float4 g = reinterpret_cast<const float4*>(a.d + p * a.C)[c4];
      g.x = z.x > 0.f ? g.x : 0.f; g.y = z.y > 0.f ? g.y : 0.f;
      g.z = z.z > 0.f ? g.z : 0.f; g.w = z.w > 0.f ? g.w : 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        acc[k].x = fmaf(g.x, xv[k], acc[k].x); acc[k].y = fmaf(g.y, xv[k], acc[k].y);
        acc[k].z = fmaf(g.z, xv[k], acc[k].z); acc[k].w = fmaf(g.w, xv[k], acc[k].w);
      }
      acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
    }
  }
  // the position lanes of a channel quad meet in LDS, one of the 10 sums at a time
  float* dst = a.partial + (long)blockIdx.x * 10 * a.C;
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    __syncthreads();
    red[threadIdx.x] = acc[k];
    __syncthreads();
    if (pl == 0) {
      float4 s = acc[k];
      for (int j = 1; j < npl; ++j) {
        const float4 t = red[j * cg + c4];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      reinterpret_cast<float4*>(dst + (long)k * a.C)[c4] = s;
    }
  }
}

// grid (ceil(C/64), 10): thread (channel, row group of 16) folds the workgroups' partials
__global__ __launch_bounds__(1024) void conv1_fold_kernel(const float* __restrict__ partial, int nwg,
                                                          int C, float* __restrict__ dw,
                                                          float* __restrict__ db) {
  __shared__ float red[16][64];
  const int t = threadIdx.x, chl = t & 63, grp = t >> 6, k = blockIdx.y;
  const int c = blockIdx.x * 64 + chl;
  float acc = 0.f;
  if (c < C) {
    const float* p = partial + (long)k * C + c;
#pragma unroll 4
    for (int b = grp; b < nwg; b += 16) acc += p[(long)b * 10 * C];
  }
  red[grp][chl] = acc;
  __syncthreads();
  if (grp == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][chl];
    if (k < 9) dw[(long)c * 9 + k] += s;
    else if (db) db[c] += s;
  }
}

constexpr int C1_WGS = 1024;

bool c1_ok(int B, int T, int F, int C) {
  return B > 0 && T >= 3 && F >= 3 && C >= 4 && (C & 3) == 0 && C <= 1024 && 256 % (C >> 2) == 0;
}

}  // namespace

extern "C" {

long s2t_conv1_relu_workspace_floats(int C) { return (long)C1_WGS * 10 * C; }

int s2t_conv1_relu_fwd(const float* x, const float* w, const float* bias, int B, int T, int F,
                       int C, float* out, void* stream) {
  if (!c1_ok(B, T, F, C) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15))
    return -2;
  C1Args a{x, w, bias, B, T, F, C, (T - 3) / 2 + 1, (F - 3) / 2 + 1, out, nullptr, nullptr};
  const long npos = (long)B * a.T1 * a.F1;
  const int npl = 256 / (C >> 2);
  long grid = (npos + npl - 1) / npl;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(conv1_relu_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_conv1_relu_wgrad(const float* x, const float* w, const float* bias, const float* d_out,
                         int B, int T, int F, int C, float* dw, float* db, float* workspace,
                         void* stream) {
  if (!c1_ok(B, T, F, C) || (reinterpret_cast<uintptr_t>(d_out) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return -2;
  C1Args a{x, w, bias, B, T, F, C, (T - 3) / 2 + 1, (F - 3) / 2 + 1, nullptr, d_out, workspace};
  const long npos = (long)B * a.T1 * a.F1;
  const int npl = 256 / (C >> 2);
  long grid = (npos + npl - 1) / npl;
  if (grid > C1_WGS) grid = C1_WGS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv1_relu_wgrad_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv1_fold_kernel, dim3((C + 63) / 64, 10), dim3(1024), 0, st, workspace,
                     (int)grid, C, dw, db);
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
