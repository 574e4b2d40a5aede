// Whiten / limit_param_value gradient shaping (reference model/layer/scaling.py:949-1095 Whiten +
// WhiteningPenaltyFunction, 1153-1190 limit_param_value) as a handful of launches:
//   forward (when the module fires):  x^T x and column sums come from s2t_linear_wgrad; one
//     kernel turns them into the covariance, its whitening metric and the host-visible flag;
//   backward: one kernel builds d metric / d cov and the bias row, a plain GEMM applies it to x,
//     a two-tensor sum-of-squares pass and a fused  g + pg * (grad_scale |g| / |pg|)  finish.
// The C x C work is tiny (C <= 512), so those kernels are single-workgroup on purpose.
#include "common.h"
#include <algorithm>
#include <cstdint>

namespace {

constexpr int NT = 1024;

__device__ __forceinline__ float block_sum(float v, float* s_tmp) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_tmp[wave] = v;
  __syncthreads();
  float t = (threadIdx.x < (NT >> 6)) ? s_tmp[threadIdx.x] : 0.f;
  if (wave == 0) {
    t = wave_sum(t);
    if (lane == 0) s_tmp[0] = t;
  }
  __syncthreads();
  return s_tmp[0];
}

// xtx: (C,C) = x^T x over all groups' channels (upper 64x64 tiles; a full matrix works too); cov: (G,cg,cg) per-group centred covariance
// scal: [md, covsq, denom, metric].  One workgroup per row ci: its group's block of the centred
// covariance, the row's partial sums, then the row of xtx is ZEROED (the accumulating TN GEMM that
// fills it finds it clean next time -- no fill launch per call).  The last workgroup (ticket) adds
// the partials up, writes scal / mean / the host-visible metric, zeroes colsum and the ticket.
// ws: [ticket (unsigned)][pad to 4][2 floats per row]
__global__ __launch_bounds__(256) void whiten_metric_kernel(float* __restrict__ xtx,
                                                            float* __restrict__ colsum, float n,
                                                            int G, int cg, float* __restrict__ cov,
                                                            float* __restrict__ mean,
                                                            float* __restrict__ scal,
                                                            float* host_metric,
                                                            float* __restrict__ ws) {
  __shared__ float s_red[2][4];
  __shared__ unsigned s_old;
  const int C = G * cg;
  const int ci = blockIdx.x;
  const int g = ci / cg, i = ci - g * cg;
  const float inv_n = 1.f / n;
  const float mi = colsum[ci] * inv_n;
  float dsum = 0.f, sq = 0.f;
  float* xrow = xtx + (long)ci * C;
  // xtx holds the 64x64 tiles on and above the diagonal (s2t_gemm_xtx): a row takes the pairs
  // whose column tile is not left of its own, and mirrors those of tiles strictly to the right
  // (the row that owns the mirror image has no valid copy of it)
  for (int j = threadIdx.x; j < cg; j += 256) {
    const int cj = g * cg + j;
    if ((cj >> 6) < (ci >> 6)) continue;
    const float v = xrow[cj] - mi * colsum[cj];
    cov[((long)g * cg + i) * cg + j] = v;
    sq = fmaf(v, v, sq);
    if (i == j) dsum = v;
    if ((cj >> 6) > (ci >> 6)) {
      cov[((long)g * cg + j) * cg + i] = v;
      sq = fmaf(v, v, sq);
    }
  }
  __syncthreads();                                   // row fully read before it is cleared
  for (int j = threadIdx.x; j < C; j += 256) xrow[j] = 0.f;
  dsum = wave_sum(dsum);
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) {
    s_red[0][threadIdx.x >> 6] = dsum;
    s_red[1][threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  unsigned* ticket = reinterpret_cast<unsigned*>(ws);
  float* part = ws + 4;
  if (threadIdx.x == 0) {
    part[2 * ci] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
    part[2 * ci + 1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    __threadfence();
    s_old = atomicAdd(ticket, 1u);
  }
  __syncthreads();
  if (s_old != gridDim.x - 1) return;
  __threadfence();
  dsum = 0.f;
  sq = 0.f;
  for (int r = threadIdx.x; r < C; r += 256) {
    dsum += __builtin_nontemporal_load(part + 2 * r);
    sq += __builtin_nontemporal_load(part + 2 * r + 1);
    mean[r] = colsum[r] * inv_n;
  }
  dsum = wave_sum(dsum);
  sq = wave_sum(sq);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    s_red[0][threadIdx.x >> 6] = dsum;
    s_red[1][threadIdx.x >> 6] = sq;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < C; r += 256) colsum[r] = 0.f;
  if (threadIdx.x == 0) {
    const float dtot = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
    const float sqtot = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    const float md = dtot / (float)C;
    const float covsq = sqtot / (float)C;
    const float denom = md * md + 1.0e-20f;
    const float metric = covsq / denom;
    scal[0] = md;
    scal[1] = covsq;
    scal[2] = denom;
    scal[3] = metric;
    *ticket = 0u;
    if (host_metric)
      __hip_atomic_store(host_metric, metric, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// dcov (C,C) block diagonal = 4/(G cg) (cov/denom - covsq md/denom^2 I); bias = -mean . dcov.
// One workgroup per row; dcov is symmetric (cov is), so bias[ci] is the row's dot with mean.
__global__ __launch_bounds__(256) void whiten_dcov_kernel(const float* __restrict__ cov,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ scal, int G,
                                                          int cg, float* __restrict__ dcov,
                                                          float* __restrict__ bias,
                                                          float* __restrict__ sums) {
  __shared__ float s_red[4];
  const int C = G * cg;
  const int ci = blockIdx.x;
  const float md = scal[0], covsq = scal[1], denom = scal[2];
  const float k = 4.f / (float)C;
  const float a = 1.f / denom, d = covsq * md / (denom * denom);
  const int g = ci / cg;
  const float* crow = cov + ((long)g * cg + (ci - g * cg)) * cg;
  float acc = 0.f;
  for (int cj = threadIdx.x; cj < C; cj += 256) {
    float v = 0.f;
    if (cj / cg == g) {
      v = crow[cj - g * cg] * a;
      if (ci == cj) v -= d;
      v *= k;
    }
    dcov[(long)ci * C + cj] = v;
    acc = fmaf(mean[cj], v, acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    bias[ci] = -((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    if (ci == 0) {
      sums[0] = 0.f;
      sums[1] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void sumsq2_kernel(const float* __restrict__ a,
                                                     const float* __restrict__ b, long n,
                                                     float* __restrict__ sums) {
  __shared__ float s_a[4], s_b[4];
  float sa = 0.f, sb = 0.f;
  const long n4 = n >> 2;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    const float4 x = a4[e], y = b4[e];
    sa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    sb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
  }
  if (blockIdx.x == 0)
    for (long e = (n4 << 2) + threadIdx.x; e < n; e += 256) {
      sa = fmaf(a[e], a[e], sa);
      sb = fmaf(b[e], b[e], sb);
    }
  sa = wave_sum(sa);
  sb = wave_sum(sb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_a[wave] = sa; s_b[wave] = sb; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], s_a[0] + s_a[1] + s_a[2] + s_a[3]);
    atomicAdd(&sums[1], s_b[0] + s_b[1] + s_b[2] + s_b[3]);
  }
}

// out = g + pg * grad_scale * |g| / (|pg| + 1e-20)
__global__ __launch_bounds__(256) void whiten_apply_kernel(const float* __restrict__ g,
                                                           const float* __restrict__ pg, long n,
                                                           float grad_scale,
                                                           const float* __restrict__ sums,
                                                           float* __restrict__ out) {
  const float scale = grad_scale * (sqrtf(sums[0]) / (sqrtf(sums[1]) + 1.0e-20f));
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* p4 = reinterpret_cast<const float4*>(pg);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    const float4 x = g4[e], y = p4[e];
    o4[e] = make_float4(fmaf(y.x, scale, x.x), fmaf(y.y, scale, x.y), fmaf(y.z, scale, x.z),
                        fmaf(y.w, scale, x.w));
  }
  if (blockIdx.x == 0)
    for (long e = (n4 << 2) + threadIdx.x; e < n; e += 256) out[e] = fmaf(pg[e], scale, g[e]);
}

// out = g + pg * grad_scale * |g| / (|pg| + 1e-20) with the two squared norms given as [2][64] partial
// sums (s2t_gemm_x3p_sq's slots): every workgroup adds them up for itself (128 floats)
__global__ __launch_bounds__(256) void whiten_apply64_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ pg, long n,
                                                             float grad_scale,
                                                             const float* __restrict__ sums64,
                                                             float* __restrict__ out) {
  __shared__ float s_sc;
  if (threadIdx.x < 64) {
    const float a = wave_sum(sums64[threadIdx.x]), b = wave_sum(sums64[64 + threadIdx.x]);
    if (threadIdx.x == 0) s_sc = grad_scale * (sqrtf(a) / (sqrtf(b) + 1.0e-20f));
  }
  __syncthreads();
  const float scale = s_sc;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* p4 = reinterpret_cast<const float4*>(pg);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
    const float4 x = g4[e], y = p4[e];
    o4[e] = make_float4(fmaf(y.x, scale, x.x), fmaf(y.y, scale, x.y), fmaf(y.z, scale, x.z),
                        fmaf(y.w, scale, x.w));
  }
  if (blockIdx.x == 0)
    for (long e = (n4 << 2) + threadIdx.x; e < n; e += 256) out[e] = fmaf(pg[e], scale, g[e]);
}

// limit_param_value backward: flip the sign of gradient entries that push an out-of-range
// parameter further out (first the lower bound, then the upper bound on the updated value)
__global__ __launch_bounds__(256) void limit_param_grad_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ g,
                                                               float lo, float hi, long n,
                                                               float* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float v = g[e];
  const float xv = x[e];
  if (v > 0.f && xv < lo) v = -v;
  if (v < 0.f && xv > hi) v = -v;
  out[e] = v;
}

}  // namespace

extern "C" int s2t_whiten_metric(float* xtx, float* colsum, long n, int G, int cg, float* cov,
                                 float* mean, float* scal, float* host_metric, float* workspace,
                                 void* stream) {
  if (G <= 0 || cg <= 0 || n <= 0 || !workspace) return -1;
  hipLaunchKernelGGL(whiten_metric_kernel, dim3(G * cg), dim3(256), 0, (hipStream_t)stream, xtx,
                     colsum, (float)n, G, cg, cov, mean, scal, host_metric, workspace);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_whiten_dcov(const float* cov, const float* mean, const float* scal, int G,
                               int cg, float* dcov, float* bias, float* sums, void* stream) {
  if (G <= 0 || cg <= 0) return -1;
  hipLaunchKernelGGL(whiten_dcov_kernel, dim3(G * cg), dim3(256), 0, (hipStream_t)stream, cov,
                     mean, scal, G, cg, dcov, bias, sums);
  S2T_CHECK_LAUNCH();
  return 0;
}

// s2t_whiten_dcov for the form whose backward is s2t_gemm_x3p_sq + s2t_whiten_combine64: dcov / bias
// as there, and sums64 = the [2][64] partial-sum slots of the two norms, zeroed.  Everything here
// depends on x only: the caller runs it in FORWARD, on the statistics' stream, followed by s2t_x3p_split
// of dcov -- backward's chain is then two launches on the data-gradient stream instead of three.
extern "C" int s2t_whiten_prep(const float* cov, const float* mean, const float* scal, int G, int cg,
                               float* dcov, float* bias, float* sums64, void* stream) {
  if (G <= 0 || cg <= 0 || !sums64) return -1;
  if (hipMemsetAsync(sums64, 0, 128 * sizeof(float), (hipStream_t)stream) != hipSuccess) return -3;
  hipLaunchKernelGGL(whiten_dcov_kernel, dim3(G * cg), dim3(256), 0, (hipStream_t)stream, cov, mean,
                     scal, G, cg, dcov, bias, sums64);
  S2T_CHECK_LAUNCH();
  return 0;
}

// sums64[slot] += partial sums of g^2 (64 slots: the same layout s2t_gemm_x3p_sq's epilogue adds into)
namespace {
__global__ __launch_bounds__(256) void sumsq64_kernel(const float* __restrict__ g, long numel,
                                                      float* __restrict__ sums64) {
  const long n4 = numel >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = g4[i];
    acc += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += stride) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) atomicAdd(sums64 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 63), acc);
}
}  // namespace

extern "C" int s2t_sumsq64(const float* g, long numel, float* sums64, void* stream) {
  if (numel <= 0) return 0;
  if (!g || !sums64 || (reinterpret_cast<uintptr_t>(g) & 15)) return -1;
  const long n4 = numel >> 2;
  const unsigned blocks = (unsigned)std::min<long>(1024, std::max<long>(1, (n4 + 255) / 256));
  hipLaunchKernelGGL(sumsq64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, numel, sums64);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_whiten_combine64(const float* g, const float* pg, long numel, float grad_scale,
                                    const float* sums64, float* out, void* stream) {
  if (numel <= 0) return 0;
  if (!sums64 || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(pg) |
                   reinterpret_cast<uintptr_t>(out)) & 15))
    return -1;
  const long n4 = numel >> 2;
  const unsigned blocks = (unsigned)std::min<long>(2048, std::max<long>(1, (n4 + 255) / 256));
  hipLaunchKernelGGL(whiten_apply64_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, pg,
                     numel, grad_scale, sums64, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_whiten_apply(const float* g, const float* pg, long numel, float grad_scale,
                                float* sums, float* out, void* stream) {
  if (numel <= 0) return 0;
  if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(pg) |
       reinterpret_cast<uintptr_t>(out)) & 15)
    return -1;
  const long n4 = numel >> 2;
  const unsigned blocks = (unsigned)std::min<long>(2048, std::max<long>(1, (n4 + 255) / 256));
  // every workgroup ends with two atomics on the SAME two words (serialised by the memory
  // system): keep that tail short -- 512 workgroups stream the two tensors just as fast
  const unsigned sblocks = std::min(blocks, 512u);
  hipLaunchKernelGGL(sumsq2_kernel, dim3(sblocks), dim3(256), 0, (hipStream_t)stream, g, pg, numel,
                     sums);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(whiten_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, pg,
                     numel, grad_scale, sums, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

// the update alone: sums = (||g||^2, ||pg||^2) were taken by the product that wrote pg (s2t_gemm_f32_sq)
extern "C" int s2t_whiten_combine(const float* g, const float* pg, long numel, float grad_scale,
                                  const float* sums, float* out, void* stream) {
  if (numel <= 0) return 0;
  if (!sums || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(pg) |
                 reinterpret_cast<uintptr_t>(out)) & 15))
    return -1;
  const long n4 = numel >> 2;
  const unsigned blocks = (unsigned)std::min<long>(2048, std::max<long>(1, (n4 + 255) / 256));
  hipLaunchKernelGGL(whiten_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, pg,
                     numel, grad_scale, sums, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_limit_param_grad(const float* x, const float* g, float lo, float hi, long n,
                                    float* out, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(limit_param_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x, g, lo, hi, n, out);
  S2T_CHECK_LAUNCH();
  return 0;
}
