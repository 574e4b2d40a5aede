// Row / channel normalisations and the SiLU passes of the conformer block, for gfx950.
//
// Replaces, per torchaudio.models.Conformer layer (call site model/encoder/conformer.py:170-178,
// 193; block structure: FFN(0.5) -> MHSA -> conv module -> FFN(0.5) -> LayerNorm):
//   * nn.LayerNorm forward / backward with the residual adds fused in: the forward optionally
//     forms x = x0 + alpha * y first (the "0.5 * ffn(x) + x" and "x + module(x)" sums) and writes
//     both the sum and its normalised copy; the backward adds the residual branch's gradient to
//     the normalisation's input gradient and accumulates d gamma / d beta straight into the flat
//     gradient buffer (one pass, per-workgroup partial sums, then one atomic per channel);
//   * nn.SiLU forward / backward on the (rows, ffn_dim) hidden tensor;
//   * nn.BatchNorm1d (training statistics over all B*T frames, running-stat update) + nn.SiLU of
//     the conv module as statistics pass + one fused normalise-activate pass each way.  The
//     per-channel sums are per-workgroup partials reduced in a fixed order (no atomics): replicas
//     of a data-parallel job see bit-identical statistics.
// All tensors are row-major (rows, C) fp32, C % 4 == 0, C <= 1024, 16-byte aligned.
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

constexpr int MAXV = 4;   // float4 per lane per row: C <= 64 * 4 * MAXV = 1024

__device__ __forceinline__ float sigmoid_f(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float silu_deriv(float x) {
  const float s = sigmoid_f(x);
  return s * (1.f + x * (1.f - s));
}

// ------------------------------------------------------------------ LayerNorm
// One wave per row (4 rows per workgroup); a lane owns the channels 4 (lane + 64 v) .. + 3.
template <bool ADD>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ y, float alpha,
    const float* __restrict__ gamma, const float* __restrict__ beta, long rows, int C, float eps,
    float* __restrict__ xsum, float* __restrict__ out, float* __restrict__ stats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const int nv = C >> 2;
  float4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < nv) {
      v[i] = reinterpret_cast<const float4*>(x + row * C)[c4];
      if (ADD) {
        const float4 t = reinterpret_cast<const float4*>(y + row * C)[c4];
        v[i].x += alpha * t.x; v[i].y += alpha * t.y; v[i].z += alpha * t.z; v[i].w += alpha * t.w;
        reinterpret_cast<float4*>(xsum + row * C)[c4] = v[i];
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (lane + 64 * i < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    if (c4 < nv) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
      const float4 b = reinterpret_cast<const float4*>(beta)[c4];
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x + b.x;
      o.y = (v[i].y - mean) * rstd * g.y + b.y;
      o.z = (v[i].z - mean) * rstd * g.z + b.z;
      o.w = (v[i].w - mean) * rstd * g.w + b.w;
      reinterpret_cast<float4*>(out + row * C)[c4] = o;
    }
  }
}

// dx = rstd (gamma dy - mean_C(gamma dy) - xhat mean_C(gamma dy xhat)) [+ resid];
// dgamma += sum_rows dy xhat, dbeta += sum_rows dy.  A wave walks rows wave_id, wave_id + W, ...
// and keeps its lanes' channel sums in registers; one LDS reduction + one atomic per channel
// and workgroup at the end (<= 256 workgroups: one per CU).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ dy, const float* __restrict__ resid, long rows, int C,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[3][2 * 256 * MAXV];   // waves 1..3 park their channel sums here
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  float4 g[MAXV], ag[MAXV], ab[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c4 = lane + 64 * i;
    g[i] = c4 < nv ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const long stride = (long)gridDim.x * 4;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += stride) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float4 xh[MAXV], d[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        const float4 xv = reinterpret_cast<const float4*>(x + row * C)[c4];
        d[i] = reinterpret_cast<const float4*>(dy + row * C)[c4];
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                            (xv.w - mean) * rstd);
        ag[i].x += d[i].x * xh[i].x; ag[i].y += d[i].y * xh[i].y;
        ag[i].z += d[i].z * xh[i].z; ag[i].w += d[i].w * xh[i].w;
        ab[i].x += d[i].x; ab[i].y += d[i].y; ab[i].z += d[i].z; ab[i].w += d[i].w;
        d[i].x *= g[i].x; d[i].y *= g[i].y; d[i].z *= g[i].z; d[i].w *= g[i].w;
        s1 += (d[i].x + d[i].y) + (d[i].z + d[i].w);
        s2 += (d[i].x * xh[i].x + d[i].y * xh[i].y) + (d[i].z * xh[i].z + d[i].w * xh[i].w);
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        float4 o;
        o.x = rstd * (d[i].x - s1 - xh[i].x * s2);
        o.y = rstd * (d[i].y - s1 - xh[i].y * s2);
        o.z = rstd * (d[i].z - s1 - xh[i].z * s2);
        o.w = rstd * (d[i].w - s1 - xh[i].w * s2);
        if (resid) {
          const float4 r = reinterpret_cast<const float4*>(resid + row * C)[c4];
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        reinterpret_cast<float4*>(dx + row * C)[c4] = o;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      float* p = red[wave - 1] + 8 * (lane + 64 * i);
      *reinterpret_cast<float4*>(p) = ag[i];
      *reinterpret_cast<float4*>(p + 4) = ab[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 >= nv) continue;
      float4 a = ag[i], b = ab[i];
      for (int w = 0; w < 3; ++w) {
        const float* p = red[w] + 8 * c4;
        const float4 a2 = *reinterpret_cast<const float4*>(p);
        const float4 b2 = *reinterpret_cast<const float4*>(p + 4);
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      }
      atomicAdd(dgamma + 4 * c4, a.x); atomicAdd(dgamma + 4 * c4 + 1, a.y);
      atomicAdd(dgamma + 4 * c4 + 2, a.z); atomicAdd(dgamma + 4 * c4 + 3, a.w);
      atomicAdd(dbeta + 4 * c4, b.x); atomicAdd(dbeta + 4 * c4 + 1, b.y);
      atomicAdd(dbeta + 4 * c4 + 2, b.z); atomicAdd(dbeta + 4 * c4 + 3, b.w);
    }
  }
}

// ------------------------------------------------------------------ SiLU
__global__ __launch_bounds__(256) void silu_fwd_kernel(const float4* __restrict__ h, long n4,
                                                       float4* __restrict__ a) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = h[i];
    a[i] = make_float4(silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w));
  }
}
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float4* __restrict__ h,
                                                       const float4* __restrict__ da, long n4,
                                                       float scale, float4* __restrict__ dh) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = h[i], g = da[i];
    dh[i] = make_float4(scale * g.x * silu_deriv(v.x), scale * g.y * silu_deriv(v.y),
                        scale * g.z * silu_deriv(v.z), scale * g.w * silu_deriv(v.w));
  }
}

// ------------------------------------------------------------------ BatchNorm1d + SiLU
// Statistics: workgroup b sums its slab of rows per channel -> partial[b][2][C] (plain stores).
// kind 0: (sum x, sum x^2);  kind 1: (sum dz, sum dz xhat) with dz = ds * silu'(gamma xhat + beta).
template <int KIND>
__global__ __launch_bounds__(256) void bn_stats_kernel(
    const float* __restrict__ x, const float* __restrict__ ds, const float* __restrict__ mean,
    const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, long rows, int C, long rows_per_wg,
    float* __restrict__ partial) {
  __shared__ float4 sa[256], sb[256];
  const int cg = C >> 2;                 // float4 column groups
  const int rl = 256 / cg;               // row lanes (>= 1 since C <= 1024)
  const int t = threadIdx.x;
  const int c4 = t % cg, r0 = t / cg;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (r0 < rl) {
    float4 mu, rs, ga, be;
    if (KIND == 1) {
      mu = reinterpret_cast<const float4*>(mean)[c4];
      rs = reinterpret_cast<const float4*>(rstd)[c4];
      ga = reinterpret_cast<const float4*>(gamma)[c4];
      be = reinterpret_cast<const float4*>(beta)[c4];
    }
    const long lo = (long)blockIdx.x * rows_per_wg;
    const long hi = min(rows, lo + rows_per_wg);
    for (long r = lo + r0; r < hi; r += rl) {
      const float4 v = reinterpret_cast<const float4*>(x + r * C)[c4];
      if (KIND == 0) {
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        b.x += v.x * v.x; b.y += v.y * v.y; b.z += v.z * v.z; b.w += v.w * v.w;
      } else {
        const float4 g = reinterpret_cast<const float4*>(ds + r * C)[c4];
        const float hx = (v.x - mu.x) * rs.x, hy = (v.y - mu.y) * rs.y, hz = (v.z - mu.z) * rs.z,
                    hw = (v.w - mu.w) * rs.w;
        const float dx = g.x * silu_deriv(ga.x * hx + be.x), dy = g.y * silu_deriv(ga.y * hy + be.y),
                    dz = g.z * silu_deriv(ga.z * hz + be.z), dw = g.w * silu_deriv(ga.w * hw + be.w);
        a.x += dx; a.y += dy; a.z += dz; a.w += dw;
        b.x += dx * hx; b.y += dy * hy; b.z += dz * hz; b.w += dw * hw;
      }
    }
  }
  sa[t] = a;
  sb[t] = b;
  __syncthreads();
  if (t < cg) {
    for (int j = 1; j < rl; ++j) {
      const float4 a2 = sa[t + j * cg], b2 = sb[t + j * cg];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
    }
    float* p = partial + (long)blockIdx.x * 2 * C;
    reinterpret_cast<float4*>(p)[t] = a;
    reinterpret_cast<float4*>(p + C)[t] = b;
  }
}

// every workgroup folds the NB partials of all channels (fixed order, double) into LDS
__device__ __forceinline__ void bn_fold(const float* __restrict__ partial, int nb, int C,
                                        double* __restrict__ s0, double* __restrict__ s1) {
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    double acc = 0.0;
    for (int b = 0; b < nb; ++b) acc += (double)partial[(long)b * 2 * C + c];
    if (c < C) s0[c] = acc; else s1[c - C] = acc;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void bn_silu_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ partial, int nb,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var,
    long* __restrict__ num_batches, long rows, int C, float* __restrict__ y,
    float* __restrict__ save_mean, float* __restrict__ save_rstd) {
  __shared__ double s0[1024], s1[1024];
  __shared__ float smu[1024], srs[1024];
  bn_fold(partial, nb, C, s0, s1);
  for (int c = threadIdx.x; c < C; c += 256) {
    const double m = s0[c] / (double)rows;
    double var = s1[c] / (double)rows - m * m;           // biased (normalisation)
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    smu[c] = (float)m;
    srs[c] = rs;
    if (blockIdx.x == 0) {
      save_mean[c] = (float)m;
      save_rstd[c] = rs;
      if (running_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  __syncthreads();
  const long n4 = rows * (long)(C >> 2);
  const int cg = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c = 4 * (int)(i % cg);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = silu_f((v.x - smu[c]) * srs[c] * gamma[c] + beta[c]);
    o.y = silu_f((v.y - smu[c + 1]) * srs[c + 1] * gamma[c + 1] + beta[c + 1]);
    o.z = silu_f((v.z - smu[c + 2]) * srs[c + 2] * gamma[c + 2] + beta[c + 2]);
    o.w = silu_f((v.w - smu[c + 3]) * srs[c + 3] * gamma[c + 3] + beta[c + 3]);
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// evaluation mode: y = silu((x - mean) * rstd * gamma + beta) with the running statistics
__global__ __launch_bounds__(256) void bn_silu_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, long rows, int C,
    float* __restrict__ y) {
  const long n4 = rows * (long)(C >> 2);
  const int cg = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % cg);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[c4];
    const float4 rs = reinterpret_cast<const float4*>(rstd)[c4];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 be = reinterpret_cast<const float4*>(beta)[c4];
    reinterpret_cast<float4*>(y)[i] =
        make_float4(silu_f((v.x - mu.x) * rs.x * ga.x + be.x), silu_f((v.y - mu.y) * rs.y * ga.y + be.y),
                    silu_f((v.z - mu.z) * rs.z * ga.z + be.z), silu_f((v.w - mu.w) * rs.w * ga.w + be.w));
  }
}

__global__ __launch_bounds__(256) void bn_silu_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ ds, const float* __restrict__ partial,
    int nb, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, long rows, int C,
    float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double s0[1024], s1[1024];
  __shared__ float m1[1024], m2[1024];
  bn_fold(partial, nb, C, s0, s1);
  for (int c = threadIdx.x; c < C; c += 256) {
    m1[c] = (float)(s0[c] / (double)rows);             // mean of dz
    m2[c] = (float)(s1[c] / (double)rows);             // mean of dz * xhat
    if (blockIdx.x == 0) {                             // gradients are accumulated (flat buffer)
      dbeta[c] += (float)s0[c];
      dgamma[c] += (float)s1[c];
    }
  }
  __syncthreads();
  const long n4 = rows * (long)(C >> 2);
  const int cg = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c = 4 * (int)(i % cg);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(ds)[i];
    const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (vv[j] - mean[c + j]) * rstd[c + j];
      const float dz = gg[j] * silu_deriv(gamma[c + j] * xh + beta[c + j]);
      o[j] = gamma[c + j] * rstd[c + j] * (dz - m1[c + j] - xh * m2[c + j]);
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

inline bool bad_rows(const void* p, int C) {
  return (reinterpret_cast<uintptr_t>(p) & 15) != 0 || C < 4 || (C & 3) || C > 1024;
}
inline int stream_grid(long n4) {
  long g = (n4 + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" {

int s2t_layernorm_fwd(const float* x, const float* y, float alpha, const float* gamma,
                      const float* beta, long rows, int C, float eps, float* xsum, float* out,
                      float* stats, void* stream) {
  if (rows <= 0) return 0;
  if (bad_rows(x, C) || bad_rows(out, C) || (y && (bad_rows(y, C) || bad_rows(xsum, C)))) return -2;
  const dim3 grid((unsigned)((rows + 3) / 4));
  if (y)
    hipLaunchKernelGGL(layernorm_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, y,
                       alpha, gamma, beta, rows, C, eps, xsum, out, stats);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, y,
                       alpha, gamma, beta, rows, C, eps, xsum, out, stats);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_layernorm_bwd(const float* x, const float* stats, const float* gamma, const float* dy,
                      const float* resid, long rows, int C, float* dx, float* dgamma, float* dbeta,
                      void* stream) {
  if (rows <= 0) return 0;
  if (bad_rows(x, C) || bad_rows(dy, C) || bad_rows(dx, C) || (resid && bad_rows(resid, C)))
    return -2;
  long wgs = (rows + 3) / 4;
  if (wgs > 256) wgs = 256;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream,
                     x, stats, gamma, dy, resid, rows, C, dx, dgamma, dbeta);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_silu_fwd(const float* h, long n, float* a, void* stream) {
  if (n <= 0) return 0;
  if ((n & 3) || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(a) & 15))
    return -2;
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(h), n / 4, reinterpret_cast<float4*>(a));
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_silu_bwd(const float* h, const float* da, long n, float scale, float* dh, void* stream) {
  if (n <= 0) return 0;
  if ((n & 3) || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(da) & 15) ||
      (reinterpret_cast<uintptr_t>(dh) & 15))
    return -2;
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(h), reinterpret_cast<const float4*>(da), n / 4,
                     scale, reinterpret_cast<float4*>(dh));
  S2T_CHECK_LAUNCH();
  return 0;
}

// workspace: S2T_BN_PARTIALS x 2 x C floats
long s2t_bn_workspace_floats(int C) { return (long)S2T_BN_PARTIALS * 2 * C; }

int s2t_bn_silu_fwd(const float* x, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, long* num_batches,
                    long rows, int C, float* y, float* save_mean, float* save_rstd,
                    float* workspace, void* stream) {
  if (rows <= 0) return 0;
  if (bad_rows(x, C) || bad_rows(y, C)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long per = (rows + S2T_BN_PARTIALS - 1) / S2T_BN_PARTIALS;
  const int nb = (int)((rows + per - 1) / per);
  hipLaunchKernelGGL(bn_stats_kernel<0>, dim3(nb), dim3(256), 0, st, x, nullptr, nullptr, nullptr,
                     nullptr, nullptr, rows, C, per, workspace);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_silu_fwd_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0, st, x,
                     workspace, nb, gamma, beta, eps, momentum, running_mean, running_var,
                     num_batches, rows, C, y, save_mean, save_rstd);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_bn_silu_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                      const float* beta, long rows, int C, float* y, void* stream) {
  if (rows <= 0) return 0;
  if (bad_rows(x, C) || bad_rows(y, C) || bad_rows(mean, C) || bad_rows(rstd, C) ||
      bad_rows(gamma, C) || bad_rows(beta, C))
    return -2;
  hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, mean, rstd, gamma, beta, rows, C, y);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_bn_silu_bwd(const float* x, const float* ds, const float* save_mean,
                    const float* save_rstd, const float* gamma, const float* beta, long rows, int C,
                    float* dx, float* dgamma, float* dbeta, float* workspace, void* stream) {
  if (rows <= 0) return 0;
  if (bad_rows(x, C) || bad_rows(ds, C) || bad_rows(dx, C)) return -2;
  hipStream_t st = (hipStream_t)stream;
  const long per = (rows + S2T_BN_PARTIALS - 1) / S2T_BN_PARTIALS;
  const int nb = (int)((rows + per - 1) / per);
  hipLaunchKernelGGL(bn_stats_kernel<1>, dim3(nb), dim3(256), 0, st, x, ds, save_mean, save_rstd,
                     gamma, beta, rows, C, per, workspace);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_silu_bwd_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0, st, x, ds,
                     workspace, nb, save_mean, save_rstd, gamma, beta, rows, C, dx, dgamma, dbeta);
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
