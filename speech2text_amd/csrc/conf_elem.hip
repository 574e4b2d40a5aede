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
// and keeps its lanes' channel sums in registers; after one LDS reduction the workgroup stores
// its 2C partial sums (plain stores), and ln_fold_kernel adds the partials of up to 8
// LayerNorms into their gradient words in a fixed order.  (Atomics straight into the 2C gradient
// words cost ~25 us per call: 256 workgroups adding to the same 512 addresses at once.)
// The pass is latency-bound (one row = 3 loads + two wave reductions), so a workgroup carries
// NW waves (16 for C <= 256): many rows in flight per CU.
template <int NV, int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ dy, const float* __restrict__ resid, long rows, int C,
    float* __restrict__ dx, float* __restrict__ partial) {
  __shared__ float red[NW - 1][2 * 256 * NV];   // waves 1.. park their channel sums here
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = C >> 2;
  float4 g[NV], ag[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c4 = lane + 64 * i;
    g[i] = c4 < nv ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const long stride = (long)gridDim.x * NW;
  for (long row = (long)blockIdx.x * NW + wave; row < rows; row += stride) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float4 xh[NV], d[NV], rr[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        const float4 xv = reinterpret_cast<const float4*>(x + row * C)[c4];
        d[i] = reinterpret_cast<const float4*>(dy + row * C)[c4];
        if (resid) rr[i] = reinterpret_cast<const float4*>(resid + row * C)[c4];
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
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 < nv) {
        float4 o;
        o.x = rstd * (d[i].x - s1 - xh[i].x * s2);
        o.y = rstd * (d[i].y - s1 - xh[i].y * s2);
        o.z = rstd * (d[i].z - s1 - xh[i].z * s2);
        o.w = rstd * (d[i].w - s1 - xh[i].w * s2);
        if (resid) { o.x += rr[i].x; o.y += rr[i].y; o.z += rr[i].z; o.w += rr[i].w; }
        reinterpret_cast<float4*>(dx + row * C)[c4] = o;
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float* p = red[wave - 1] + 8 * (lane + 64 * i);
      *reinterpret_cast<float4*>(p) = ag[i];
      *reinterpret_cast<float4*>(p + 4) = ab[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c4 = lane + 64 * i;
      if (c4 >= nv) continue;
      float4 a = ag[i], b = ab[i];
      for (int w = 0; w < NW - 1; ++w) {
        const float* p = red[w] + 8 * c4;
        const float4 a2 = *reinterpret_cast<const float4*>(p);
        const float4 b2 = *reinterpret_cast<const float4*>(p + 4);
        a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        b.x += b2.x; b.y += b2.y; b.z += b2.z; b.w += b2.w;
      }
      float* p = partial + (long)blockIdx.x * 2 * C;
      reinterpret_cast<float4*>(p)[c4] = a;
      reinterpret_cast<float4*>(p + C)[c4] = b;
    }
  }
}

constexpr int LN_MAX_FOLD = 8;
struct LnFoldArgs {
  int n, C;
  const float* partial[LN_MAX_FOLD];
  int nwg[LN_MAX_FOLD];
  float* dgamma[LN_MAX_FOLD];
  float* dbeta[LN_MAX_FOLD];
};
// grid (ceil(C / 64), n, 2 stats): thread (channel, row group of 16) sums every 16th partial row
// (short dependent chains, many loads in flight), then the 16 pieces meet in LDS
__global__ __launch_bounds__(1024) void ln_fold_kernel(LnFoldArgs a) {
  __shared__ float red[16][64];
  const int t = threadIdx.x, chl = t & 63, grp = t >> 6, stat = blockIdx.z;
  const int c = blockIdx.x * 64 + chl, it = blockIdx.y, C = a.C;
  float acc = 0.f;
  if (c < C) {
    const float* p = a.partial[it] + stat * C + c;
    const int nwg = a.nwg[it];
#pragma unroll 4
    for (int b = grp; b < nwg; b += 16) acc += p[(long)b * 2 * C];
  }
  red[grp][chl] = acc;
  __syncthreads();
  if (grp == 0 && c < C) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[g][chl];
    float* dst = stat == 0 ? a.dgamma[it] : a.dbeta[it];
    dst[c] += s;
  }
}

inline int ln_bwd_waves(int C) { return C <= 256 ? 16 : (C <= 512 ? 8 : 4); }
inline int ln_bwd_wgs(long rows, int C) {
  const int nw = ln_bwd_waves(C);
  const long wgs = (rows + nw - 1) / nw;
  return (int)(wgs > 256 ? 256 : (wgs < 1 ? 1 : wgs));
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

// ------------------------------------------------------------------ dropout (nn.Dropout sites)
// The keep decision of element i is the stateless hash the attention kernels use (splitmix64
// finaliser of seed + i * golden): forward and backward regenerate the same mask from (seed, p),
// no mask tensor is stored.  thr = p * 2^32, kept elements are scaled by 1 / (1 - p).
__device__ __forceinline__ float keep_scale(unsigned long long seed, long idx, unsigned thr,
                                            float inv_keep) {
  unsigned long long z = seed + (unsigned long long)idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32) >= thr ? inv_keep : 0.f;
}

// out = x + alpha * drop(y)   (x may be null: the masked, scaled gradient of a dropout site)
__global__ __launch_bounds__(256) void dropout_add_kernel(const float4* __restrict__ x,
                                                          const float4* __restrict__ y, long n4,
                                                          float alpha, unsigned thr, float inv_keep,
                                                          unsigned long long seed,
                                                          float4* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = y[i];
    float4 o = x ? x[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x = fmaf(alpha * keep_scale(seed, 4 * i, thr, inv_keep), v.x, o.x);
    o.y = fmaf(alpha * keep_scale(seed, 4 * i + 1, thr, inv_keep), v.y, o.y);
    o.z = fmaf(alpha * keep_scale(seed, 4 * i + 2, thr, inv_keep), v.z, o.z);
    o.w = fmaf(alpha * keep_scale(seed, 4 * i + 3, thr, inv_keep), v.w, o.w);
    out[i] = o;
  }
}

// a = drop(silu(h));  dh = scale * da * drop-mask * silu'(h)
__global__ __launch_bounds__(256) void silu_drop_fwd_kernel(const float4* __restrict__ h, long n4,
                                                            unsigned thr, float inv_keep,
                                                            unsigned long long seed,
                                                            float4* __restrict__ a) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = h[i];
    a[i] = make_float4(silu_f(v.x) * keep_scale(seed, 4 * i, thr, inv_keep),
                       silu_f(v.y) * keep_scale(seed, 4 * i + 1, thr, inv_keep),
                       silu_f(v.z) * keep_scale(seed, 4 * i + 2, thr, inv_keep),
                       silu_f(v.w) * keep_scale(seed, 4 * i + 3, thr, inv_keep));
  }
}
__global__ __launch_bounds__(256) void silu_drop_bwd_kernel(const float4* __restrict__ h,
                                                            const float4* __restrict__ da, long n4,
                                                            float scale, unsigned thr,
                                                            float inv_keep, unsigned long long seed,
                                                            float4* __restrict__ dh) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = h[i], g = da[i];
    dh[i] = make_float4(scale * g.x * silu_deriv(v.x) * keep_scale(seed, 4 * i, thr, inv_keep),
                        scale * g.y * silu_deriv(v.y) * keep_scale(seed, 4 * i + 1, thr, inv_keep),
                        scale * g.z * silu_deriv(v.z) * keep_scale(seed, 4 * i + 2, thr, inv_keep),
                        scale * g.w * silu_deriv(v.w) * keep_scale(seed, 4 * i + 3, thr, inv_keep));
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

// Fold of the partials: one workgroup per 64 channels, thread (ch, stat, row group of 8) adds every
// 8th partial row in a fixed order (double), then the pieces meet in LDS.
// KIND 0 -> save_mean / save_rstd (+ running statistics);  KIND 1 -> m[0][c] = mean(dz),
// m[1][c] = mean(dz xhat) for the apply pass, dgamma / dbeta accumulated.
template <int KIND>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(
    const float* __restrict__ partial, int nb, long rows, int C, float eps, float momentum,
    float* __restrict__ running_mean, float* __restrict__ running_var,
    long* __restrict__ num_batches, float* __restrict__ out0, float* __restrict__ out1,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double red[16][64];
  const int t = threadIdx.x, chl = t & 63, grp = t >> 6, stat = grp & 1, piece = grp >> 1;
  const int c = blockIdx.x * 64 + chl;
  double acc = 0.0;
  if (c < C) {
    const float* p = partial + stat * C + c;
#pragma unroll 4
    for (int b = piece; b < nb; b += 8) acc += (double)p[(long)b * 2 * C];
  }
  red[grp][chl] = acc;
  __syncthreads();
  if (grp == 0 && c < C) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      s0 += red[2 * g][chl];
      s1 += red[2 * g + 1][chl];
    }
    if (KIND == 0) {
      const double m = s0 / (double)rows;
      double var = s1 / (double)rows - m * m;           // biased (normalisation)
      if (var < 0.0) var = 0.0;
      out0[c] = (float)m;
      out1[c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    } else {
      out0[c] = (float)(s0 / (double)rows);
      out1[c] = (float)(s1 / (double)rows);
      dbeta[c] += (float)s0;                            // gradients are accumulated (flat buffer)
      dgamma[c] += (float)s1;
    }
  }
  if (KIND == 0 && blockIdx.x == 0 && t == 0 && num_batches) *num_batches += 1;
}

__global__ __launch_bounds__(256) void bn_silu_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ ds, const float* __restrict__ m1,
    const float* __restrict__ m2, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, long rows, int C,
    float* __restrict__ dx) {
  const long n4 = rows * (long)(C >> 2);
  const int cg = C >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % cg);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 g = reinterpret_cast<const float4*>(ds)[i];
    const float4 mu = reinterpret_cast<const float4*>(mean)[c4];
    const float4 rs = reinterpret_cast<const float4*>(rstd)[c4];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 be = reinterpret_cast<const float4*>(beta)[c4];
    const float4 a1 = reinterpret_cast<const float4*>(m1)[c4];
    const float4 a2 = reinterpret_cast<const float4*>(m2)[c4];
    const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {g.x, g.y, g.z, g.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
    const float gav[4] = {ga.x, ga.y, ga.z, ga.w}, bev[4] = {be.x, be.y, be.z, be.w};
    const float a1v[4] = {a1.x, a1.y, a1.z, a1.w}, a2v[4] = {a2.x, a2.y, a2.z, a2.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xh = (vv[j] - muv[j]) * rsv[j];
      const float dz = gg[j] * silu_deriv(gav[j] * xh + bev[j]);
      o[j] = gav[j] * rsv[j] * (dz - a1v[j] - xh * a2v[j]);
    }
    reinterpret_cast<float4*>(dx)[i] = make_float4(o[0], o[1], o[2], o[3]);
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

long s2t_layernorm_bwd_partial_floats(long rows, int C) {
  return (long)ln_bwd_wgs(rows, C) * 2 * C;
}

int s2t_layernorm_bwd(const float* x, const float* stats, const float* gamma, const float* dy,
                      const float* resid, long rows, int C, float* dx, float* partial,
                      void* stream) {
  if (rows <= 0) return 0;
  if (bad_rows(x, C) || bad_rows(dy, C) || bad_rows(dx, C) || (resid && bad_rows(resid, C)))
    return -2;
  const int wgs = ln_bwd_wgs(rows, C);
#define S2T_LN_BWD(NV, NW)                                                                       \
  hipLaunchKernelGGL((layernorm_bwd_kernel<NV, NW>), dim3((unsigned)wgs), dim3(64 * NW), 0,      \
                     (hipStream_t)stream, x, stats, gamma, dy, resid, rows, C, dx, partial);
  if (C <= 256) S2T_LN_BWD(1, 16)
  else if (C <= 512) S2T_LN_BWD(2, 8)
  else if (C <= 768) S2T_LN_BWD(3, 4)
  else S2T_LN_BWD(4, 4)
#undef S2T_LN_BWD
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_layernorm_param_grad(int n, const S2tLnFold* items, int C, void* stream) {
  if (n <= 0) return 0;
  if (!items || C < 4 || (C & 3) || C > 1024) return -1;
  for (int base = 0; base < n; base += LN_MAX_FOLD) {
    LnFoldArgs a;
    a.n = n - base < LN_MAX_FOLD ? n - base : LN_MAX_FOLD;
    a.C = C;
    for (int i = 0; i < a.n; ++i) {
      const S2tLnFold& it = items[base + i];
      a.partial[i] = it.partial;
      a.nwg[i] = ln_bwd_wgs(it.rows, C);
      a.dgamma[i] = it.dgamma;
      a.dbeta[i] = it.dbeta;
    }
    hipLaunchKernelGGL(ln_fold_kernel, dim3((C + 63) / 64, a.n, 2), dim3(1024), 0,
                       (hipStream_t)stream, a);
    S2T_CHECK_LAUNCH();
  }
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

static bool drop_params(float p, unsigned* thr, float* inv_keep) {
  if (!(p >= 0.f) || p >= 1.f) return false;
  *thr = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  if (p > 0.f && *thr == 0u) *thr = 1u;
  *inv_keep = 1.f / (1.f - p);
  return true;
}

int s2t_dropout_add(const float* x, const float* y, long n, float alpha, float p,
                    unsigned long long seed, float* out, void* stream) {
  if (n <= 0) return 0;
  unsigned thr;
  float inv_keep;
  if (!drop_params(p, &thr, &inv_keep)) return -1;
  if ((n & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return -2;
  hipLaunchKernelGGL(dropout_add_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<const float4*>(y), n / 4,
                     alpha, thr, inv_keep, seed, reinterpret_cast<float4*>(out));
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_silu_drop_fwd(const float* h, long n, float p, unsigned long long seed, float* a,
                      void* stream) {
  if (n <= 0) return 0;
  unsigned thr;
  float inv_keep;
  if (!drop_params(p, &thr, &inv_keep)) return -1;
  if ((n & 3) || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(a) & 15))
    return -2;
  hipLaunchKernelGGL(silu_drop_fwd_kernel, dim3(stream_grid(n / 4)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float4*>(h), n / 4, thr, inv_keep,
                     seed, reinterpret_cast<float4*>(a));
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_silu_drop_bwd(const float* h, const float* da, long n, float scale, float p,
                      unsigned long long seed, float* dh, void* stream) {
  if (n <= 0) return 0;
  unsigned thr;
  float inv_keep;
  if (!drop_params(p, &thr, &inv_keep)) return -1;
  if ((n & 3) || (reinterpret_cast<uintptr_t>(h) & 15) || (reinterpret_cast<uintptr_t>(da) & 15) ||
      (reinterpret_cast<uintptr_t>(dh) & 15))
    return -2;
  hipLaunchKernelGGL(silu_drop_bwd_kernel, dim3(stream_grid(n / 4)), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const float4*>(h),
                     reinterpret_cast<const float4*>(da), n / 4, scale, thr, inv_keep, seed,
                     reinterpret_cast<float4*>(dh));
  S2T_CHECK_LAUNCH();
  return 0;
}

// workspace: S2T_BN_PARTIALS x 2 x C floats of partial sums + 2 x C folded means (backward)
long s2t_bn_workspace_floats(int C) { return (long)(S2T_BN_PARTIALS + 1) * 2 * C; }

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
  hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3((C + 63) / 64), dim3(1024), 0, st, workspace, nb,
                     rows, C, eps, momentum, running_mean, running_var, num_batches, save_mean,
                     save_rstd, nullptr, nullptr);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_silu_apply_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0, st, x,
                     save_mean, save_rstd, gamma, beta, rows, C, y);
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
  float* m1 = workspace + (long)S2T_BN_PARTIALS * 2 * C;
  hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((C + 63) / 64), dim3(1024), 0, st, workspace, nb,
                     rows, C, 0.f, 0.f, nullptr, nullptr, nullptr, m1, m1 + C, dgamma, dbeta);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_silu_bwd_kernel, dim3(stream_grid(rows * (C / 4))), dim3(256), 0, st, x, ds,
                     m1, m1 + C, save_mean, save_rstd, gamma, beta, rows, C, dx);
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
