// Streaming (HBM-bound) zipformer kernels for gfx950: Swoosh activations, BiasNorm,
// Balancer statistics / gradient update.  All are one-read one-write passes with
// 16-byte per-lane accesses where the layout allows; reductions use 64-lane waves.
//
// Reference semantics: model/layer/scaling.py:1340-1343,1418-1423 (SwooshL/R),
// :347-399 (BiasNormFunction), :741-789 (BalancerFunction.backward, closed form
// derived in DESIGN.md), :1559-1578 (activation derivative recomputed in backward).
#include "common.h"
#include <algorithm>
#include <stdlib.h>

namespace {

// log1p(e) for e in (0, 1]: hardware log of u = fl(1 + e), times e / (u - 1) to undo the rounding
// of the sum (the classic compensation) -- a handful of instructions instead of libm's log1pf
__device__ __forceinline__ float log1p_fast(float e) {
  const float u = 1.f + e;
  return u == 1.f ? e : __logf(u) * __fdividef(e, u - 1.f);
}

__device__ __forceinline__ float swoosh_f(float x, float off, float c) {
  const float z = x - off;
  // log(1+exp(z)) = max(z,0) + log1p(exp(-|z|))
  return fmaxf(z, 0.f) + log1p_fast(__expf(-fabsf(z))) - 0.08f * x - c;
}
__device__ __forceinline__ float swoosh_d(float x, float off) {
  return __fdividef(1.f, 1.f + __expf(off - x)) - 0.08f;
}

__global__ __launch_bounds__(256) void swoosh_fwd_kernel(const float* __restrict__ x,
                                                         float* __restrict__ y, long n, float off,
                                                         float c) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* y4 = reinterpret_cast<float4*>(y);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = x4[i];
    v.x = swoosh_f(v.x, off, c);
    v.y = swoosh_f(v.y, off, c);
    v.z = swoosh_f(v.z, off, c);
    v.w = swoosh_f(v.w, off, c);
    y4[i] = v;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = swoosh_f(x[i], off, c);
}

__global__ __launch_bounds__(256) void swoosh_bwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ g,
                                                         float* __restrict__ d, long n,
                                                         float off) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* d4 = reinterpret_cast<float4*>(d);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = x4[i], gg = g4[i];
    float4 o;
    o.x = gg.x * swoosh_d(v.x, off);
    o.y = gg.y * swoosh_d(v.y, off);
    o.z = gg.z * swoosh_d(v.z, off);
    o.w = gg.w * swoosh_d(v.w, off);
    d4[i] = o;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    d[i] = g[i] * swoosh_d(x[i], off);
}

// ---------------------------------------------------------------- BiasNorm
// one wave per row of D channels; scales[row] = exp(ls) * mean((x-b)^2)^-0.5
// tbT > 0: input row r = b tbT + t (batch-major) is WRITTEN as row t tbB + b (time-major): the
// (B,T,C) -> (T,B,C) transposition of the frontend's output rides in this pass
__global__ __launch_bounds__(256) void biasnorm_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ ls,
    long rows, int D, float* __restrict__ y, float* __restrict__ scales, int tbT, int tbB) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * D;
  float acc = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] - bias[c];
    acc = fmaf(d, d, acc);
  }
  acc = wave_sum(acc);
  const float s = rsqrtf(acc / D) * expf(ls[0]);
  if (lane == 0) scales[row] = s;
  const long orow = tbT > 0 ? (row % tbT) * tbB + row / tbT : row;
  float* yr = y + orow * D;
  for (int c = lane; c < D; c += 64) yr[c] = xr[c] * s;
}

// dx = s*g - s*(x-b)*A/(D*ms), A = sum_j g_j x_j, ms = mean((x-b)^2)
// dbias += s*(x-b)*A/(D*ms) ; dls += A*s.   Persistent blocks keep per-column partial
// sums in registers over all their rows, then one atomic per column per block.
template <int CPL>  // columns per lane: D <= 64*CPL
__global__ __launch_bounds__(256) void biasnorm_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ scales,
    const float* __restrict__ g, long rows, int D, float* __restrict__ dx,
    float* __restrict__ dbias, float* __restrict__ dls, int tbT, int tbB) {
  // tbT > 0: g is in the time-major row order the forward wrote (row r = b tbT + t <-> t tbB + b)
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  float b[CPL], db[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = lane + 64 * j;
    b[j] = c < D ? bias[c] : 0.f;
    db[j] = 0.f;
  }
  float dl = 0.f;
  // two rows per trip: both rows' loads are issued before either reduction (a wave that walks one
  // row at a time sits through a full memory round trip per row)
  for (long row = wave; row < rows; row += 2 * nwaves) {
    const long row2 = row + nwaves;
    const bool has2 = row2 < rows;
    const long rowb = has2 ? row2 : row;
    const float* xr = x + row * D;
    const float* gr = g + (tbT > 0 ? (row % tbT) * tbB + row / tbT : row) * D;
    const float* xr2 = x + rowb * D;
    const float* gr2 = g + (tbT > 0 ? (rowb % tbT) * tbB + rowb / tbT : rowb) * D;
    float xv[CPL], gv[CPL], xw[CPL], gw[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      xv[j] = c < D ? xr[c] : 0.f;
      gv[j] = c < D ? gr[c] : 0.f;
      xw[j] = c < D ? xr2[c] : 0.f;
      gw[j] = c < D ? gr2[c] : 0.f;
    }
    float A = 0.f, ss = 0.f, A2 = 0.f, ss2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      A = fmaf(gv[j], xv[j], A);
      A2 = fmaf(gw[j], xw[j], A2);
      const float d = c < D ? xv[j] - b[j] : 0.f, d2 = c < D ? xw[j] - b[j] : 0.f;
      ss = fmaf(d, d, ss);
      ss2 = fmaf(d2, d2, ss2);
    }
    A = wave_sum(A);
    ss = wave_sum(ss);
    A2 = wave_sum(A2);
    ss2 = wave_sum(ss2);
    const float s = scales[row], s2 = has2 ? scales[row2] : 0.f;
    const float coef = s * A / ss, coef2 = has2 ? s2 * A2 / ss2 : 0.f;   // ss = D*ms
    dl += A * s + (has2 ? A2 * s2 : 0.f);
    float* dr = dx + row * D;
    float* dr2 = dx + row2 * D;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < D) {
        const float t = coef * (xv[j] - b[j]);
        dr[c] = s * gv[j] - t;
        db[j] += t;
        if (has2) {
          const float t2 = coef2 * (xw[j] - b[j]);
          dr2[c] = s2 * gw[j] - t2;
          db[j] += t2;
        }
      }
    }
  }
  // one atomic per column per WORKGROUP (the four waves add up through LDS first): every atomic
  // of this pass lands on the same D addresses, so their count is what the tail costs
  __shared__ float s_db[4][64 * CPL];
  __shared__ float s_dl[4];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < CPL; ++j) s_db[wv][lane + 64 * j] = db[j];
  if (lane == 0) s_dl[wv] = dl;              // dl is wave-uniform (built from wave sums)
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      const float t = (s_db[0][c] + s_db[1][c]) + (s_db[2][c] + s_db[3][c]);
      if (c < D && t != 0.f) atomicAdd(&dbias[c], t);
    }
    const float tl = (s_dl[0] + s_dl[1]) + (s_dl[2] + s_dl[3]);
    if (lane == 0 && tl != 0.f) atomicAdd(dls, tl);
  }
}

// x * y rounded on its own (never contracted into a following add / subtract): the fused passes
// below must produce the value the separate passes STORED
__device__ __forceinline__ float mul_rounded(float x, float y) {
#pragma clang fp contract(off)
  const float p = x * y;
  return p;
}

// ---------------------------------------------------------------- BiasNorm + bypass, fused
// The end of a Zipformer2EncoderLayer (model/encoder/zipformer.py:1330-1337): out = orig +
// (norm(x) - orig) * scale[c] (* fm[b, c]: the stack's feature mask).  One wave per row: the row's
// scale, the normalised row and the bypass combination in ONE pass -- norm(x) itself is never
// stored (backward recomputes it as x * scales[row]).
__global__ __launch_bounds__(256) void norm_bypass_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ ls,
    const float* __restrict__ orig, const float* __restrict__ bscale, const float* __restrict__ fm,
    int B, long rows, int D, float* __restrict__ out, float* __restrict__ scales) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + row * D;
  float acc = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float d = xr[c] - bias[c];
    acc = fmaf(d, d, acc);
  }
  acc = wave_sum(acc);
  const float s = rsqrtf(acc / D) * expf(ls[0]);
  if (lane == 0) scales[row] = s;
  const float* orow = orig + row * D;
  const float* mrow = fm ? fm + (row % B) * D : nullptr;
  float* yr = out + row * D;
  for (int c = lane; c < D; c += 64) {
    const float a = orow[c];
    float v = fmaf(mul_rounded(xr[c], s) - a, bscale[c], a);   // (the product rounded as the stored norm(x) was)
    if (mrow) v *= mrow[c];
    yr[c] = v;
  }
}

// Backward of the pair: g' = g * fm;  d_orig = g' (1 - scale),  g10 = g' scale,  d_scale[c] += sum g'
// (x s - orig);  then BiasNorm's backward on g10 (biasnorm_bwd_kernel's formulas).  One wave per row,
// two rows per trip; the per-column sums stay in registers over all rows of a wave.
template <int CPL>
__global__ __launch_bounds__(256) void norm_bypass_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ scales,
    const float* __restrict__ orig, const float* __restrict__ bscale, const float* __restrict__ g,
    const float* __restrict__ fm, int B, long rows, int D, float* __restrict__ dx,
    float* __restrict__ d_orig, float* __restrict__ d_bscale, float* __restrict__ dbias,
    float* __restrict__ dls) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  float b[CPL], k[CPL], db[CPL], dk[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = lane + 64 * j;
    b[j] = c < D ? bias[c] : 0.f;
    k[j] = c < D ? bscale[c] : 0.f;
    db[j] = 0.f;
    dk[j] = 0.f;
  }
  float dl = 0.f;
  for (long row = wave; row < rows; row += 2 * nwaves) {
    const long row2 = row + nwaves;
    const bool has2 = row2 < rows;
    const long ra = row, rb = has2 ? row2 : row;
    float xv[CPL], gv[CPL], ov[CPL], xw[CPL], gw[CPL], ow[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      const bool ok = c < D;
      xv[j] = ok ? x[ra * D + c] : 0.f;
      gv[j] = ok ? g[ra * D + c] : 0.f;
      ov[j] = ok ? orig[ra * D + c] : 0.f;
      xw[j] = ok ? x[rb * D + c] : 0.f;
      gw[j] = ok ? g[rb * D + c] : 0.f;
      ow[j] = ok ? orig[rb * D + c] : 0.f;
      if (fm) {
        gv[j] *= ok ? fm[(ra % B) * D + c] : 0.f;
        gw[j] *= ok ? fm[(rb % B) * D + c] : 0.f;
      }
    }
    const float s = scales[ra], s2 = scales[rb];
    float A = 0.f, ss = 0.f, A2 = 0.f, ss2 = 0.f;
    float* d0 = d_orig + ra * D;
    float* d02 = d_orig + rb * D;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < D) {
        // the bypass: this row's share of d_scale, the gradient that stays on orig, and g10
        const float t = gv[j] * k[j];
        dk[j] = fmaf(gv[j], mul_rounded(xv[j], s) - ov[j], dk[j]);
        d0[c] = gv[j] - t;
        gv[j] = t;
        if (has2) {
          const float t2 = gw[j] * k[j];
          dk[j] = fmaf(gw[j], mul_rounded(xw[j], s2) - ow[j], dk[j]);
          d02[c] = gw[j] - t2;
          gw[j] = t2;
        }
      }
      A = fmaf(gv[j], xv[j], A);
      A2 = fmaf(gw[j], xw[j], A2);
      const float d = c < D ? xv[j] - b[j] : 0.f, d2 = c < D ? xw[j] - b[j] : 0.f;
      ss = fmaf(d, d, ss);
      ss2 = fmaf(d2, d2, ss2);
    }
    A = wave_sum(A);
    ss = wave_sum(ss);
    A2 = wave_sum(A2);
    ss2 = wave_sum(ss2);
    const float coef = s * A / ss, coef2 = has2 ? s2 * A2 / ss2 : 0.f;   // ss = D * mean((x - b)^2)
    dl += A * s + (has2 ? A2 * s2 : 0.f);
    float* dr = dx + ra * D;
    float* dr2 = dx + rb * D;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < D) {
        const float t = coef * (xv[j] - b[j]);
        dr[c] = s * gv[j] - t;
        db[j] += t;
        if (has2) {
          const float t2 = coef2 * (xw[j] - b[j]);
          dr2[c] = s2 * gw[j] - t2;
          db[j] += t2;
        }
      }
    }
  }
  // one atomic per column per WORKGROUP (the four waves add up through LDS first)
  __shared__ float s_db[4][64 * CPL], s_dk[4][64 * CPL];
  __shared__ float s_dl[4];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    s_db[wv][lane + 64 * j] = db[j];
    s_dk[wv][lane + 64 * j] = dk[j];
  }
  if (lane == 0) s_dl[wv] = dl;
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      const float t = (s_db[0][c] + s_db[1][c]) + (s_db[2][c] + s_db[3][c]);
      const float u = (s_dk[0][c] + s_dk[1][c]) + (s_dk[2][c] + s_dk[3][c]);
      if (c < D && t != 0.f) atomicAdd(&dbias[c], t);
      if (c < D && u != 0.f) atomicAdd(&d_bscale[c], u);
    }
    const float tl = (s_dl[0] + s_dl[1]) + (s_dl[2] + s_dl[3]);
    if (lane == 0 && tl != 0.f) atomicAdd(dls, tl);
  }
}

// The same pass in a 16-byte form (D % 4 == 0): lane = float4 chunks lane, lane + 64, ... of a row, RT rows
// requested together (3 Q RT loads of 16 bytes in flight per lane; the scalar form above issues 4-byte loads
// and needs ~1000 workgroups of them -- whose per-column atomics then serialise at the end: 2.5 TB/s at the
// C3 shapes).  Per-element arithmetic as above; the two row sums are taken in another order (last bits).
template <int Q, int RT>
__global__ __launch_bounds__(256) void norm_bypass_bwd16_kernel(
    const float* __restrict__ x, const float* __restrict__ bias, const float* __restrict__ scales,
    const float* __restrict__ orig, const float* __restrict__ bscale, const float* __restrict__ g,
    const float* __restrict__ fm, int B, long rows, int D, float* __restrict__ dx,
    float* __restrict__ d_orig, float* __restrict__ d_bscale, float* __restrict__ dbias,
    float* __restrict__ dls) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long wave = (long)blockIdx.x * 4 + wv;
  const long nwaves = (long)gridDim.x * 4;
  const int D4 = D >> 2;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* o4 = reinterpret_cast<const float4*>(orig);
  const float4* f4 = reinterpret_cast<const float4*>(fm);
  float4 b[Q], k[Q], db[Q], dk[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int cq = lane + 64 * q;
    b[q] = cq < D4 ? reinterpret_cast<const float4*>(bias)[cq] : z4;
    k[q] = cq < D4 ? reinterpret_cast<const float4*>(bscale)[cq] : z4;
    db[q] = z4;
    dk[q] = z4;
  }
  float dl = 0.f;
  for (long r0 = wave; r0 < rows; r0 += RT * nwaves) {
    float4 xv[RT][Q], gv[RT][Q], ov[RT][Q];
    float sc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long row = r0 + i * nwaves;
      const long rr = row < rows ? row : r0;
      sc[i] = scales[rr];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int cq = lane + 64 * q;
        const bool ok = cq < D4;
        const long idx = rr * D4 + (ok ? cq : 0);
        xv[i][q] = x4[idx];
        gv[i][q] = g4[idx];
        ov[i][q] = o4[idx];
        if (fm) {
          const float4 m = f4[(rr % B) * D4 + (ok ? cq : 0)];
          gv[i][q].x *= m.x; gv[i][q].y *= m.y; gv[i][q].z *= m.z; gv[i][q].w *= m.w;
        }
        if (!ok) { xv[i][q] = z4; gv[i][q] = z4; ov[i][q] = z4; }
      }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const long row = r0 + i * nwaves;
      if (row >= rows) break;                       // (wave-uniform)
      const float s = sc[i];
      float A = 0.f, ss = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int cq = lane + 64 * q;
        float* xe = reinterpret_cast<float*>(&xv[i][q]);
        float* ge = reinterpret_cast<float*>(&gv[i][q]);
        const float* oe = reinterpret_cast<const float*>(&ov[i][q]);
        const float* ke = reinterpret_cast<const float*>(&k[q]);
        const float* be = reinterpret_cast<const float*>(&b[q]);
        float* dke = reinterpret_cast<float*>(&dk[q]);
        float4 d0;
        float* d0e = reinterpret_cast<float*>(&d0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // the bypass: this row's share of d_scale, the gradient that stays on orig, and g10
          const float t = mul_rounded(ge[e], ke[e]);          // (g10 as the separate bypass pass stores it)
          dke[e] = fmaf(ge[e], mul_rounded(xe[e], s) - oe[e], dke[e]);
          d0e[e] = fmaf(-ge[e], ke[e], ge[e]);                // (g - g k, contracted as the separate pass has it)
          ge[e] = t;
          A = fmaf(t, xe[e], A);
          const float d = xe[e] - be[e];
          ss = fmaf(d, d, ss);
        }
        if (cq < D4) reinterpret_cast<float4*>(d_orig)[row * D4 + cq] = d0;
      }
      A = wave_sum(A);
      ss = wave_sum(ss);
      const float coef = s * A / ss;                // ss = D * mean((x - b)^2)
      dl += A * s;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int cq = lane + 64 * q;
        const float* xe = reinterpret_cast<const float*>(&xv[i][q]);
        const float* ge = reinterpret_cast<const float*>(&gv[i][q]);
        const float* be = reinterpret_cast<const float*>(&b[q]);
        float* dbe = reinterpret_cast<float*>(&db[q]);
        float4 o;
        float* oe = reinterpret_cast<float*>(&o);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = coef * (xe[e] - be[e]);
          oe[e] = s * ge[e] - t;
          dbe[e] += t;
        }
        if (cq < D4) reinterpret_cast<float4*>(dx)[row * D4 + cq] = o;
      }
    }
  }
  // one atomic per column per WORKGROUP (the four waves add up through LDS first), few workgroups
  __shared__ float4 s_db[4][64 * Q], s_dk[4][64 * Q];
  __shared__ float s_dl[4];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    s_db[wv][lane + 64 * q] = db[q];
    s_dk[wv][lane + 64 * q] = dk[q];
  }
  if (lane == 0) s_dl[wv] = dl;
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    const float* p0 = reinterpret_cast<const float*>(&s_db[0][0]);
    const float* p1 = reinterpret_cast<const float*>(&s_dk[0][0]);
    const int W = 256 * Q;
    const float t = (p0[c] + p0[W + c]) + (p0[2 * W + c] + p0[3 * W + c]);
    const float u = (p1[c] + p1[W + c]) + (p1[2 * W + c] + p1[3 * W + c]);
    if (t != 0.f) atomicAdd(&dbias[c], t);
    if (u != 0.f) atomicAdd(&d_bscale[c], u);
  }
  if (threadIdx.x == 0) {
    const float tl = (s_dl[0] + s_dl[1]) + (s_dl[2] + s_dl[3]);
    if (tl != 0.f) atomicAdd(dls, tl);
  }
}

// ---------------------------------------------------------------- column statistics
// x viewed as [rows][ld] with C used columns: sum[c] += x, sumsq[c] += x^2 (atomics once per
// block).  Each thread owns column (threadIdx.x % cols_per_pass) and strides over rows.
// (Round 6 tried the 16-byte form of this pass and of the update below -- a wave owns whole rows,
// lane = column quads -- and measured it SLOWER in the step, 34.7 against 34.3 ms, two pairs on one
// box: at 20-30 us these launches are their fixed parts -- the per-workgroup coefficient prologue,
// the atomics' tail, the launch itself -- not their loads.  Removed again)
__global__ __launch_bounds__(256) void col_stats_kernel(const float* __restrict__ x, long rows,
                                                        int C, long ld, float* __restrict__ sum,
                                                        float* __restrict__ sumsq) {
  // blockDim = (64, 4): x-dim over columns (coalesced), y-dim over rows
  for (int c0 = blockIdx.y * 64; c0 < C; c0 += gridDim.y * 64) {
    const int c = c0 + threadIdx.x;
    float s = 0.f, q = 0.f;
    if (c < C) {
      // 8 row loads in flight per thread: the pass is latency-bound otherwise
      const long step = (long)gridDim.x * 4;
      long r = (long)blockIdx.x * 4 + threadIdx.y;
      for (; r + 7 * step < rows; r += 8 * step) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[(r + u * step) * ld + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s += v[u];
          q = fmaf(v[u], v[u], q);
        }
      }
      for (; r < rows; r += step) {
        const float v = x[r * ld + c];
        s += v;
        q = fmaf(v, v, q);
      }
    }
    __shared__ float sh[2][4][64];
    sh[0][threadIdx.y][threadIdx.x] = s;
    sh[1][threadIdx.y][threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.y == 0 && c < C) {
      s = sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x] + sh[0][2][threadIdx.x] +
          sh[0][3][threadIdx.x];
      q = sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x] + sh[1][2][threadIdx.x] +
          sh[1][3][threadIdx.x];
      atomicAdd(&sum[c], s);
      atomicAdd(&sumsq[c], q);
    }
    __syncthreads();
  }
}

// Few channels (the frontend's first convolutions: C = 8 and 32 over 5.1 M / 1.2 M rows): the kernel
// above gives a lane a COLUMN of a 64-wide group, so at C = 8 one lane in eight works.  Here the
// (rows, C) matrix is contiguous (ld == C, C a power of two <= 32) and read flat, 16 bytes per lane:
// quad i of the stream holds channels 4 (i mod C/4) ... + 3; a lane's quads all belong to one channel
// quad (the grid stride is a multiple of 64 lanes), so it sums in registers, lanes of the same channel
// quad meet by shuffles, one atomic pair per channel and workgroup.
__global__ __launch_bounds__(256) void col_stats_small_kernel(const float* __restrict__ x, long nquads, int C,
                                                              float* __restrict__ sum,
                                                              float* __restrict__ sumsq) {
  const int cq = C >> 2;                                   // channel quads: 1, 2, 4 or 8
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const long step = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < nquads; i += 4 * step) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + u * step];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w;
      q.x = fmaf(v[u].x, v[u].x, q.x); q.y = fmaf(v[u].y, v[u].y, q.y);
      q.z = fmaf(v[u].z, v[u].z, q.z); q.w = fmaf(v[u].w, v[u].w, q.w);
    }
  }
  for (; i < nquads; i += step) {
    const float4 v = x4[i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    q.x = fmaf(v.x, v.x, q.x); q.y = fmaf(v.y, v.y, q.y); q.z = fmaf(v.z, v.z, q.z); q.w = fmaf(v.w, v.w, q.w);
  }
  // lanes l, l + cq, l + 2 cq, ... hold the same channel quad
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    if (o >= cq) {
      s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
      s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
      q.x += __shfl_xor(q.x, o, 64); q.y += __shfl_xor(q.y, o, 64);
      q.z += __shfl_xor(q.z, o, 64); q.w += __shfl_xor(q.w, o, 64);
    }
  }
  __shared__ float4 sh[2][4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < cq) {
    sh[0][wave][lane] = s;
    sh[1][wave][lane] = q;
  }
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const float* f0 = reinterpret_cast<const float*>(&sh[0][0][0]);
    const float* f1 = reinterpret_cast<const float*>(&sh[1][0][0]);
    atomicAdd(&sum[c], (f0[c] + f0[32 + c]) + (f0[64 + c] + f0[96 + c]));
    atomicAdd(&sumsq[c], (f1[c] + f1[32 + c]) + (f1[64 + c] + f1[96 + c]));
  }
}

// the update for the same layouts: flat 16-byte stream, the channel quad's coefficients from LDS
__global__ __launch_bounds__(256) void balancer_apply_small_kernel(
    const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ stats,
    float* __restrict__ stats_next, float n, float min_mean, float max_mean, float min_rms, float max_rms,
    float grad_scale, long nquads, int C, float* __restrict__ out, float act_off) {
  __shared__ __attribute__((aligned(16))) float s_a[32], s_b[32];
  const float inv_n = 1.f / n;
  if (threadIdx.x < C) {                                   // (formulas and clamps of balancer_apply_fused_kernel)
    const int c = threadIdx.x;
    const float mean = stats[c] * inv_n, uvar = stats[1024 + c] * inv_n;
    const float raw_var = uvar - mean * mean;
    const bool live_v = raw_var > 1.0e-20f, live_r = uvar > 1.0e-20f;
    const float var = fmaxf(raw_var, 1.0e-20f);
    const float sd = sqrtf(var);
    const float rms = sqrtf(fmaxf(uvar, 1.0e-20f));
    const float m = mean / sd;
    const float mc = fminf(fmaxf(m, min_mean), max_mean);
    const float s_m = (m > mc) ? 1.f : ((m < mc) ? -1.f : 0.f);
    const float rc = fminf(fmaxf(rms, min_rms), max_rms);
    const float lq = logf(rc / rms);
    const float s_r = (lq > 0.f) ? -1.f : ((lq < 0.f) ? 1.f : 0.f);
    const float a = s_m * inv_n * (live_v ? (1.f / sd + mean * mean / (sd * var)) : 1.f / sd);
    const float b = (live_v ? -s_m * inv_n * mean / (sd * var) : 0.f) +
                    (live_r ? s_r * inv_n / (rms * rms) : 0.f);
    const float lg_rms =
        fmaxf(sqrtf(fmaxf(a * a + 2.f * a * b * mean + b * b * uvar, 0.f)), 1.0e-20f);
    const float coef = grad_scale / lg_rms;
    s_a[c] = a * coef;
    s_b[c] = b * coef;
  }
  if (blockIdx.x == 0 && stats_next != nullptr)
    for (int c = threadIdx.x; c < 2 * 1024; c += 256) stats_next[c] = 0.f;
  __syncthreads();
  const int cq = C >> 2;
  const float4 a4 = *reinterpret_cast<const float4*>(&s_a[4 * (threadIdx.x & (cq - 1))]);
  const float4 b4 = *reinterpret_cast<const float4*>(&s_b[4 * (threadIdx.x & (cq - 1))]);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* o4 = reinterpret_cast<float4*>(out);
  const long step = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nquads; i += 2 * step) {
    const bool two = i + step < nquads;
    float4 xv[2], gv[2];
    xv[0] = x4[i];
    gv[0] = g4[i];
    xv[1] = x4[two ? i + step : i];
    gv[1] = g4[two ? i + step : i];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      float4 gg = gv[u];
      if (act_off >= 0.f) {
        gg.x *= swoosh_d(xv[u].x, act_off);
        gg.y *= swoosh_d(xv[u].y, act_off);
        gg.z *= swoosh_d(xv[u].z, act_off);
        gg.w *= swoosh_d(xv[u].w, act_off);
      }
      o4[i + u * step] = make_float4(gg.x + fabsf(gg.x) * fmaf(b4.x, xv[u].x, a4.x),
                                     gg.y + fabsf(gg.y) * fmaf(b4.y, xv[u].y, a4.y),
                                     gg.z + fabsf(gg.z) * fmaf(b4.z, xv[u].z, a4.z),
                                     gg.w + fabsf(gg.w) * fmaf(b4.w, xv[u].w, a4.w));
    }
  }
}

// Balancer backward, second pass: out = g + |g| * (a'[c] + b'[c] x).  Every workgroup first turns
// the column statistics (sum, sumsq over n rows, from col_stats_kernel) into the coefficients of
// its own copy in LDS (balancer_coef's formulas; C <= 1024), so no coefficient kernel runs; block 0
// also clears `stats_next`, the accumulator the NEXT call's statistics pass will add into (the two
// accumulators alternate, so no fill launch is needed either).
__global__ __launch_bounds__(256) void balancer_apply_fused_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg,
    const float* __restrict__ stats, float* __restrict__ stats_next, float n, float min_mean,
    float max_mean, float min_rms, float max_rms, float grad_scale, long rows, int C,
    float* __restrict__ out, long ldo, float act_off) {
  // act_off >= 0: g is the gradient w.r.t. swoosh(x) and is first taken through the activation
  // (g *= sigmoid(x - act_off) - 0.08): Swoosh backward and the Balancer update in one pass
  __shared__ float s_a[1024], s_b[1024];
  const float inv_n = 1.f / n;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mean = stats[c] * inv_n, uvar = stats[1024 + c] * inv_n;
    const float raw_var = uvar - mean * mean;
    const bool live_v = raw_var > 1.0e-20f, live_r = uvar > 1.0e-20f;
    const float var = fmaxf(raw_var, 1.0e-20f);
    const float sd = sqrtf(var);
    const float rms = sqrtf(fmaxf(uvar, 1.0e-20f));
    const float m = mean / sd;
    const float mc = fminf(fmaxf(m, min_mean), max_mean);
    const float s_m = (m > mc) ? 1.f : ((m < mc) ? -1.f : 0.f);
    const float rc = fminf(fmaxf(rms, min_rms), max_rms);
    const float lq = logf(rc / rms);
    const float s_r = (lq > 0.f) ? -1.f : ((lq < 0.f) ? 1.f : 0.f);
    const float a = s_m * inv_n * (live_v ? (1.f / sd + mean * mean / (sd * var)) : 1.f / sd);
    const float b = (live_v ? -s_m * inv_n * mean / (sd * var) : 0.f) +
                    (live_r ? s_r * inv_n / (rms * rms) : 0.f);
    const float lg_rms =
        fmaxf(sqrtf(fmaxf(a * a + 2.f * a * b * mean + b * b * uvar, 0.f)), 1.0e-20f);
    const float coef = grad_scale / lg_rms;
    s_a[c] = a * coef;
    s_b[c] = b * coef;
  }
  if (blockIdx.x == 0 && stats_next != nullptr)   // the whole accumulator: the previous user may have had more channels
    for (int c = threadIdx.x; c < 2 * 1024; c += 256) stats_next[c] = 0.f;
  __syncthreads();
  // thread = (column within a 64-wide group, row lane): no per-element division, four rows of
  // loads in flight; a workgroup walks all column groups of its rows
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long step = (long)gridDim.x * 4;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + tx;
    if (c >= C) continue;
    const float a = s_a[c], b = s_b[c];
    long r = (long)blockIdx.x * 4 + ty;
    for (; r + 3 * step < rows; r += 4 * step) {
      float gv[4], xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        gv[u] = g[(r + u * step) * ldg + c];
        xv[u] = x[(r + u * step) * ldx + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (act_off >= 0.f) gv[u] *= swoosh_d(xv[u], act_off);
        out[(r + u * step) * ldo + c] = gv[u] + fabsf(gv[u]) * fmaf(b, xv[u], a);
      }
    }
    for (; r < rows; r += step) {
      const float xs = x[r * ldx + c];
      float gv = g[r * ldg + c];
      if (act_off >= 0.f) gv *= swoosh_d(xs, act_off);
      out[r * ldo + c] = gv + fabsf(gv) * fmaf(b, xs, a);
    }
  }
}

inline unsigned grid_for(long n, int per_block) {
  long b = (n + per_block - 1) / per_block;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int s2t_swoosh_fwd(const float* x, float* y, long n, float offset, float constant,
                              void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(swoosh_fwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0,
                     (hipStream_t)stream, x, y, n, offset, constant);
  S2T_CHECK_LAUNCH();
  return 0;
}

// out = a + b (out may be a or b): the residual adds the library GEMM path of the layer executor
// cannot take in its epilogue (zip_layer.hip lt_matmul)
namespace {
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* out, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4* o4 = reinterpret_cast<float4*>(out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 u = a4[i], v = b4[i];
    o4[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = a[i] + b[i];
}
}  // namespace

extern "C" int s2t_add_f32(const float* a, const float* b, float* out, long n, void* stream) {
  if (n <= 0) return 0;
  if (!a || !b || !out || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
                            reinterpret_cast<uintptr_t>(out)) & 15))
    return -1;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, a, b,
                     out, n);
  S2T_CHECK_LAUNCH();
  return 0;
}

// The two alternating accumulators of s2t_balancer_bwd are shared by every caller of the process
// (the Python call sites and the native layer executor): one parity sequence for all of them.
static int g_bal_parity = 0;
extern "C" int s2t_balancer_next_parity(void) {
  g_bal_parity ^= 1;
  return g_bal_parity;
}

extern "C" int s2t_swoosh_bwd(const float* x, const float* g, float* d, long n, float offset,
                              void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(swoosh_bwd_kernel, dim3(grid_for(n, 1024)), dim3(256), 0,
                     (hipStream_t)stream, x, g, d, n, offset);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_biasnorm_fwd(const float* x, const float* bias, const float* log_scale,
                                long rows, int D, float* y, float* scales, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(biasnorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, bias, log_scale, rows, D, y, scales, 0, 0);
  S2T_CHECK_LAUNCH();
  return 0;
}

// the same pass with the output rows written time-major: x (B,T,D) -> y (T,B,D)
extern "C" int s2t_biasnorm_fwd_tb(const float* x, const float* bias, const float* log_scale, int T,
                                   int B, int D, float* y, float* scales, void* stream) {
  if (T <= 0 || B <= 0) return 0;
  const long rows = (long)T * B;
  hipLaunchKernelGGL(biasnorm_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, bias, log_scale, rows, D, y, scales, T, B);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_norm_bypass_fwd(const float* x, const float* bias, const float* log_scale,
                                   const float* orig, const float* bypass_scale, const float* fm, int B,
                                   long rows, int D, float* out, float* scales, void* stream) {
  if (rows <= 0) return 0;
  if (D <= 0 || B <= 0) return -1;
  hipLaunchKernelGGL(norm_bypass_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, bias, log_scale, orig, bypass_scale, fm, B, rows, D, out, scales);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_norm_bypass_bwd(const float* x, const float* bias, const float* scales,
                                   const float* orig, const float* bypass_scale, const float* g,
                                   const float* fm, int B, long rows, int D, float* dx, float* d_orig,
                                   float* d_bypass_scale, float* dbias, float* dls, void* stream) {
  if (rows <= 0) return 0;
  if (D <= 0 || D > 1024 || B <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  static const int form16 = [] { const char* e = getenv("S2T_NB_BWD16"); return e ? atoi(e) : 1; }();
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) |
                       reinterpret_cast<uintptr_t>(orig) | reinterpret_cast<uintptr_t>(bypass_scale) |
                       reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(fm) |
                       reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(d_orig);
  if (form16 && (D & 3) == 0 && (al & 15) == 0) {
    constexpr unsigned cap16 = 512u;
    // workgroups: ~16 rows per wave (four trips of four; 8 below 8 192 rows), at most 512 -- measured at the C3 shapes
    // (tools/bench_atomics.py): more workgroups lengthen the atomics' tail, fewer starve the loads
    const unsigned nb16 = std::min(grid_for(rows, 4 * (rows >= 8192 ? 16 : 8)), cap16);
#define NB_BWD16(Q, RT)                                                                                  \
  hipLaunchKernelGGL((norm_bypass_bwd16_kernel<Q, RT>), dim3(nb16), dim3(256), 0, st, x, bias, scales,   \
                     orig, bypass_scale, g, fm, B, rows, D, dx, d_orig, d_bypass_scale, dbias, dls)
    if (D <= 256) NB_BWD16(1, 4);
    else if (D <= 512) NB_BWD16(2, 2);
    else NB_BWD16(4, 1);
#undef NB_BWD16
    S2T_CHECK_LAUNCH();
    return 0;
  }
  const unsigned nb = std::min(grid_for(rows, 4 * 8), 1024u);
#define NB_BWD(CPL)                                                                                    \
  hipLaunchKernelGGL(norm_bypass_bwd_kernel<CPL>, dim3(nb), dim3(256), 0, st, x, bias, scales, orig,   \
                     bypass_scale, g, fm, B, rows, D, dx, d_orig, d_bypass_scale, dbias, dls)
  if (D <= 64) NB_BWD(1);
  else if (D <= 128) NB_BWD(2);
  else if (D <= 256) NB_BWD(4);
  else if (D <= 512) NB_BWD(8);
  else NB_BWD(16);
#undef NB_BWD
  S2T_CHECK_LAUNCH();
  return 0;
}

static int biasnorm_bwd_launch(const float* x, const float* bias, const float* scales, const float* g,
                               long rows, int D, float* dx, float* dbias, float* dls, int tbT, int tbB,
                               hipStream_t st) {
  // (a 16-byte form -- a 16-lane row of a wave per matrix row -- measured 29 / 25 us against 44 / 27
  // alone and 78-99 us against 40 IN the step: 94-120 registers leave it two waves per SIMD next to
  // the side stream's resident workgroups, where this kernel keeps four.  Removed in round 5.)
  const unsigned nb = std::min(grid_for(rows, 4 * 8), 1024u);
#define BN_BWD(CPL)                                                                                  \
  hipLaunchKernelGGL(biasnorm_bwd_kernel<CPL>, dim3(nb), dim3(256), 0, st, x, bias, scales, g, rows, \
                     D, dx, dbias, dls, tbT, tbB)
  if (D <= 64) BN_BWD(1);
  else if (D <= 128) BN_BWD(2);
  else if (D <= 256) BN_BWD(4);
  else if (D <= 512) BN_BWD(8);
  else if (D <= 1024) BN_BWD(16);
  else return -1;
#undef BN_BWD
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_biasnorm_bwd(const float* x, const float* bias, const float* scales,
                                const float* g, long rows, int D, float* dx, float* dbias,
                                float* dls, void* stream) {
  if (rows <= 0) return 0;
  return biasnorm_bwd_launch(x, bias, scales, g, rows, D, dx, dbias, dls, 0, 0, (hipStream_t)stream);
}

// backward of s2t_biasnorm_fwd_tb: g (T,B,D) time-major as the forward wrote y; x, dx (B,T,D)
extern "C" int s2t_biasnorm_bwd_tb(const float* x, const float* bias, const float* scales, const float* g,
                                   int T, int B, int D, float* dx, float* dbias, float* dls, void* stream) {
  if (T <= 0 || B <= 0) return 0;
  return biasnorm_bwd_launch(x, bias, scales, g, (long)T * B, D, dx, dbias, dls, T, B, (hipStream_t)stream);
}

// the flat forms for few channels: C in {4, 8, 16, 32}, every matrix contiguous (ld == C) and 16-byte aligned
static bool bal_small(int C, const float* x, long ldx, const float* g, long ldg, const float* out, long ldo) {
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return (C == 4 || C == 8 || C == 16 || C == 32) && ldx == C && al(x) && (!g || (ldg == C && al(g))) &&
         (!out || (ldo == C && al(out)));
}
static unsigned bal_small_grid(long nquads) {
  return (unsigned)std::min<long>(2048, std::max<long>(1, (nquads + 256 * 8 - 1) / (256 * 8)));
}

// workspace: two alternating (sum[C], sumsq[C]) accumulators of 2 * BAL_MAXC floats each
constexpr int BAL_MAXC = 1024;
extern "C" long s2t_balancer_bwd_workspace_floats(void) { return 4L * BAL_MAXC; }

extern "C" int s2t_balancer_bwd(const float* x, long ldx, const float* g, long ldg, long rows,
                                int C, float min_mean, float max_mean, float min_rms,
                                float max_rms, float grad_scale, float* out, long ldo,
                                float* workspace, int parity, float act_off, void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  if (C > BAL_MAXC || !workspace) return -1;
  hipStream_t st = (hipStream_t)stream;
  float* cur = workspace + (parity & 1) * 2 * BAL_MAXC;
  float* nxt = workspace + ((parity + 1) & 1) * 2 * BAL_MAXC;
  if (bal_small(C, x, ldx, g, ldg, out, ldo)) {
    const long nq = rows * C / 4;
    hipLaunchKernelGGL(col_stats_small_kernel, dim3(bal_small_grid(nq)), dim3(256), 0, st, x, nq, C, cur,
                       cur + BAL_MAXC);
    S2T_CHECK_LAUNCH();
    hipLaunchKernelGGL(balancer_apply_small_kernel, dim3(bal_small_grid(nq)), dim3(256), 0, st, x, g, cur, nxt,
                       (float)rows, min_mean, max_mean, min_rms, max_rms, grad_scale, nq, C, out, act_off);
    S2T_CHECK_LAUNCH();
    return 0;
  }
  int gy = (C + 63) / 64;
  if (gy > 16) gy = 16;
  long gx = (rows + 4 * 16 - 1) / (4 * 16);
  gx = gx > 256 ? 256 : (gx < 1 ? 1 : gx);
  hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)gx, gy), dim3(64, 4), 0, st, x, rows, C, ldx,
                     cur, cur + BAL_MAXC);
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(balancer_apply_fused_kernel, dim3((unsigned)std::min<long>((rows + 15) / 16, 2048)), dim3(256), 0, st,
                     x, ldx, g, ldg, cur, nxt, (float)rows, min_mean, max_mean, min_rms, max_rms,
                     grad_scale, rows, C, out, ldo, act_off);
  S2T_CHECK_LAUNCH();
  return 0;
}

// The same two passes as separate entry points: the statistics depend on x only, so the caller
// may take them when x is PRODUCED (forward pass, side stream) and run only the update on the
// data-gradient chain.  stats: 2 * 1024 floats (sum | sum of squares), zeroed by the caller.
extern "C" int s2t_balancer_stats(const float* x, long ldx, long rows, int C, float* stats,
                                  void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  if (C > BAL_MAXC || !stats) return -1;
  if (bal_small(C, x, ldx, nullptr, 0, nullptr, 0)) {
    const long nq = rows * C / 4;
    hipLaunchKernelGGL(col_stats_small_kernel, dim3(bal_small_grid(nq)), dim3(256), 0, (hipStream_t)stream, x,
                       nq, C, stats, stats + BAL_MAXC);
    S2T_CHECK_LAUNCH();
    return 0;
  }
  int gy = (C + 63) / 64;
  if (gy > 16) gy = 16;
  long gx = (rows + 4 * 16 - 1) / (4 * 16);
  gx = gx > 256 ? 256 : (gx < 1 ? 1 : gx);
  hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)gx, gy), dim3(64, 4), 0, (hipStream_t)stream, x,
                     rows, C, ldx, stats, stats + BAL_MAXC);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_balancer_apply(const float* x, long ldx, const float* g, long ldg, long rows,
                                  int C, float min_mean, float max_mean, float min_rms,
                                  float max_rms, float grad_scale, float* out, long ldo,
                                  const float* stats, float act_off, void* stream) {
  if (rows <= 0 || C <= 0) return 0;
  if (C > BAL_MAXC || !stats) return -1;
  if (bal_small(C, x, ldx, g, ldg, out, ldo)) {
    const long nq = rows * C / 4;
    hipLaunchKernelGGL(balancer_apply_small_kernel, dim3(bal_small_grid(nq)), dim3(256), 0, (hipStream_t)stream,
                       x, g, stats, (float*)nullptr, (float)rows, min_mean, max_mean, min_rms, max_rms, grad_scale,
                       nq, C, out, act_off);
    S2T_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(balancer_apply_fused_kernel, dim3((unsigned)std::min<long>((rows + 15) / 16, 2048)),
                     dim3(256), 0, (hipStream_t)stream, x, ldx, g, ldg, stats, (float*)nullptr,
                     (float)rows, min_mean, max_mean, min_rms, max_rms, grad_scale, rows, C, out, ldo,
                     act_off);
  S2T_CHECK_LAUNCH();
  return 0;
}
