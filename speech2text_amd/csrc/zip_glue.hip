// Fused glue of the zipformer layer (the ops between the GEMMs that the reference runs as
// chains of elementwise torch kernels), time-major (T,B,C) rows of C channels:
//   bypass      out = orig + (src - orig) * scale[c]                     zipformer.py:1523-1555
//   nonlin gate xs[b,t,c] = x[t,b,c] * tanh(s[t,b,c])   (batch-major for the W0 @ x bmm)
//   nonlin out  o[t,b,c]  = z[b,t,c] * y[t,b,c]                          zipformer.py:2438-2483
// and their backward passes; per-channel parameter gradients are block partial sums + one
// atomic per channel per workgroup.  All HBM-bound single passes.
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

constexpr int RB = 64;   // rows per workgroup in the column-reducing kernels

__global__ __launch_bounds__(256) void bypass_fwd_kernel(const float* __restrict__ orig,
                                                         const float* __restrict__ src,
                                                         const float* __restrict__ scale, long n4,
                                                         int C4, float* __restrict__ out) {
  const float4* o4 = reinterpret_cast<const float4*>(orig);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  const float4* sc4 = reinterpret_cast<const float4*>(scale);
  float4* y4 = reinterpret_cast<float4*>(out);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 a = o4[i], b = s4[i], k = sc4[i % C4];
    y4[i] = make_float4(fmaf(b.x - a.x, k.x, a.x), fmaf(b.y - a.y, k.y, a.y),
                        fmaf(b.z - a.z, k.z, a.z), fmaf(b.w - a.w, k.w, a.w));
  }
}

// d_src = g * scale, d_orig = g - d_src, d_scale[c] += sum_rows g * (src - orig)
__global__ __launch_bounds__(256) void bypass_bwd_kernel(const float* __restrict__ orig,
                                                         const float* __restrict__ src,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ g, long rows,
                                                         int C, float* __restrict__ d_orig,
                                                         float* __restrict__ d_src,
                                                         float* __restrict__ d_scale) {
  const long r0 = (long)blockIdx.x * RB, r1 = min(rows, r0 + RB);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float k = scale[c];
    float acc = 0.f;
    long r = r0;
    for (; r + 3 < r1; r += 4) {                         // 12 loads in flight per thread
      float gq[4], sq[4], oq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (r + u) * C + c;
        gq[u] = g[i];
        sq[u] = src[i];
        oq[u] = orig[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (r + u) * C + c;
        const float ds = gq[u] * k;
        d_src[i] = ds;
        d_orig[i] = gq[u] - ds;
        acc = fmaf(gq[u], sq[u] - oq[u], acc);
      }
    }
    for (; r < r1; ++r) {
      const long i = r * C + c;
      const float gv = g[i], ds = gv * k;
      d_src[i] = ds;
      d_orig[i] = gv - ds;
      acc = fmaf(gv, src[i] - orig[i], acc);
    }
    atomicAdd(d_scale + c, acc);
  }
}

// as bypass_bwd, d_orig additionally receives acc_in (the gradient already collected for orig)
__global__ __launch_bounds__(256) void bypass_bwd_acc_kernel(const float* __restrict__ orig,
                                                             const float* __restrict__ src,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ g,
                                                             const float* __restrict__ acc_in,
                                                             long rows, int C,
                                                             float* __restrict__ d_orig,
                                                             float* __restrict__ d_src,
                                                             float* __restrict__ d_scale) {
  const long r0 = (long)blockIdx.x * RB, r1 = min(rows, r0 + RB);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float k = scale[c];
    float acc = 0.f;
    long r = r0;
    for (; r + 3 < r1; r += 4) {
      float gq[4], sq[4], oq[4], aq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (r + u) * C + c;
        gq[u] = g[i];
        sq[u] = src[i];
        oq[u] = orig[i];
        aq[u] = acc_in[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (r + u) * C + c;
        const float ds = gq[u] * k;
        d_src[i] = ds;
        d_orig[i] = gq[u] - ds + aq[u];
        acc = fmaf(gq[u], sq[u] - oq[u], acc);
      }
    }
    for (; r < r1; ++r) {
      const long i = r * C + c;
      const float gv = g[i], ds = gv * k;
      d_src[i] = ds;
      d_orig[i] = gv - ds + acc_in[i];
      acc = fmaf(gv, src[i] - orig[i], acc);
    }
    atomicAdd(d_scale + c, acc);
  }
}

// delta[h,b,i] = sum_d dO1 O1 + sum_d dO2 O2 (+ sum_j W[0,b,i,j] dW0[b,i,j] for h == 0):
// the softmax-backward row constants from the deferred consumers; one wave per (h,b,i)
__global__ __launch_bounds__(256) void attn_delta_pairs_kernel(
    const float* __restrict__ W, const float* __restrict__ dW0, const float* __restrict__ dO1,
    const float* __restrict__ O1, int dv1, const float* __restrict__ dO2,
    const float* __restrict__ O2, int dv2, int T, int B, int H, float* __restrict__ delta) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)H * B * T) return;
  const int i = (int)(row % T);
  const int b = (int)((row / T) % B);
  const int h = (int)(row / ((long)T * B));
  float acc = 0.f;
  if (dO1 && lane < dv1) {
    const long o = ((long)i * B + b) * H * dv1 + (long)h * dv1 + lane;
    acc = dO1[o] * O1[o];
  }
  if (dO2 && lane < dv2) {
    const long o = ((long)i * B + b) * H * dv2 + (long)h * dv2 + lane;
    acc = fmaf(dO2[o], O2[o], acc);
  }
  if (h == 0 && dW0) {
    const float* w = W + ((long)b * T + i) * T;
    const float* d = dW0 + ((long)b * T + i) * T;
    for (int j = lane; j < T; j += 64) acc = fmaf(w[j], d[j], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) delta[row] = acc;
}

// bypass with the encoder stack's per-utterance feature mask folded in (zipformer.py:1095-1113:
// `output = mod(output, ...); output = output * feature_mask`): out = (orig + (src - orig) * scale)
// * fm[b, c], rows ordered (t, b).
__global__ __launch_bounds__(256) void bypass_fwd_mask_kernel(const float* __restrict__ orig,
                                                              const float* __restrict__ src,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ fm, int B,
                                                              long n4, int C4,
                                                              float* __restrict__ out) {
  const float4* o4 = reinterpret_cast<const float4*>(orig);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  const float4* sc4 = reinterpret_cast<const float4*>(scale);
  const float4* m4 = reinterpret_cast<const float4*>(fm);
  float4* y4 = reinterpret_cast<float4*>(out);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    const int b = (int)((i / C4) % B);
    const float4 a = o4[i], v = s4[i], k = sc4[c], m = m4[(long)b * C4 + c];
    y4[i] = make_float4(fmaf(v.x - a.x, k.x, a.x) * m.x, fmaf(v.y - a.y, k.y, a.y) * m.y,
                        fmaf(v.z - a.z, k.z, a.z) * m.z, fmaf(v.w - a.w, k.w, a.w) * m.w);
  }
}

// backward of the above: the incoming gradient is multiplied by fm[b, c] on the fly
__global__ __launch_bounds__(256) void bypass_bwd_mask_kernel(const float* __restrict__ orig,
                                                              const float* __restrict__ src,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ g,
                                                              const float* __restrict__ fm, int B,
                                                              long rows, int C,
                                                              float* __restrict__ d_orig,
                                                              float* __restrict__ d_src,
                                                              float* __restrict__ d_scale) {
  const long r0 = (long)blockIdx.x * RB, r1 = min(rows, r0 + RB);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float k = scale[c];
    float acc = 0.f;
    long r = r0;
    for (; r + 3 < r1; r += 4) {
      float gq[4], sq[4], oq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (r + u) * C + c;
        gq[u] = g[i] * fm[((r + u) % B) * C + c];
        sq[u] = src[i];
        oq[u] = orig[i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = (r + u) * C + c;
        const float ds = gq[u] * k;
        d_src[i] = ds;
        d_orig[i] = gq[u] - ds;
        acc = fmaf(gq[u], sq[u] - oq[u], acc);
      }
    }
    for (; r < r1; ++r) {
      const long i = r * C + c;
      const float gv = g[i] * fm[(r % B) * C + c], ds = gv * k;
      d_src[i] = ds;
      d_orig[i] = gv - ds;
      acc = fmaf(gv, src[i] - orig[i], acc);
    }
    atomicAdd(d_scale + c, acc);
  }
}

struct CommitGroup {
  int n;
  S2tCommit it[8];
};

// grad += d for up to 8 small parameters at once, limit_param_value's sign flip applied where
// asked; d is CLEARED afterwards (the accumulators the layer's kernels add into stay clean)
__global__ __launch_bounds__(256) void param_grad_commit_n_kernel(CommitGroup grp) {
  for (int q = 0; q < grp.n; ++q) {
    const S2tCommit& it = grp.it[q];
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < it.n; e += (long)gridDim.x * 256) {
      float v = it.d[e];
      it.d[e] = 0.f;
      if (it.limit) {
        const float xv = it.x[e];
        if (v > 0.f && xv < it.lo) v = -v;
        if (v < 0.f && xv > it.hi) v = -v;
      }
      it.grad[e] += v;
    }
  }
}

// SimpleDownsample (zipformer.py:1653-1695): out[tt,b,c] = sum_k w[k] src[min(tt*ds + k, T-1), b, c]
// (the reference pads by repeating the last frame), w = softmax(bias) computed by the caller.
// rowlen = B*C; one thread per output element, coalesced over (b,c).
// btC > 0: the output is written batch-major, out[b][tt][c] with C = btC (the encoder's final
// x.transpose(0, 1), zipformer.py:199, rides in this pass), dT = frames of the output
__global__ __launch_bounds__(256) void downsample_fwd_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ w, int ds,
                                                             int T, long rowlen, long n_out,
                                                             float* __restrict__ out, int btC, int dT) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (long)gridDim.x * 256) {
    const long tt = i / rowlen, e = i - tt * rowlen;
    float acc = 0.f;
    for (int k = 0; k < ds; ++k) {
      const long t = min((long)tt * ds + k, (long)T - 1);
      acc = fmaf(w[k], src[t * rowlen + e], acc);
    }
    if (btC > 0) {
      const long b = e / btC, c = e - b * btC;
      out[(b * dT + tt) * btC + c] = acc;
    } else {
      out[i] = acc;
    }
  }
}

// backward: d_src[t] = sum over the (tt,k) that read frame t of w[k] g[tt]  (the last frame also
// collects the padded taps); dw[k] += sum g[tt] * src[frame(tt,k)]  (block partials + atomics)
// any row length / alignment (one element per thread and trip)
__global__ __launch_bounds__(256) void downsample_bwd_any_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ g, int ds,
                                                             int T, int dT, long rowlen,
                                                             float* __restrict__ d_src,
                                                             float* __restrict__ dw, int btC) {
  // btC > 0: g is batch-major, g[b][tt][c] with C = btC (as downsample_fwd_kernel wrote the output)
  __shared__ float s_dw[8][4];
  float pw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) pw[k] = 0.f;
  const long n = (long)dT * rowlen;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long tt = i / rowlen, e = i - tt * rowlen;
    const float gv = btC > 0 ? g[((e / btC) * dT + tt) * btC + (e % btC)] : g[i];
    float last = 0.f;                                    // contributions landing on frame T-1
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < ds) {
        const long t = (long)tt * ds + k;
        const long tc = min(t, (long)T - 1);
        pw[k] = fmaf(gv, src[tc * rowlen + e], pw[k]);
        if (t < T - 1) d_src[t * rowlen + e] = w[k] * gv;
        else last = fmaf(w[k], gv, last);
      }
    }
    if ((long)tt * ds + ds - 1 >= T - 1) d_src[(long)(T - 1) * rowlen + e] = last;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = wave_sum(pw[k]);
    if ((threadIdx.x & 63) == 0) s_dw[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < ds)
    atomicAdd(dw + threadIdx.x, (s_dw[threadIdx.x][0] + s_dw[threadIdx.x][1]) +
                                    (s_dw[threadIdx.x][2] + s_dw[threadIdx.x][3]));
}

__global__ __launch_bounds__(256) void downsample_bwd_kernel(const float* __restrict__ src,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ g, int ds,
                                                             int T, int dT, long rowlen,
                                                             float* __restrict__ d_src,
                                                             float* __restrict__ dw, int btC4) {
  // workgroup = (output frame tt, slice of the row): float4 everywhere, no index division, the
  // ds source rows of a frame are loaded as one batch.  btC4 > 0: g is batch-major (C / 4 = btC4)
  __shared__ float s_dw[8][4];
  float pw[8], wk[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pw[k] = 0.f;
    wk[k] = k < ds ? w[k] : 0.f;
  }
  const long n4 = rowlen >> 2;
  for (int tt = blockIdx.x; tt < dT; tt += gridDim.x) {
    const float4* g4 = reinterpret_cast<const float4*>(g + (btC4 > 0 ? 0 : (long)tt * rowlen));
    for (long e = (long)blockIdx.y * 256 + threadIdx.x; e < n4; e += (long)gridDim.y * 256) {
      const float4 gv = btC4 > 0 ? g4[((e / btC4) * dT + tt) * btC4 + (e % btC4)] : g4[e];
      float4 sv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < ds) {
          const long tc = min((long)tt * ds + k, (long)T - 1);
          sv[k] = reinterpret_cast<const float4*>(src + tc * rowlen)[e];
        }
      float4 last = make_float4(0.f, 0.f, 0.f, 0.f);     // contributions landing on frame T-1
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < ds) {
          const long t = (long)tt * ds + k;
          pw[k] += gv.x * sv[k].x + gv.y * sv[k].y + gv.z * sv[k].z + gv.w * sv[k].w;
          const float4 o = make_float4(wk[k] * gv.x, wk[k] * gv.y, wk[k] * gv.z, wk[k] * gv.w);
          if (t < T - 1) {
            reinterpret_cast<float4*>(d_src + t * rowlen)[e] = o;
          } else {
            last.x += o.x; last.y += o.y; last.z += o.z; last.w += o.w;
          }
        }
      if ((long)tt * ds + ds - 1 >= T - 1)
        reinterpret_cast<float4*>(d_src + (long)(T - 1) * rowlen)[e] = last;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = wave_sum(pw[k]);
    if ((threadIdx.x & 63) == 0) s_dw[k][threadIdx.x >> 6] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < ds)
    atomicAdd(dw + threadIdx.x, (s_dw[threadIdx.x][0] + s_dw[threadIdx.x][1]) +
                                    (s_dw[threadIdx.x][2] + s_dw[threadIdx.x][3]));
}

// SimpleUpsample + out_combiner bypass of a downsampled stack (zipformer.py:1253-1283,
// 1698-1719, 1523-1555): out[t] = orig[t] + (src[t / up] - orig[t]) * scale[c] -- the upsampled
// tensor is never materialised.  rows of C channels; Bn = rows per frame (batch).
__global__ __launch_bounds__(256) void bypass_up_fwd_kernel(const float* __restrict__ orig,
                                                            const float* __restrict__ src,
                                                            const float* __restrict__ scale,
                                                            int up, int Bn, long n4, int C4,
                                                            float* __restrict__ out) {
  const float4* o4 = reinterpret_cast<const float4*>(orig);
  const float4* s4 = reinterpret_cast<const float4*>(src);
  const float4* sc4 = reinterpret_cast<const float4*>(scale);
  float4* y4 = reinterpret_cast<float4*>(out);
  const long frame4 = (long)Bn * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long t = i / frame4, e = i - t * frame4;
    const float4 a = o4[i], v = s4[(t / up) * frame4 + e], k = sc4[e % C4];
    y4[i] = make_float4(fmaf(v.x - a.x, k.x, a.x), fmaf(v.y - a.y, k.y, a.y),
                        fmaf(v.z - a.z, k.z, a.z), fmaf(v.w - a.w, k.w, a.w));
  }
}

// backward: block = one source frame tt x a slab of its (b) rows; thread = channel
__global__ __launch_bounds__(256) void bypass_up_bwd_kernel(const float* __restrict__ orig,
                                                            const float* __restrict__ src,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ g, int up,
                                                            int T, int Bn, int C,
                                                            float* __restrict__ d_orig,
                                                            float* __restrict__ d_src,
                                                            float* __restrict__ d_scale) {
  const int tt = blockIdx.x;
  const int b0 = blockIdx.y * 16, b1 = min(Bn, b0 + 16);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float k = scale[c];
    float acc = 0.f;
    for (int b = b0; b < b1; ++b) {
      const float sv = src[((long)tt * Bn + b) * C + c];
      float ds_ = 0.f;
      for (int u = 0; u < up; ++u) {
        const long t = (long)tt * up + u;
        if (t >= T) break;
        const long i = (t * Bn + b) * C + c;
        const float gv = g[i];
        d_orig[i] = gv * (1.f - k);
        ds_ = fmaf(gv, k, ds_);
        acc = fmaf(gv, sv - orig[i], acc);
      }
      d_src[((long)tt * Bn + b) * C + c] = ds_;
    }
    atomicAdd(d_scale + c, acc);
  }
}

// 16-byte form (C % 4 == 0, up = UP): an item = 4 channels of one (source frame, utterance) pair -- its UP
// gradient / orig pieces and the source piece are requested together (2 UP + 1 loads of 16 bytes in flight
// per lane; the scalar form above has one 4-byte load chain per channel: 1.2 TB/s at the C3 shapes), the
// grid's stride is a multiple of C / 4 so that a lane keeps its channels and d_scale stays in registers.
template <int UP, int IU>
__global__ __launch_bounds__(256) void bypass_up_bwd4_kernel(const float4* __restrict__ o4,
                                                             const float4* __restrict__ s4,
                                                             const float4* __restrict__ sc4,
                                                             const float4* __restrict__ g4, int T, long F,
                                                             int C4, long n, float4* __restrict__ do4,
                                                             float4* __restrict__ ds4,
                                                             float* __restrict__ d_scale) {
  __shared__ float s_acc[1024];
  for (int i = threadIdx.x; i < 4 * C4; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  const long stride = (long)gridDim.x * 256;
  const long w0 = (long)blockIdx.x * 256 + threadIdx.x;
  const int c4 = (int)(w0 % C4);
  const float4 k = sc4[c4];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long wb = w0; wb < n; wb += stride * IU) {
    float4 sv[IU], gv[IU][UP], ov[IU][UP];
#pragma unroll
    for (int j = 0; j < IU; ++j) {                 // every request of IU items first
      const long w = wb + j * stride < n ? wb + j * stride : wb;
      const long tt = w / F, e = w - tt * F;
      sv[j] = s4[w];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const long t = tt * UP + u;
        const long i = (t < T ? t : tt * UP) * F + e;
        gv[j][u] = g4[i];
        ov[j][u] = o4[i];
      }
    }
#pragma unroll
    for (int j = 0; j < IU; ++j) {
      const long w = wb + j * stride;
      if (w < n) {
        const long tt = w / F, e = w - tt * F;
        float4 ds = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < UP; ++u) {
          const long t = tt * UP + u;
          if (t < T) {
            const float4 gq = gv[j][u], oq = ov[j][u], sq = sv[j];
            do4[t * F + e] =
                make_float4(gq.x * (1.f - k.x), gq.y * (1.f - k.y), gq.z * (1.f - k.z), gq.w * (1.f - k.w));
            ds = make_float4(fmaf(gq.x, k.x, ds.x), fmaf(gq.y, k.y, ds.y), fmaf(gq.z, k.z, ds.z),
                             fmaf(gq.w, k.w, ds.w));
            acc = make_float4(fmaf(gq.x, sq.x - oq.x, acc.x), fmaf(gq.y, sq.y - oq.y, acc.y),
                              fmaf(gq.z, sq.z - oq.z, acc.z), fmaf(gq.w, sq.w - oq.w, acc.w));
          }
        }
        ds4[w] = ds;
      }
    }
  }
  // d_scale: through LDS, then ONE atomic per channel and workgroup -- and few workgroups: atomics on one
  // address serialise (the grid is capped in the launcher: 2 000 workgroups spent 30 us of a 59 us launch here)
  atomicAdd(&s_acc[4 * c4 + 0], acc.x);
  atomicAdd(&s_acc[4 * c4 + 1], acc.y);
  atomicAdd(&s_acc[4 * c4 + 2], acc.z);
  atomicAdd(&s_acc[4 * c4 + 3], acc.w);
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C4; i += 256) {
    const float v = s_acc[i];
    if (v != 0.f) atomicAdd(d_scale + i, v);
  }
}

// u (T,B,3C) = [s | x | y]  ->  xs (B,T,C) = x * tanh(s)
__global__ __launch_bounds__(256) void nonlin_gate_fwd_kernel(const float* __restrict__ u, int T,
                                                              int B, int C,
                                                              float* __restrict__ xs) {
  const long n = (long)T * B * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long tb = i / C;
    const int b = (int)(tb % B);
    const long t = tb / B;
    const float* row = u + tb * 3 * C;
    xs[((long)b * T + t) * C + c] = row[C + c] * tanhf(row[c]);
  }
}

// z (B,T,C), u (T,B,3C) -> o (T,B,C) = z * y
__global__ __launch_bounds__(256) void nonlin_out_fwd_kernel(const float* __restrict__ z,
                                                             const float* __restrict__ u, int T,
                                                             int B, int C, float* __restrict__ o) {
  const long n = (long)T * B * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long tb = i / C;
    const int b = (int)(tb % B);
    const long t = tb / B;
    o[i] = z[((long)b * T + t) * C + c] * u[tb * 3 * C + 2 * C + c];
  }
}

// g (T,B,C) -> dz (B,T,C) = g * y ;  du[.., 2C:3C] = dy = g * z
__global__ __launch_bounds__(256) void nonlin_out_bwd_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ z,
                                                             const float* __restrict__ u, int T,
                                                             int B, int C, float* __restrict__ dz,
                                                             float* __restrict__ du) {
  const long n = (long)T * B * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long tb = i / C;
    const int b = (int)(tb % B);
    const long t = tb / B;
    const long zi = ((long)b * T + t) * C + c;
    const float gv = g[i];
    dz[zi] = gv * u[tb * 3 * C + 2 * C + c];
    du[tb * 3 * C + 2 * C + c] = gv * z[zi];
  }
}

// dxs (B,T,C) -> du[.., 0:C] = ds = dxs * x * (1 - tanh(s)^2) ; du[.., C:2C] = dx = dxs * tanh(s)
__global__ __launch_bounds__(256) void nonlin_gate_bwd_kernel(const float* __restrict__ dxs,
                                                              const float* __restrict__ u, int T,
                                                              int B, int C,
                                                              float* __restrict__ du) {
  const long n = (long)T * B * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    const long tb = i / C;
    const int b = (int)(tb % B);
    const long t = tb / B;
    const float d = dxs[((long)b * T + t) * C + c];
    const float* row = u + tb * 3 * C;
    const float th = tanhf(row[c]);
    du[tb * 3 * C + c] = d * row[C + c] * (1.f - th * th);
    du[tb * 3 * C + C + c] = d * th;
  }
}

inline unsigned grid1(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int s2t_bypass_fwd(const float* orig, const float* src, const float* scale, long rows,
                              int C, float* out, void* stream) {
  if (rows <= 0) return 0;
  if (C <= 0 || (C & 3)) return -1;
  const long n4 = rows * C / 4;
  hipLaunchKernelGGL(bypass_fwd_kernel, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream, orig,
                     src, scale, n4, C / 4, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_bypass_bwd(const float* orig, const float* src, const float* scale,
                              const float* g, long rows, int C, float* d_orig, float* d_src,
                              float* d_scale, void* stream) {
  if (rows <= 0) return 0;
  if (C <= 0) return -1;
  hipLaunchKernelGGL(bypass_bwd_kernel, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), 0,
                     (hipStream_t)stream, orig, src, scale, g, rows, C, d_orig, d_src, d_scale);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_bypass_bwd_acc(const float* orig, const float* src, const float* scale,
                                  const float* g, const float* acc_in, long rows, int C,
                                  float* d_orig, float* d_src, float* d_scale, void* stream) {
  if (rows <= 0) return 0;
  if (C <= 0 || !acc_in) return -1;
  hipLaunchKernelGGL(bypass_bwd_acc_kernel, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), 0,
                     (hipStream_t)stream, orig, src, scale, g, acc_in, rows, C, d_orig, d_src,
                     d_scale);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_attn_delta_pairs(const float* W, const float* dW0, const float* dO1,
                                    const float* O1, int dv1, const float* dO2, const float* O2,
                                    int dv2, int T, int B, int H, float* delta, void* stream) {
  if (T <= 0 || B <= 0 || H <= 0) return 0;
  if (dv1 > 64 || dv2 > 64 || dv1 < 0 || dv2 < 0) return -1;
  const long rows = (long)H * B * T;
  hipLaunchKernelGGL(attn_delta_pairs_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, W, dW0, dO1, O1, dv1, dO2, O2, dv2, T, B, H, delta);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_bypass_fwd_mask(const float* orig, const float* src, const float* scale,
                                   const float* fm, int B, long rows, int C, float* out,
                                   void* stream) {
  if (rows <= 0) return 0;
  if (C <= 0 || (C & 3) || B <= 0 || !fm) return -1;
  const long n4 = rows * C / 4;
  hipLaunchKernelGGL(bypass_fwd_mask_kernel, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream,
                     orig, src, scale, fm, B, n4, C / 4, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_bypass_bwd_mask(const float* orig, const float* src, const float* scale,
                                   const float* g, const float* fm, int B, long rows, int C,
                                   float* d_orig, float* d_src, float* d_scale, void* stream) {
  if (rows <= 0) return 0;
  if (C <= 0 || B <= 0 || !fm) return -1;
  hipLaunchKernelGGL(bypass_bwd_mask_kernel, dim3((unsigned)((rows + RB - 1) / RB)), dim3(256), 0,
                     (hipStream_t)stream, orig, src, scale, g, fm, B, rows, C, d_orig, d_src,
                     d_scale);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_param_grad_commit_n(int n, const S2tCommit* items, void* stream) {
  if (n <= 0) return 0;
  if (n > 8 || !items) return -1;
  CommitGroup grp;
  grp.n = n;
  long longest = 1;
  for (int i = 0; i < n; ++i) {
    grp.it[i] = items[i];
    longest = items[i].n > longest ? items[i].n : longest;
  }
  hipLaunchKernelGGL(param_grad_commit_n_kernel, dim3((unsigned)((longest + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, grp);
  S2T_CHECK_LAUNCH();
  return 0;
}

static int downsample_fwd_impl(const float* src, const float* w, int ds, int T, int B, int C, float* out,
                               int bt, void* stream) {
  if (T <= 0 || B <= 0 || C <= 0) return 0;
  if (ds < 1 || ds > 8) return -1;
  const int dT = (T + ds - 1) / ds;
  const long rowlen = (long)B * C, n_out = (long)dT * rowlen;
  hipLaunchKernelGGL(downsample_fwd_kernel, dim3(grid1(n_out)), dim3(256), 0, (hipStream_t)stream,
                     src, w, ds, T, rowlen, n_out, out, bt ? C : 0, dT);
  S2T_CHECK_LAUNCH();
  return 0;
}
static int downsample_bwd_impl(const float* src, const float* w, const float* g, int ds, int T, int B, int C,
                               float* d_src, float* dw, int bt, void* stream);

extern "C" int s2t_downsample_fwd(const float* src, const float* w, int ds, int T, int B, int C,
                                  float* out, void* stream) {
  return downsample_fwd_impl(src, w, ds, T, B, C, out, 0, stream);
}
// out (B, dT, C) batch-major / g (B, dT, C): the transposition the encoder applies to its output
// (model/encoder/zipformer.py:199) done by the pass that writes / reads it
extern "C" int s2t_downsample_fwd_bt(const float* src, const float* w, int ds, int T, int B, int C,
                                     float* out, void* stream) {
  return downsample_fwd_impl(src, w, ds, T, B, C, out, 1, stream);
}
extern "C" int s2t_downsample_bwd(const float* src, const float* w, const float* g, int ds, int T,
                                  int B, int C, float* d_src, float* dw, void* stream) {
  return downsample_bwd_impl(src, w, g, ds, T, B, C, d_src, dw, 0, stream);
}
extern "C" int s2t_downsample_bwd_bt(const float* src, const float* w, const float* g, int ds, int T,
                                     int B, int C, float* d_src, float* dw, void* stream) {
  return downsample_bwd_impl(src, w, g, ds, T, B, C, d_src, dw, 1, stream);
}

static int downsample_bwd_impl(const float* src, const float* w, const float* g, int ds, int T, int B, int C,
                               float* d_src, float* dw, int bt, void* stream) {
  if (T <= 0 || B <= 0 || C <= 0) return 0;
  if (ds < 1 || ds > 8) return -1;
  const int dT = (T + ds - 1) / ds;
  const long rowlen = (long)B * C;
  if ((rowlen & 3) || (C & 3) || ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(g) |
                                  reinterpret_cast<uintptr_t>(d_src)) & 15)) {
    long blocks = ((long)dT * rowlen + 1023) / 1024;
    blocks = blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(downsample_bwd_any_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, src, w, g, ds, T, dT, rowlen, d_src, dw, bt ? C : 0);
    S2T_CHECK_LAUNCH();
    return 0;
  }
  // (frames, row slices): at most `cap` workgroups, each finishing with ds atomics on the SAME words --
  // those serialise (~25 ns each): with ~1000 workgroups the launch took 26 us whatever its size
  constexpr long cap = 512;          // (1 024: 27-29 us, 512: 23-25, 256: 24-27, 128: 30-41 at the C3 shapes)
  int gy = (int)std::min<long>((rowlen / 4 + 255) / 256, std::max<long>(1, cap / dT));
  if (gy < 1) gy = 1;
  const int gx = (int)std::min<long>(dT, std::max<long>(1, cap / gy));
  hipLaunchKernelGGL(downsample_bwd_kernel, dim3(gx, gy), dim3(256), 0,
                     (hipStream_t)stream, src, w, g, ds, T, dT, rowlen, d_src, dw, bt ? C / 4 : 0);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_bypass_up_fwd(const float* orig, const float* src, const float* scale, int up,
                                 int T, int B, int C, float* out, void* stream) {
  if (T <= 0 || B <= 0) return 0;
  if (C <= 0 || (C & 3) || up < 1) return -1;
  const long n4 = (long)T * B * C / 4;
  hipLaunchKernelGGL(bypass_up_fwd_kernel, dim3(grid1(n4)), dim3(256), 0, (hipStream_t)stream,
                     orig, src, scale, up, B, n4, C / 4, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_bypass_up_bwd(const float* orig, const float* src, const float* scale,
                                 const float* g, int up, int T, int B, int C, float* d_orig,
                                 float* d_src, float* d_scale, void* stream) {
  if (T <= 0 || B <= 0) return 0;
  if (C <= 0 || up < 1) return -1;
  const int Ts = (T + up - 1) / up;
  const uintptr_t al = reinterpret_cast<uintptr_t>(orig) | reinterpret_cast<uintptr_t>(src) |
                       reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(g) |
                       reinterpret_cast<uintptr_t>(d_orig) | reinterpret_cast<uintptr_t>(d_src);
  static const bool form16 = [] { const char* e = getenv("S2T_BYPASS_UP_BWD16"); return !e || e[0] != '0'; }();
  if (form16 && (C & 3) == 0 && C <= 1024 && (al & 15) == 0 && (up == 2 || up == 4 || up == 8)) {
    const int C4 = C / 4;
    const long F = (long)B * C4, n = (long)Ts * F;
    // the grid's stride (256 x blocks) must be a multiple of C4: blocks in multiples of C4 / gcd(256, C4)
    int gc = C4, r = 256;
    while (r) { const int t = gc % r; gc = r; r = t; }
    const long m = C4 / gc;
    constexpr long cap = 256;        // (128: 28-33 us, 192 / 256 / 384: 25-30, 512: 30-34, 1 024: 38-40, 2 048: 38-60)
    long nb = (n + 255) / 256;
    if (nb > cap) nb = cap;
    nb = (nb + m - 1) / m * m;
#define BUP4(UP, IU)                                                                                          \
  hipLaunchKernelGGL((bypass_up_bwd4_kernel<UP, IU>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,  \
                     reinterpret_cast<const float4*>(orig), reinterpret_cast<const float4*>(src),             \
                     reinterpret_cast<const float4*>(scale), reinterpret_cast<const float4*>(g), T, F, C4, n, \
                     reinterpret_cast<float4*>(d_orig), reinterpret_cast<float4*>(d_src), d_scale)
    if (up == 2) BUP4(2, 4);
    else if (up == 4) BUP4(4, 2);
    else BUP4(8, 1);
#undef BUP4
    S2T_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(bypass_up_bwd_kernel, dim3(Ts, (B + 15) / 16), dim3(256), 0,
                     (hipStream_t)stream, orig, src, scale, g, up, T, B, C, d_orig, d_src, d_scale);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_nonlin_gate_fwd(const float* u, int T, int B, int C, float* xs, void* stream) {
  if (T <= 0 || B <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(nonlin_gate_fwd_kernel, dim3(grid1((long)T * B * C)), dim3(256), 0,
                     (hipStream_t)stream, u, T, B, C, xs);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_nonlin_out_fwd(const float* z, const float* u, int T, int B, int C, float* o,
                                  void* stream) {
  if (T <= 0 || B <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(nonlin_out_fwd_kernel, dim3(grid1((long)T * B * C)), dim3(256), 0,
                     (hipStream_t)stream, z, u, T, B, C, o);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_nonlin_out_bwd(const float* g, const float* z, const float* u, int T, int B,
                                  int C, float* dz, float* du, void* stream) {
  if (T <= 0 || B <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(nonlin_out_bwd_kernel, dim3(grid1((long)T * B * C)), dim3(256), 0,
                     (hipStream_t)stream, g, z, u, T, B, C, dz, du);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_nonlin_gate_bwd(const float* dxs, const float* u, int T, int B, int C,
                                   float* du, void* stream) {
  if (T <= 0 || B <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(nonlin_gate_bwd_kernel, dim3(grid1((long)T * B * C)), dim3(256), 0,
                     (hipStream_t)stream, dxs, u, T, B, C, du);
  S2T_CHECK_LAUNCH();
  return 0;
}
