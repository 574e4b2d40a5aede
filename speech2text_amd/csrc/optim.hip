// Fused multi-tensor ScaledAdam (+ the trainer's grad-norm clip + zero_grad) over one flat
// parameter / gradient buffer.  Replaces the per-batch-of-tensors torch loop of the reference's
// optimizer/scaled_adam.py:563-736 (_step_one_batch, _size_update, _step, _step_scalar) and
// :408-527 (_get_clipping_scale) with three launches per optimizer step:
//   1. seg_stats      one workgroup per 8192-element chunk of a tensor: sum g^2, sum p*g, sum p^2
//   2. coef           ONE workgroup per param group: per-tensor sums (fixed order: deterministic),
//                     global grad-norm clip factor, ScaledAdam clipping scale (median threshold
//                     re-estimated in-kernel every `period` steps), learned-scale step, per-tensor
//                     Adam coefficients
//   3. apply          one workgroup per chunk: g *= clip, delta/exp_avg_sq/param update, g = 0
// HBM-bound: 2 + 8 fp32 streams over the parameter count (C3: 22.9 M params -> 0.92 GB/step).
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

constexpr int kChunk = 8192;
constexpr int kSegC = 12;   // floats per tensor handed from coef to apply

__global__ __launch_bounds__(256) void seg_stats_kernel(const float* __restrict__ p,
                                                        const float* __restrict__ g,
                                                        const int* __restrict__ chunk_off,
                                                        const int* __restrict__ chunk_len,
                                                        float* __restrict__ partial) {
  __shared__ float red[3][4];
  const int b = blockIdx.x;
  const long off = chunk_off[b];
  const int len = chunk_len[b];              // multiple of 4 (tensors are padded to 16 bytes)
  const float4* p4 = reinterpret_cast<const float4*>(p + off);
  const float4* g4 = reinterpret_cast<const float4*>(g + off);
  float sgg = 0.f, spg = 0.f, spp = 0.f;
  for (int i = threadIdx.x; i < (len >> 2); i += 256) {
    const float4 a = p4[i], c = g4[i];
    sgg += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
    spg += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
    spp += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  sgg = wave_sum(sgg);
  spg = wave_sum(spg);
  spp = wave_sum(spp);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][w] = sgg;
    red[1][w] = spg;
    red[2][w] = spp;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const float* r = red[threadIdx.x];
    partial[3 * (long)b + threadIdx.x] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}

struct CoefArgs {
  const float* partial;        // [nchunks_all][3]
  const int* seg_chunk_begin;  // [nseg_all + 1]
  const int* seg_len;          // [nseg_all] true element counts
  int nchunks_all, seg_lo, seg_hi;
  // hyper-parameters of this param group
  float lr, beta1, beta2, eps, scalar_lr_scale, param_min_rms, param_max_rms, scalar_max;
  float clip_val;              // trainer gradient_clip_val (norm), <= 0: off
  float clipping_scale;        // ScaledAdam clipping_scale, <= 0: off
  int step, size_update_period, clipping_update_period;
  float bc2, bc2_size, beta2c; // 1-beta2^(k+1), 1-(beta2^P)^((k+1)/P), beta2^P
  // state of this group (device)
  float* param_rms;            // [nseg_g]
  float* scale_exp_avg_sq;     // [nseg_g]
  float* scale_grads;          // [P][nseg_g]
  float* model_norms;          // [period]
  float* fstate;               // [0] threshold  [1] last tot_norm  [2] last clip factor
  int* istate;                 // [0] has_threshold [1] num_clipped [2] non-finite median flag
  float* segstat;              // [nseg_all][3] scratch
  float* segc;                 // [nseg_all][kSegC] out
  const float* skip;           // device flag or NULL: != 0 -> this step leaves parameters and state alone
};

__device__ __forceinline__ bool nan_less(float a, float b) {   // torch.sort order: NaN last
  const bool na = a != a, nb = b != b;
  if (na || nb) return !na && nb;
  return a < b;
}

__global__ __launch_bounds__(256) void scaled_adam_coef_kernel(CoefArgs A) {
  __shared__ float scratch[8];
  __shared__ float sorted[1024];
  __shared__ float sh_thr;
  const int tid = threadIdx.x;
  const int ng = A.seg_hi - A.seg_lo;
  if (A.skip != nullptr && A.skip[0] != 0.f) {
    // the data-parallel reducer dropped this step's gradient on every rank (ddp.py: a parameter
    // fired after its bucket had left): no norm is recorded, no moment decays, apply only clears
    // the gradient.  The (p . g) sample of this step counts as 0.
    for (int s = A.seg_lo + tid; s < A.seg_hi; s += 256) {
      A.scale_grads[(long)(A.step % A.size_update_period) * ng + (s - A.seg_lo)] = 0.f;
      A.segc[(long)s * kSegC + 9] = 2.f;
    }
    // the clipping threshold's window gets the previous step's norm again, not a stale entry from
    // one period ago (the step count advances on a dropped step, as on the host path)
    if (tid == 0 && A.clipping_scale > 0.f && A.step > 0) {
      const int period = A.clipping_update_period;
      A.model_norms[A.step % period] = A.model_norms[(A.step - 1) % period];
    }
    return;
  }
  // ---- per-tensor sums (each thread walks its tensors' chunks in order)
  for (int s = A.seg_lo + tid; s < A.seg_hi; s += 256) {
    float a = 0.f, b = 0.f, c = 0.f;
    for (int ch = A.seg_chunk_begin[s]; ch < A.seg_chunk_begin[s + 1]; ++ch) {
      a += A.partial[3 * (long)ch];
      b += A.partial[3 * (long)ch + 1];
      c += A.partial[3 * (long)ch + 2];
    }
    A.segstat[3 * (long)s] = a;
    A.segstat[3 * (long)s + 1] = b;
    A.segstat[3 * (long)s + 2] = c;
  }
  // ---- trainer clip: global gradient norm over EVERY tensor of the store
  float c = 1.f;
  if (A.clip_val > 0.f) {
    float t = 0.f;
    for (int ch = tid; ch < A.nchunks_all; ch += 256) t += A.partial[3 * (long)ch];
    t = block_sum(t, scratch);
    c = fminf(A.clip_val / (sqrtf(t) + 1.0e-6f), 1.f);
  }
  __syncthreads();
  // ---- ScaledAdam clipping scale (reference scaled_adam.py:408-527)
  float ans = 1.f;
  bool sanitize = false;
  const int k = A.step, period = A.clipping_update_period;
  if (A.clipping_scale > 0.f && k > 0) {
    float t = 0.f;
    const float slr2 = A.scalar_lr_scale * A.scalar_lr_scale;
    for (int s = A.seg_lo + tid; s < A.seg_hi; s += 256) {
      const float rms = A.param_rms[s - A.seg_lo];
      const float w = A.seg_len[s] == 1 ? slr2 : rms * rms;
      t += (c * c) * A.segstat[3 * (long)s] * w;
    }
    t = block_sum(t, scratch);
    const float tot_norm = sqrtf(t);
    if (tid == 0) {
      A.model_norms[k % period] = tot_norm;
      A.fstate[1] = tot_norm;
    }
    __syncthreads();
    const bool irregular = (k == 10 || k == 20 || k == 40) && k < period;
    if (k % period == 0 || irregular) {
      for (int i = tid; i < period; i += 256) {
        const float v = A.model_norms[i];
        int rank = 0;
        for (int j = 0; j < period; ++j) {
          const float u = A.model_norms[j];
          rank += (nan_less(u, v) || (!nan_less(v, u) && j < i)) ? 1 : 0;
        }
        sorted[rank] = v;
      }
      __syncthreads();
      if (tid == 0) {
        const int num = irregular ? k : period;
        const int base = irregular ? period - k : 0;
        int mi = (num / 4) * 2;
        if (mi > num - 1) mi = num - 1;
        const float median = sorted[base + mi];
        float thr = A.clipping_scale * median;
        if (irregular) thr *= 2.f;
        A.fstate[0] = thr;
        A.istate[0] = 1;
        A.istate[1] = 0;
        if (!(fabsf(median) <= 3.0e38f)) A.istate[2] = 1;   // host raises (reference :458-460)
      }
      __syncthreads();
    }
    if (tid == 0) sh_thr = A.fstate[0];
    __syncthreads();
    if (A.istate[0]) {
      const float r = sh_thr / (tot_norm + 1.0e-20f);
      ans = (r != r) ? 0.f : fminf(r, 1.f);       // clamp(max=1) keeps NaN; nan_to_num -> 0
      sanitize = true;
      if (tid == 0 && ans < 1.f) A.istate[1] += 1;
    }
  }
  if (tid == 0) A.fstate[2] = c * ans;
  // ---- per-tensor coefficients
  const int P = A.size_update_period;
  const float gm = c * ans;
  for (int s = A.seg_lo + tid; s < A.seg_hi; s += 256) {
    const int q = s - A.seg_lo;
    const int len = A.seg_len[s];
    const bool scalar = len == 1;
    // a zero factor under live clipping means the gradients were non-finite and are zeroed by
    // apply (reference: nan_to_num after the scale): their p.g sum is 0, not 0 * inf
    A.scale_grads[(long)(k % P) * ng + q] =
        (sanitize && gm == 0.f) ? 0.f : gm * A.segstat[3 * (long)s + 1];
    float sstep = 0.f;
    if (k % P == P - 1) {
      const float rms = sqrtf(A.segstat[3 * (long)s + 2] / (float)len);
      A.param_rms[q] = rms;
      if (k > 0) {
        float sum = 0.f, sumsq = 0.f;
        for (int r = 0; r < P; ++r) {
          const float v = A.scale_grads[(long)r * ng + q];
          sum += v;
          sumsq += v * v;
        }
        const float seas = A.scale_exp_avg_sq[q] * A.beta2c + (1.f - A.beta2c) * (sumsq / (float)P);
        A.scale_exp_avg_sq[q] = seas;
        const float denom = sqrtf(seas) + A.eps;
        float st = -(A.lr * A.scalar_lr_scale) * sqrtf(A.bc2_size) * sum / denom;
        if (rms < A.param_min_rms) st = 0.f;
        st = fminf(st, (A.param_max_rms - rms) / rms);
        if (scalar) st = 0.f;
        sstep = st;
      }
    }
    const float rms_now = A.param_rms[q];
    float* o = A.segc + (long)s * kSegC;
    o[0] = gm;
    o[1] = sstep;
    o[2] = scalar ? -A.lr * A.scalar_lr_scale * (1.f - A.beta1)
                  : -A.lr * (1.f - A.beta1) * fmaxf(rms_now, A.param_min_rms);
    o[3] = scalar ? A.bc2 : (A.bc2 < 0.99f ? A.bc2 : 1.f);
    o[4] = scalar ? A.scalar_max : __builtin_huge_valf();
    o[5] = A.beta1;
    o[6] = A.beta2;
    o[7] = A.eps;
    o[8] = sanitize ? 1.f : 0.f;
    o[9] = 1.f;   // tensor belongs to an optimizer group (apply skips tensors with 0)
  }
}

__global__ __launch_bounds__(256) void scaled_adam_apply_kernel(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ delta,
    float* __restrict__ eas, const int* __restrict__ chunk_off, const int* __restrict__ chunk_len,
    const int* __restrict__ chunk_seg, const float* __restrict__ segc, int zero_grad) {
  const int b = blockIdx.x;
  const long off = chunk_off[b];
  const int len = chunk_len[b];
  const float* o = segc + (long)chunk_seg[b] * kSegC;
  if (o[9] != 1.f) {
    // 0: trainable tensor that no optimizer group lists: it is never updated, but its gradient must
    // not pile up across steps (it would keep growing the global clip norm); 2: dropped step
    if (zero_grad) {
      float4* z4 = reinterpret_cast<float4*>(g + off);
      for (int i = threadIdx.x; i < (len >> 2); i += 256) z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const float gm = o[0], sstep = o[1], coef = o[2], bc = o[3], lim = o[4], beta1 = o[5],
              beta2 = o[6], eps = o[7];
  const bool sanitize = o[8] != 0.f;
  float4* p4 = reinterpret_cast<float4*>(p + off);
  float4* g4 = reinterpret_cast<float4*>(g + off);
  float4* d4 = reinterpret_cast<float4*>(delta + off);
  float4* e4 = reinterpret_cast<float4*>(eas + off);
  const float om1 = 1.f - beta1, om2 = 1.f - beta2;
  for (int i = threadIdx.x; i < (len >> 2); i += 256) {
    float4 pv = p4[i], gv = g4[i], dv = d4[i], ev = e4[i];
    float* pp = &pv.x;
    float* gg = &gv.x;
    float* dd = &dv.x;
    float* ee = &ev.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = gg[j] * gm;
      if (sanitize && !(fabsf(gj) <= 3.4028234e38f)) gj = 0.f;   // nan_to_num(nan/inf -> 0)
      float d = dd[j] * beta1 + om1 * (pp[j] * sstep);
      const float e = ee[j] * beta2 + om2 * (gj * gj);
      const float denom = sqrtf(e / bc) + eps;
      d += gj / denom * coef;
      const float pc = fminf(fmaxf(pp[j], -lim), lim);
      pp[j] = pc + d;
      dd[j] = d;
      ee[j] = e;
      gg[j] = zero_grad ? 0.f : gj;
    }
    p4[i] = pv;
    d4[i] = dv;
    e4[i] = ev;
    g4[i] = gv;
  }
}

// ---- Adam / AdamW on the flat buffers (torch.optim.Adam / AdamW semantics, amsgrad off):
// the conformer configs (config/training/conformer_*.yaml `optimizer: type: "AdamW"`).
// clip_coef: ONE workgroup folds the per-chunk sums of g^2 (seg_stats) in a fixed order and
// writes min(1, clip / (norm + 1e-6)); adam_apply: one workgroup per chunk.
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial,
                                                        int nchunks, float clip_val,
                                                        float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nchunks; i += 256) s += (double)partial[3 * (long)i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double norm = sqrt((red[0] + red[1]) + (red[2] + red[3]));
    out[0] = clip_val > 0.f ? (float)fmin(1.0, (double)clip_val / (norm + 1.0e-6)) : 1.f;
    out[1] = (float)norm;
  }
}

struct AdamGroups {
  int n;
  int chunk_hi[S2T_ADAM_MAX_GROUPS];      // group q owns chunks [chunk_hi[q-1], chunk_hi[q])
  float lr[S2T_ADAM_MAX_GROUPS], beta1[S2T_ADAM_MAX_GROUPS], beta2[S2T_ADAM_MAX_GROUPS],
      eps[S2T_ADAM_MAX_GROUPS], wd[S2T_ADAM_MAX_GROUPS], bc1[S2T_ADAM_MAX_GROUPS],
      bc2s[S2T_ADAM_MAX_GROUPS];          // bc2s = sqrt(1 - beta2^step)
  int decoupled[S2T_ADAM_MAX_GROUPS];     // 1: AdamW (p *= 1 - lr wd), 0: Adam (g += wd p)
};

__global__ __launch_bounds__(256) void adam_apply_kernel(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
    const int* __restrict__ chunk_off, const int* __restrict__ chunk_len, AdamGroups G,
    const float* __restrict__ coef, int zero_grad, const float* __restrict__ skip) {
  const int b = blockIdx.x;
  int q = 0;
  while (q < G.n && b >= G.chunk_hi[q]) ++q;
  if (skip != nullptr && skip[0] != 0.f) q = G.n;   // dropped step (ddp.py): only clear the gradient
  const long off = chunk_off[b];
  const int len = chunk_len[b];
  float4* g4 = reinterpret_cast<float4*>(g + off);
  if (q >= G.n) {                              // trainable tensor outside every group
    if (zero_grad)
      for (int i = threadIdx.x; i < (len >> 2); i += 256) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float lr = G.lr[q], b1 = G.beta1[q], b2 = G.beta2[q], eps = G.eps[q], wd = G.wd[q];
  const float step_size = lr / G.bc1[q], inv_bc2s = 1.f / G.bc2s[q];
  const bool dec = G.decoupled[q] != 0;
  const float shrink = dec ? 1.f - lr * wd : 1.f;
  const float c = coef[0];
  float4* p4 = reinterpret_cast<float4*>(p + off);
  float4* m4 = reinterpret_cast<float4*>(m + off);
  float4* v4 = reinterpret_cast<float4*>(v + off);
  for (int i = threadIdx.x; i < (len >> 2); i += 256) {
    float4 pv = p4[i], gv = g4[i], mv = m4[i], vv = v4[i];
    float* pp = &pv.x;
    float* gg = &gv.x;
    float* mm = &mv.x;
    float* ww = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = gg[j] * c;
      float pj = pp[j] * shrink;
      if (!dec) gj += wd * pp[j];
      mm[j] += (1.f - b1) * (gj - mm[j]);                   // lerp
      ww[j] = ww[j] * b2 + (1.f - b2) * gj * gj;
      const float denom = sqrtf(ww[j]) * inv_bc2s + eps;
      pp[j] = pj - step_size * mm[j] / denom;
      gg[j] = zero_grad ? 0.f : gj;
    }
    p4[i] = pv;
    m4[i] = mv;
    v4[i] = vv;
    g4[i] = gv;
  }
}

}  // namespace

extern "C" {

int s2t_clip_coef(const float* partial, int nchunks, float clip_val, float* out, void* stream) {
  if (nchunks <= 0) return -1;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nchunks,
                     clip_val, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_adam_apply(float* p, float* g, float* exp_avg, float* exp_avg_sq, const int* chunk_off,
                   const int* chunk_len, int nchunks, int ngroups, const S2tAdamGroup* groups,
                   const float* coef, int zero_grad, const float* skip, void* stream) {
  if (nchunks <= 0 || ngroups < 0 || ngroups > S2T_ADAM_MAX_GROUPS) return -1;
  AdamGroups G;
  G.n = ngroups;
  for (int q = 0; q < ngroups; ++q) {
    G.chunk_hi[q] = groups[q].chunk_hi;
    G.lr[q] = groups[q].lr;
    G.beta1[q] = groups[q].beta1;
    G.beta2[q] = groups[q].beta2;
    G.eps[q] = groups[q].eps;
    G.wd[q] = groups[q].weight_decay;
    G.bc1[q] = groups[q].bias_correction1;
    G.bc2s[q] = groups[q].sqrt_bias_correction2;
    G.decoupled[q] = groups[q].decoupled;
  }
  hipLaunchKernelGGL(adam_apply_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, p, g,
                     exp_avg, exp_avg_sq, chunk_off, chunk_len, G, coef, zero_grad, skip);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_optim_chunk_elems(void) { return kChunk; }
int s2t_optim_segc_floats(void) { return kSegC; }

int s2t_seg_stats(const float* p, const float* g, const int* chunk_off, const int* chunk_len,
                  int nchunks, float* partial, void* stream) {
  if (nchunks <= 0) return -1;
  hipLaunchKernelGGL(seg_stats_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, p, g,
                     chunk_off, chunk_len, partial);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_scaled_adam_coef(const float* partial, const int* seg_chunk_begin, const int* seg_len,
                         int nchunks_all, int seg_lo, int seg_hi, float lr, float beta1,
                         float beta2, float eps, float scalar_lr_scale, float param_min_rms,
                         float param_max_rms, float scalar_max, float clip_val,
                         float clipping_scale, int step, int size_update_period,
                         int clipping_update_period, float bc2, float bc2_size, float beta2c,
                         float* param_rms, float* scale_exp_avg_sq, float* scale_grads,
                         float* model_norms, float* fstate, int* istate, float* segstat,
                         float* segc, const float* skip, void* stream) {
  if (seg_hi <= seg_lo || clipping_update_period < 1 || clipping_update_period > 1024 ||
      size_update_period < 1)
    return -1;
  CoefArgs A{partial, seg_chunk_begin, seg_len, nchunks_all, seg_lo, seg_hi, lr, beta1, beta2, eps,
             scalar_lr_scale, param_min_rms, param_max_rms, scalar_max, clip_val, clipping_scale,
             step, size_update_period, clipping_update_period, bc2, bc2_size, beta2c, param_rms,
             scale_exp_avg_sq, scale_grads, model_norms, fstate, istate, segstat, segc, skip};
  hipLaunchKernelGGL(scaled_adam_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, A);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_scaled_adam_apply(float* p, float* g, float* delta, float* exp_avg_sq,
                          const int* chunk_off, const int* chunk_len, const int* chunk_seg,
                          int nchunks, const float* segc, int zero_grad, void* stream) {
  if (nchunks <= 0) return -1;
  hipLaunchKernelGGL(scaled_adam_apply_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, p,
                     g, delta, exp_avg_sq, chunk_off, chunk_len, chunk_seg, segc, zero_grad);
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
