// Fused log-softmax + smoothed-target loss over a (rows, K) logits matrix, for the BEST-RQ SSL
// heads (reference model/loss/kl_divergence.py:36-76 MaskedKLDivergence and
// model/loss/cross_entropy.py:38-69 MaskedCELoss, called twice per codebook from
// task_factory/ssl_task.py:140-158 on (B, T/4, 8193) logits).
//
// Both losses are  row = C0 - sum_c t_c logp_c  with the target distribution
//   t_c = t_other (c != label),  t_label (c == label),
//   KL:  t_other = eps/(K-1), t_label = 1-eps, C0 = sum_c xlogy(t_c, t_c)
//   CE:  t_other = eps/K,     t_label = 1-eps+eps/K, C0 = 0
// so  row = C0 - [t_other (s sum_c x_c - K lse) + (t_label - t_other)(s x_label - lse)]  and
//   d row / d x_c = s (softmax_c - t_c)                                   (sum_c t_c = 1).
// The reference materialises log_softmax (rows x K), the smoothed-label tensor (rows x K) and
// the element-wise KL (rows x K); here forward is ONE read of the logits (online max / sum per
// row, one workgroup per row) and backward one read + one write.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void smoothed_nll_fwd_kernel(
    const float* __restrict__ logits, const long* __restrict__ labels, long rows, int K,
    float scale, float t_other, float t_label, float c0, float* __restrict__ row_loss,
    float* __restrict__ lse_out) {
  __shared__ float scratch[8];
  const long r = blockIdx.x;
  const float* x = logits + r * K;
  float m = S2T_NEG_INF, se = 0.f, sx = 0.f;     // online max / sum exp / sum of scaled logits
  for (int c = threadIdx.x; c < K; c += 256) {
    const float v = x[c] * scale;
    sx += v;
    if (v > m) {
      se = se * __expf(m - v) + 1.f;
      m = v;
    } else {
      se += __expf(v - m);
    }
  }
  const float gm = block_max(m, scratch);
  se = (m == S2T_NEG_INF) ? 0.f : se * __expf(m - gm);
  const float gs = block_sum(se, scratch);
  const float gx = block_sum(sx, scratch);
  if (threadIdx.x == 0) {
    const float lse = gm + logf(gs);
    const long lab = labels[r];
    const float lp_lab = (lab >= 0 && lab < K) ? x[lab] * scale - lse : 0.f;
    row_loss[r] = c0 - (t_other * (gx - (float)K * lse) + (t_label - t_other) * lp_lab);
    lse_out[r] = lse;
  }
}

// grad[r][c] = w[r] * scale * (exp(scale x - lse) - t_c)
__global__ __launch_bounds__(256) void smoothed_nll_bwd_kernel(
    const float* __restrict__ logits, const long* __restrict__ labels,
    const float* __restrict__ lse, const float* __restrict__ w, long rows, int K, float scale,
    float t_other, float t_label, float* __restrict__ grad) {
  const long r = blockIdx.x;
  const float* x = logits + r * K;
  float* g = grad + r * K;
  const float wr = w[r] * scale, l = lse[r];
  const int lab = (int)labels[r];
  if (wr == 0.f) {
    for (int c = threadIdx.x; c < K; c += 256) g[c] = 0.f;
    return;
  }
  for (int c = threadIdx.x; c < K; c += 256)
    g[c] = wr * (__expf(x[c] * scale - l) - (c == lab ? t_label : t_other));
}

}  // namespace

extern "C" int s2t_smoothed_nll_fwd(const float* logits, const long* labels, long rows, int K,
                                    float scale, float t_other, float t_label, float c0,
                                    float* row_loss, float* lse, void* stream) {
  if (rows <= 0) return 0;
  if (K <= 0) return -1;
  hipLaunchKernelGGL(smoothed_nll_fwd_kernel, dim3((unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, logits, labels, rows, K, scale, t_other, t_label, c0,
                     row_loss, lse);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_smoothed_nll_bwd(const float* logits, const long* labels, const float* lse,
                                    const float* row_weight, long rows, int K, float scale,
                                    float t_other, float t_label, float* grad, void* stream) {
  if (rows <= 0) return 0;
  if (K <= 0) return -1;
  hipLaunchKernelGGL(smoothed_nll_bwd_kernel, dim3((unsigned)rows), dim3(256), 0,
                     (hipStream_t)stream, logits, labels, lse, row_weight, rows, K, scale, t_other,
                     t_label, grad);
  S2T_CHECK_LAUNCH();
  return 0;
}
