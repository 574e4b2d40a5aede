// CTC loss forward + gradient for gfx950, fused with the log-softmax.
//
// Replaces model/loss/ctc_loss.py:35-41 (F.log_softmax -> transpose ->
// nn.CTCLoss(blank, reduction, zero_infinity)).  Three launches:
//   1. per (b,t) row: log-sum-exp of the logits + the lattice's 2U+1 emission
//      log-probabilities gathered into a dense row (the only gather from the logits)
//   2. alpha AND beta recursions concurrently in one workgroup per utterance (half the threads
//      each, states in LDS ping-pong rows, one barrier per time step, next emission row
//      prefetched): T dependent steps of LDS-only work; alpha / beta go to the workspace
//   3. gradient rows: one workgroup per (b,t): occupancies exp(alpha+beta-lp+nll) are
//      scattered to classes in LDS, grad = (softmax - occupancy) * scale  (the product of
//      CTCLoss' gradient and the log_softmax backward)
// Logits stay batch-major (B,T,V); no transposed copy is ever made.
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

__device__ __forceinline__ float log_add3(float a, float b, float c) {
  float m = fmaxf(a, fmaxf(b, c));
  if (m == S2T_NEG_INF) return S2T_NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// One workgroup per (t, b) row: log-sum-exp of the V logits, then the lattice's emission
// log-probabilities gathered ONCE into a dense (T, Smax) row  lp[b][t][s] = x[lab_s] - lse,
// so that the serial recursion below never touches the logits (no dependent gathers).
__global__ __launch_bounds__(64) void ctc_lse_gather_kernel(
    const float* __restrict__ x, const long* __restrict__ targets, long tgt_stride,
    const long* __restrict__ in_len, const long* __restrict__ tgt_len, int T, int V, int Smax,
    int blank, float* __restrict__ lse, float* __restrict__ lp) {
  const int t = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const long row = (long)b * T + t;
  const float* p = x + row * V;
  float m = S2T_NEG_INF;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, p[c]);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < V; c += 64) sum += expf(p[c] - m);
  sum = wave_sum(sum);
  const float l = m + logf(sum);
  if (lane == 0) lse[row] = l;
  if (t >= in_len[b]) return;
  long Ub = tgt_len[b];
  if (Ub < 0) Ub = 0;
  const int S = (int)(2 * Ub + 1);
  float* o = lp + row * Smax;
  const long* tg = targets + (long)b * tgt_stride;
  for (int s = lane; s < S; s += 64) o[s] = p[(s & 1) ? (int)tg[s >> 1] : blank] - l;
}

// One workgroup per utterance; threads 0..127 run the alpha recursion forwards in time while
// threads 128..255 run the beta recursion backwards, one barrier per time step for both.  The
// emission row of the NEXT step is prefetched into registers before the barrier.  alpha and
// beta (log domain) go to the workspace; the gradient kernel forms the occupancies.
// MAXR = lattice states per thread: 128 * MAXR >= 2U+1 (4 serves U <= 255, 16 serves U <= 1023).
template <int MAXR>
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(
    const float* __restrict__ lp, const long* __restrict__ targets, long tgt_stride,
    const long* __restrict__ in_len, const long* __restrict__ tgt_len, int T, int Smax, int blank,
    int zero_infinity, float* __restrict__ alpha_ws, float* __restrict__ beta_ws,
    float* __restrict__ nll_out, float* __restrict__ loss_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LD = Smax + 2;
  float* a0 = reinterpret_cast<float*>(smem_raw);       // alpha ping-pong
  float* a1 = a0 + LD;
  float* b0 = a1 + LD;                                   // beta ping-pong (+2 pad of -inf)
  float* b1 = b0 + LD;
  unsigned char* skip = reinterpret_cast<unsigned char*>(b1 + LD);   // s-2 / s+2 transition ok
  const int b = blockIdx.x, tid = threadIdx.x;
  long Tb = in_len[b];
  long Ub = tgt_len[b];
  if (Tb > T) Tb = T;
  if (Ub < 0) Ub = 0;
  const int S = (int)(2 * Ub + 1);
  const long* tg = targets + (long)b * tgt_stride;
  for (int s = tid; s < 4 * LD; s += 256) a0[s] = S2T_NEG_INF;
  for (int s = tid; s < S; s += 256) {
    // bit 0: alpha may come from s-2; bit 1: beta may come from s+2
    unsigned char f = 0;
    if ((s & 1) && s >= 2 && tg[s >> 1] != tg[(s >> 1) - 1]) f |= 1;
    if ((s & 1) && s + 2 < S && tg[s >> 1] != tg[(s >> 1) + 1]) f |= 2;
    skip[s] = f;
  }
  if (Tb <= 0) {
    if (tid == 0) {
      const float nll = (Ub == 0) ? 0.f : __builtin_huge_valf();
      nll_out[b] = nll;
      loss_out[b] = (zero_infinity && nll == __builtin_huge_valf()) ? 0.f : nll;
    }
    return;
  }
  __syncthreads();
  const bool is_beta = tid >= 128;
  const int h = tid & 127;
  const float* lpb = lp + (long)b * T * Smax;
  float* aw = alpha_ws + (long)b * T * Smax;
  float* bw = beta_ws + (long)b * T * Smax;
  float nxt[MAXR];
  // step 0
  {
    const int t = is_beta ? (int)Tb - 1 : 0;
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int s = h + 128 * r;
      if (s < S) {
        const float e = lpb[(long)t * Smax + s];
        if (!is_beta) {
          const float v = (s < 2) ? e : S2T_NEG_INF;
          a0[s] = v;
          aw[(long)t * Smax + s] = v;
        } else {
          const float v = (s >= S - 2) ? e : S2T_NEG_INF;
          b0[s] = v;
          bw[(long)t * Smax + s] = v;
        }
      }
    }
    if (Tb > 1) {
      const int tn = is_beta ? (int)Tb - 2 : 1;
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        const int s = h + 128 * r;
        nxt[r] = (s < S) ? lpb[(long)tn * Smax + s] : 0.f;
      }
    }
  }
  __syncthreads();
  float* ap = a0;
  float* ac = a1;
  float* bp = b0;
  float* bc = b1;
  for (int k = 1; k < Tb; ++k) {
    const int t = is_beta ? (int)Tb - 1 - k : k;
    float cur[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) cur[r] = nxt[r];
    if (k + 1 < Tb) {                                    // prefetch the next step's emissions
      const int tn = is_beta ? t - 1 : t + 1;
#pragma unroll
      for (int r = 0; r < MAXR; ++r) {
        const int s = h + 128 * r;
        nxt[r] = (s < S) ? lpb[(long)tn * Smax + s] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
      const int s = h + 128 * r;
      if (s < S) {
        if (!is_beta) {
          const float x0 = ap[s];
          const float x1 = s >= 1 ? ap[s - 1] : S2T_NEG_INF;
          const float x2 = (skip[s] & 1) ? ap[s - 2] : S2T_NEG_INF;
          const float v = log_add3(x0, x1, x2) + cur[r];
          ac[s] = v;
          aw[(long)t * Smax + s] = v;
        } else {
          const float x0 = bp[s];
          const float x1 = bp[s + 1];
          const float x2 = (skip[s] & 2) ? bp[s + 2] : S2T_NEG_INF;
          const float v = log_add3(x0, x1, x2) + cur[r];
          bc[s] = v;
          bw[(long)t * Smax + s] = v;
        }
      }
    }
    __syncthreads();
    float* tmp = ap; ap = ac; ac = tmp;
    tmp = bp; bp = bc; bc = tmp;
  }
  if (tid == 0) {
    const float l1 = ap[S - 1];
    const float l2 = S > 1 ? ap[S - 2] : S2T_NEG_INF;
    const float ll = log_add_precise(l1, l2);
    const float nll = -ll;
    nll_out[b] = nll;
    loss_out[b] = (ll == S2T_NEG_INF && zero_infinity) ? 0.f : nll;
  }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse,
    const long* __restrict__ targets, long tgt_stride, const long* __restrict__ in_len,
    const long* __restrict__ tgt_len, const float* __restrict__ nll,
    const float* __restrict__ alpha_ws, const float* __restrict__ beta_ws,
    const float* __restrict__ lp, const float* __restrict__ scale, int T, int V, int Smax,
    int blank, float* __restrict__ grad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* occ = reinterpret_cast<float*>(smem_raw);
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const long row = (long)b * T + t;
  float* g = grad + row * V;
  long Tb = in_len[b];
  long Ub = tgt_len[b];
  if (Ub < 0) Ub = 0;
  const float n = nll[b];
  if (t >= Tb || n == __builtin_huge_valf()) {
    for (int c = tid; c < V; c += blockDim.x) g[c] = 0.f;
    return;
  }
  for (int c = tid; c < V; c += blockDim.x) occ[c] = 0.f;
  __syncthreads();
  const int S = (int)(2 * Ub + 1);
  const float* al = alpha_ws + row * Smax;
  const float* be = beta_ws + row * Smax;
  const float* em = lp + row * Smax;
  float blank_sum = 0.f;
  for (int s = tid; s < S; s += blockDim.x) {
    // occupancy of state s at time t: alpha beta / (emission P(target))
    const float v = expf(al[s] + be[s] - em[s] + n);
    if (s & 1)
      atomicAdd(&occ[(int)targets[(long)b * tgt_stride + (s >> 1)]], v);
    else
      blank_sum += v;
  }
  blank_sum = wave_sum(blank_sum);
  if ((tid & 63) == 0 && blank_sum != 0.f) atomicAdd(&occ[blank], blank_sum);
  __syncthreads();
  const float* x = logits + row * V;
  const float l = lse[row], sc = scale[b];
  for (int c = tid; c < V; c += blockDim.x) g[c] = (expf(x[c] - l) - occ[c]) * sc;
}

}  // namespace

// workspace floats: B*T (lse) + 3 * B*T*Smax (emissions, alpha, beta) + B (nll)
extern "C" long s2t_ctc_workspace_floats(int B, int T, int Umax) {
  return (long)B * T + 3 * (long)B * T * (2 * (long)Umax + 1) + B;
}

extern "C" int s2t_ctc_loss_fwd_bwd(const float* logits, const long* targets, long tgt_stride,
                                    const long* in_len, const long* tgt_len, int B, int T, int V,
                                    int Umax, int blank, int zero_infinity,
                                    const float* grad_scale,  // [B] or null (no grad)
                                    float* workspace, float* loss_per_utt, float* grad_logits,
                                    void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || V <= 0 || Umax < 0 || blank < 0 || blank >= V) return -1;
  if (Umax > S2T_CTC_MAX_LABELS) return -4;   // 2U+1 lattice states must fit 128 x 16 per workgroup
  hipStream_t st = (hipStream_t)stream;
  const int Smax = 2 * Umax + 1;
  const long lat = (long)B * T * Smax;
  float* lse = workspace;
  float* lp = lse + (long)B * T;
  float* alpha = lp + lat;
  float* beta = alpha + lat;
  float* nll = beta + lat;
  hipLaunchKernelGGL(ctc_lse_gather_kernel, dim3(T, B), dim3(64), 0, st, logits, targets,
                     tgt_stride, in_len, tgt_len, T, V, Smax, blank, lse, lp);
  S2T_CHECK_LAUNCH();
  const size_t smem1 = sizeof(float) * 4 * (Smax + 2) + Smax + 16;
#define S2T_CTC_AB(R)                                                                          \
  hipLaunchKernelGGL(ctc_alpha_beta_kernel<R>, dim3(B), dim3(256), smem1, st, lp, targets,       \
                     tgt_stride, in_len, tgt_len, T, Smax, blank, zero_infinity, alpha, beta, nll, \
                     loss_per_utt)
  if (Smax <= 128 * 2) S2T_CTC_AB(2);
  else if (Smax <= 128 * 4) S2T_CTC_AB(4);
  else if (Smax <= 128 * 8) S2T_CTC_AB(8);
  else S2T_CTC_AB(16);
#undef S2T_CTC_AB
  S2T_CHECK_LAUNCH();
  if (grad_logits && grad_scale) {
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(256), sizeof(float) * V, st, logits, lse,
                       targets, tgt_stride, in_len, tgt_len, nll, alpha, beta, lp, grad_scale, T,
                       V, Smax, blank, grad_logits);
    S2T_CHECK_LAUNCH();
  }
  return 0;
}
