// CTC loss forward + gradient for gfx950, fused with the log-softmax.
//
// Replaces model/loss/ctc_loss.py:35-41 (F.log_softmax -> transpose ->
// nn.CTCLoss(blank, reduction, zero_infinity)).  Three launches:
//   1. row log-sum-exp of the logits (one wave per (b,t) row, coalesced)
//   2. alpha/beta recursion: one workgroup per utterance, the 2U+1 lattice
//      states live in LDS (ping-pong rows), T dependent steps; alpha is kept
//      in an HBM workspace and overwritten by the state occupancies gamma
//   3. gradient rows: one workgroup per (b,t): occupancies are scattered to
//      classes in LDS, grad = (softmax - occupancy) * scale  (the product of
//      CTCLoss' gradient and the log_softmax backward)
// Logits stay batch-major (B,T,V); no transposed copy is ever made.
#include "common.h"

namespace {

__device__ __forceinline__ float log_add3(float a, float b, float c) {
  float m = fmaxf(a, fmaxf(b, c));
  if (m == S2T_NEG_INF) return S2T_NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ x, long rows,
                                                      int V, float* __restrict__ lse) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = x + row * V;
  float m = S2T_NEG_INF;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, p[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < V; c += 64) s += expf(p[c] - m);
  s = wave_sum(s);
  if (lane == 0) lse[row] = m + logf(s);
}

// ws: [B][T][Smax] alpha, overwritten with gamma (state occupancy, linear).
__global__ __launch_bounds__(256) void ctc_alpha_beta_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse,
    const long* __restrict__ targets, long tgt_stride, const long* __restrict__ in_len,
    const long* __restrict__ tgt_len, int T, int V, int Smax, int blank, int zero_infinity,
    float* __restrict__ ws, float* __restrict__ nll_out, float* __restrict__ loss_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* rowA = reinterpret_cast<float*>(smem_raw);
  float* rowB = rowA + Smax + 2;
  int* lab = reinterpret_cast<int*>(rowB + Smax + 2);
  __shared__ float s_ll;

  const int b = blockIdx.x, tid = threadIdx.x;
  long Tb = in_len[b];
  long Ub = tgt_len[b];
  if (Tb > T) Tb = T;
  if (Ub < 0) Ub = 0;
  const int S = (int)(2 * Ub + 1);
  for (int s = tid; s < Smax + 2; s += blockDim.x) {
    rowA[s] = S2T_NEG_INF;
    rowB[s] = S2T_NEG_INF;
  }
  for (int s = tid; s < S; s += blockDim.x)
    lab[s] = (s & 1) ? (int)targets[(long)b * tgt_stride + (s >> 1)] : blank;
  __syncthreads();

  const float* lg = logits + (long)b * T * V;
  const float* ls = lse + (long)b * T;
  float* wsb = ws + (long)b * T * Smax;

  if (Tb <= 0) {
    if (tid == 0) {
      float nll = (Ub == 0) ? 0.f : __builtin_huge_valf();
      nll_out[b] = nll;
      loss_out[b] = (zero_infinity && nll == __builtin_huge_valf()) ? 0.f : nll;
    }
    return;
  }

  // ---------------- alpha ----------------
  float* prev = rowA;
  float* cur = rowB;
  for (int s = tid; s < S; s += blockDim.x) {
    float v = (s < 2) ? lg[lab[s]] - ls[0] : S2T_NEG_INF;
    prev[s] = v;
    wsb[s] = v;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float* lgt = lg + (long)t * V;
    const float lset = ls[t];
    for (int s = tid; s < S; s += blockDim.x) {
      const float a = prev[s];
      const float bb = s >= 1 ? prev[s - 1] : S2T_NEG_INF;
      const float c = (s >= 2 && (s & 1) && lab[s] != lab[s - 2]) ? prev[s - 2] : S2T_NEG_INF;
      const float v = log_add3(a, bb, c) + (lgt[lab[s]] - lset);
      cur[s] = v;
      wsb[(long)t * Smax + s] = v;
    }
    __syncthreads();
    float* tmp = prev;
    prev = cur;
    cur = tmp;
  }
  if (tid == 0) {
    const float l1 = prev[S - 1];
    const float l2 = S > 1 ? prev[S - 2] : S2T_NEG_INF;
    s_ll = log_add_precise(l1, l2);
  }
  __syncthreads();
  const float ll = s_ll;
  const float nll = -ll;
  const bool inf = (ll == S2T_NEG_INF);
  if (tid == 0) {
    nll_out[b] = nll;
    loss_out[b] = (inf && zero_infinity) ? 0.f : nll;
  }
  // ---------------- beta + gamma ----------------
  // beta rows are stored with index s (valid 0..S-1); s+1, s+2 read -inf pad.
  float* bprev = rowA;  // beta[t+1]
  float* bcur = rowB;   // beta[t]
  __syncthreads();
  for (int s = tid; s < Smax + 2; s += blockDim.x) {
    bprev[s] = S2T_NEG_INF;
    bcur[s] = S2T_NEG_INF;
  }
  __syncthreads();
  {
    const int t = (int)Tb - 1;
    const float* lgt = lg + (long)t * V;
    const float lset = ls[t];
    for (int s = tid; s < S; s += blockDim.x) {
      const float lp = lgt[lab[s]] - lset;
      const float v = (s >= S - 2) ? lp : S2T_NEG_INF;
      bprev[s] = v;
      const float al = wsb[(long)t * Smax + s];
      wsb[(long)t * Smax + s] = inf ? 0.f : expf(al + v - lp + nll);
    }
  }
  __syncthreads();
  for (int t = (int)Tb - 2; t >= 0; --t) {
    const float* lgt = lg + (long)t * V;
    const float lset = ls[t];
    for (int s = tid; s < S; s += blockDim.x) {
      const float lp = lgt[lab[s]] - lset;
      const float a = bprev[s];
      const float bb = bprev[s + 1];
      const float c =
          (s + 2 < S && (s & 1) && lab[s] != lab[s + 2]) ? bprev[s + 2] : S2T_NEG_INF;
      const float v = log_add3(a, bb, c) + lp;
      bcur[s] = v;
      const float al = wsb[(long)t * Smax + s];
      wsb[(long)t * Smax + s] = inf ? 0.f : expf(al + v - lp + nll);
    }
    __syncthreads();
    float* tmp = bprev;
    bprev = bcur;
    bcur = tmp;
  }
}

__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse,
    const long* __restrict__ targets, long tgt_stride, const long* __restrict__ in_len,
    const long* __restrict__ tgt_len, const float* __restrict__ nll, const float* __restrict__ ws,
    const float* __restrict__ scale, int T, int V, int Smax, int blank, float* __restrict__ grad) {
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
  const float* gam = ws + row * Smax;
  float blank_sum = 0.f;
  for (int s = tid; s < S; s += blockDim.x) {
    const float v = gam[s];
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

// workspace floats needed: B*T (lse) + B*T*Smax (alpha/gamma) + B (nll)
extern "C" long s2t_ctc_workspace_floats(int B, int T, int Umax) {
  return (long)B * T + (long)B * T * (2 * (long)Umax + 1) + B;
}

extern "C" int s2t_ctc_loss_fwd_bwd(const float* logits, const long* targets, long tgt_stride,
                                    const long* in_len, const long* tgt_len, int B, int T, int V,
                                    int Umax, int blank, int zero_infinity,
                                    const float* grad_scale,  // [B] or null (no grad)
                                    float* workspace, float* loss_per_utt, float* grad_logits,
                                    void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || V <= 0 || Umax < 0 || blank < 0 || blank >= V) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int Smax = 2 * Umax + 1;
  float* lse = workspace;
  float* ws = lse + (long)B * T;
  float* nll = ws + (long)B * T * Smax;
  const long rows = (long)B * T;
  hipLaunchKernelGGL(row_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, logits,
                     rows, V, lse);
  S2T_CHECK_LAUNCH();
  const size_t smem1 = sizeof(float) * 2 * (Smax + 2) + sizeof(int) * Smax;
  hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B), dim3(256), smem1, st, logits, lse, targets,
                     tgt_stride, in_len, tgt_len, T, V, Smax, blank, zero_infinity, ws, nll,
                     loss_per_utt);
  S2T_CHECK_LAUNCH();
  if (grad_logits && grad_scale) {
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(T, B), dim3(256), sizeof(float) * V, st, logits, lse,
                       targets, tgt_stride, in_len, tgt_len, nll, ws, grad_scale, T, V, Smax,
                       blank, grad_logits);
    S2T_CHECK_LAUNCH();
  }
  return 0;
}
