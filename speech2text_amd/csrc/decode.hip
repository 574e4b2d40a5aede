// Validation-time greedy decoders for gfx950.
//
// CTC greedy search (reference model/decoding.py:51-82 CtcGreedyDecoding.decode, called per
// utterance by batch_search :27-48): per-frame argmax, collapse repeats, drop blanks -- the
// whole batch in ONE launch, one workgroup per utterance.
// RNN-T greedy search (reference model/decoding.py:196-271 RnntGreedyDecoding.decode) for the
// stateless predictor (model/predictor/stateless_predictor.py:109-125 streaming_step) and a
// joiner without output projection (model/joiner/joiner.py:186-207 streaming_step): the
// reference runs a Python loop of predictor / joiner module calls per lattice move; here one
// workgroup per utterance walks its lattice on the device.  The predictor state is the last
// `ctx` tokens, so the language-side vector lm = pre_proj(linear(conv(embed(state)))) is
// recomputed (two wave-per-row GEMVs) only when a symbol is emitted.
#include "common.h"

namespace {

struct ArgMax {
  float v;
  int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {   // first index wins ties
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ArgMax b;
    b.v = __shfl_xor(a.v, o, 64);
    b.i = __shfl_xor(a.i, o, 64);
    a = better(a, b);
  }
  return a;
}

__global__ __launch_bounds__(256) void ctc_greedy_kernel(const float* __restrict__ logits,
                                                         const long* __restrict__ lengths, int T,
                                                         int V, int blank,
                                                         long* __restrict__ tokens,
                                                         long* __restrict__ out_len) {
  extern __shared__ int ids[];                       // [T] per-frame argmax
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long Tb = lengths[b];
  if (Tb > T) Tb = T;
  if (Tb < 0) Tb = 0;
  const float* x = logits + (long)b * T * V;
  for (int t = wave; t < Tb; t += 4) {
    ArgMax a{S2T_NEG_INF, V};
    for (int c = lane; c < V; c += 64) a = better(a, ArgMax{x[(long)t * V + c], c});
    a = wave_argmax(a);
    if (lane == 0) ids[t] = a.i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long n = 0;
    int prev = blank;
    long* out = tokens + (long)b * T;
    for (int t = 0; t < Tb; ++t) {
      const int p = ids[t];
      if ((p != prev || prev == blank) && p != blank) out[n++] = p;
      prev = p;
    }
    out_len[b] = n;
  }
}

struct RnntGreedyArgs {
  const float* am;        // [B][T][V]  = enc_proj(encoder_out), bias included
  const long* lengths;    // [B]
  const float* emb;       // [num_symbols][E]
  const float* conv_w;    // [E][ctx]   depthwise, no bias
  const float* lin_w;     // [D][E]
  const float* lin_b;     // [D]
  const float* pre_w;     // [V][D]
  const float* pre_b;     // [V]
  int T, V, E, D, ctx, act, max_token_step, max_out, blank;
  long* tokens;           // [B][max_out]
  long* out_len;          // [B]
};

// y[r] = w[r] . x + b[r], one wave per row, x in LDS
__device__ __forceinline__ void gemv_rows(const float* __restrict__ w, const float* __restrict__ bias,
                                          const float* __restrict__ x, int rows, int cols,
                                          float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < rows; r += 4) {
    const float* wr = w + (long)r * cols;
    float acc = 0.f;
    for (int c = lane; c < cols; c += 64) acc = fmaf(wr[c], x[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) y[r] = acc + (bias ? bias[r] : 0.f);
  }
}

__global__ __launch_bounds__(256) void rnnt_greedy_kernel(RnntGreedyArgs a) {
  extern __shared__ float sm[];
  float* e = sm;                       // [E]   conv(embed(state))
  float* hvec = e + a.E;               // [D]
  float* lm = hvec + a.D;              // [V]
  int* state = reinterpret_cast<int*>(lm + a.V);   // [ctx] most recent last
  __shared__ ArgMax s_red[4];
  __shared__ int s_tok;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long Tb = a.lengths[b];
  if (Tb > a.T) Tb = a.T;
  for (int k = tid; k < a.ctx; k += 256) state[k] = a.blank;   // init_state + blank start token
  __syncthreads();
  const float* amb = a.am + (long)b * a.T * a.V;
  long n = 0;
  int t = 0, nts = 0;
  bool need_lm = true;
  while (t < Tb) {
    if (need_lm) {
      for (int c = tid; c < a.E; c += 256) {
        float acc = 0.f;
        for (int k = 0; k < a.ctx; ++k) acc = fmaf(a.conv_w[c * a.ctx + k], a.emb[(long)state[k] * a.E + c], acc);
        e[c] = acc;
      }
      __syncthreads();
      gemv_rows(a.lin_w, a.lin_b, e, a.D, a.E, hvec);
      __syncthreads();
      gemv_rows(a.pre_w, a.pre_b, hvec, a.V, a.D, lm);
      __syncthreads();
      need_lm = false;
    }
    ArgMax best{S2T_NEG_INF, a.V};
    for (int c = tid; c < a.V; c += 256) {
      float v = amb[(long)t * a.V + c] + lm[c];
      v = a.act == 0 ? fmaxf(v, 0.f) : tanhf(v);
      best = better(best, ArgMax{v, c});
    }
    best = wave_argmax(best);
    if (lane == 0) s_red[wave] = best;
    __syncthreads();
    if (tid == 0) s_tok = better(better(s_red[0], s_red[1]), better(s_red[2], s_red[3])).i;
    __syncthreads();
    const int tok = s_tok;
    if (tok == a.blank || nts > a.max_token_step) {
      ++t;
      nts = 0;
    } else {
      ++nts;
      if (tid == 0 && n < a.max_out) a.tokens[(long)b * a.max_out + n] = tok;
      ++n;
      __syncthreads();
      if (tid == 0) {
        for (int k = 0; k + 1 < a.ctx; ++k) state[k] = state[k + 1];
        state[a.ctx - 1] = tok;
      }
      need_lm = true;
      if (n >= a.max_out) break;                     // output buffer full (uniform exit)
    }
    __syncthreads();
  }
  if (tid == 0) a.out_len[b] = n < a.max_out ? n : a.max_out;
}

}  // namespace

extern "C" int s2t_ctc_greedy(const float* logits, const long* lengths, int B, int T, int V,
                              int blank, long* tokens, long* out_len, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || V <= 0 || blank < 0 || blank >= V || T > 16000) return -1;
  hipLaunchKernelGGL(ctc_greedy_kernel, dim3(B), dim3(256), sizeof(int) * T, (hipStream_t)stream,
                     logits, lengths, T, V, blank, tokens, out_len);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_greedy_stateless(const float* am, const long* lengths, const float* emb,
                                         const float* conv_w, const float* lin_w,
                                         const float* lin_b, const float* pre_w, const float* pre_b,
                                         int B, int T, int V, int E, int D, int ctx, int act,
                                         int max_token_step, int max_out, int blank, long* tokens,
                                         long* out_len, void* stream) {
  if (B <= 0) return 0;
  if (T <= 0 || V <= 0 || E <= 0 || D <= 0 || ctx < 1 || ctx > 64 || max_out <= 0 || act < 0 ||
      act > 1)
    return -1;
  const size_t smem = sizeof(float) * ((size_t)E + D + V) + sizeof(int) * ctx;
  if (smem > 60 * 1024) return -1;
  RnntGreedyArgs a{am, lengths, emb, conv_w, lin_w, lin_b, pre_w, pre_b, T, V, E, D, ctx, act,
                   max_token_step, max_out, blank, tokens, out_len};
  hipLaunchKernelGGL(rnnt_greedy_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, a);
  S2T_CHECK_LAUNCH();
  return 0;
}
