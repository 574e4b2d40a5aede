// Layer-norm LSTM layer of the RNN-T predictor, whole sequence per launch, for gfx950.
//
// Replaces the per-token Python loop of torchaudio.models.rnnt._CustomLSTM as used by the
// reference's LstmPredictor (model/predictor/lstm_predictor.py:28-109 -> torchaudio 0.13.1
// `_Predictor`, not vendored; cell restated from the published module):
//     g_t   = g_norm(x2g(x_t) + p2g(h_{t-1}))              LayerNorm over the 4H gates
//     i, f, z, o = chunk(g_t, 4)
//     c_t   = c_norm(sigmoid(f) c_{t-1} + sigmoid(i) tanh(z))    LayerNorm over H (carried on)
//     h_t   = sigmoid(o) tanh(c_t)
// The input projection x2g(x) for all steps is one GEMM outside; the recurrence is latency bound
// (T dependent steps of a (4H x H) matrix-vector product), and utterances are independent, so ONE
// WORKGROUP PER UTTERANCE walks the T steps with h / the gates in LDS and the recurrent matrix
// streamed from L2 (every workgroup reads the same 4H*H floats each step: they stay L2 resident).
// Forward keeps the normalised gates / cells and the two inverse deviations per step; the
// backward walks the steps in reverse, emits d(gates) for all steps (the weight gradients of x2g
// and p2g are then two TN GEMMs over all (t, b) rows) and reduces the LayerNorm parameter
// gradients in registers (one atomic per channel and utterance at the end).
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

constexpr int MAXT = 1024;     // threads per workgroup
constexpr int RPT = 4;         // gate rows per thread: 4H <= RPT * MAXT  (H <= 1024)

struct LstmArgs {
  const float* gx;        // (T, B, 4H)  x2g(x) (+ bias)
  const float* wp;        // forward: p2g.weight TRANSPOSED (H, 4H); backward: p2g.weight (4H, H)
  const float* gg;        // g_norm weight / bias (4H) or NULL (no layer norm)
  const float* gb;
  const float* cg;        // c_norm weight / bias (H) or NULL
  const float* cb;
  const float* h0;        // (B, H) or NULL = zeros
  const float* c0;
  int T, B, H;
  float eps;
  float* hs;              // (T, B, H)
  float* ghat;            // (T, B, 4H) normalised gates (raw gates without layer norm)
  float* chat;            // (T, B, H)  normalised cell (the cell itself without layer norm)
  float* rstd;            // (T, B, 2)  inverse deviations of g_norm / c_norm
  float* hT;              // (B, H) final state
  float* cT;
  // backward
  const float* dhs;       // (T, B, H) gradient w.r.t. hs
  float* dgx;             // (T, B, 4H) gradient w.r.t. the raw gates
  float* dgg;             // accumulated parameter gradients (may be NULL without layer norm)
  float* dgb;
  float* dcg;
  float* dcb;
};

__device__ __forceinline__ float sigm(float x) { return __fdividef(1.f, 1.f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float e = __expf(-2.f * fabsf(x));
  const float t = __fdividef(1.f - e, 1.f + e);
  return x < 0.f ? -t : t;
}

// sums of (a, b) over the first `n_active` threads' values (others pass 0); scratch >= 2 * 16 floats
__device__ __forceinline__ void block_sum2(float& a, float& b, float* scratch) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) {
    scratch[2 * w] = a;
    scratch[2 * w + 1] = b;
  }
  __syncthreads();
  float ra = 0.f, rb = 0.f;
  for (int i = 0; i < nw; ++i) {
    ra += scratch[2 * i];
    rb += scratch[2 * i + 1];
  }
  a = ra;
  b = rb;
}

template <int Q>   // gate rows per thread (ceil(4H / threads))
__global__ __launch_bounds__(MAXT) void lnlstm_fwd_kernel(LstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, G = 4 * a.H, B = a.B, b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  float* hv = smem;               // [H]   h_{t-1}
  float* gate = hv + H;           // [4H]  activated-gate inputs of this step
  float* red = gate + G;          // [32]
  float* part = red + 32;         // [k groups][4H] partial W h products (16-byte aligned)
  const bool ln = a.gg != nullptr;
  float c_prev = 0.f;
  if (tid < H) {
    hv[tid] = a.h0 ? a.h0[(long)b * H + tid] : 0.f;
    c_prev = a.c0 ? a.c0[(long)b * H + tid] : 0.f;
  }
  float wg[Q], wb[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int r = tid + q * nt;
    wg[q] = (ln && r < G) ? a.gg[r] : 1.f;
    wb[q] = (ln && r < G) ? a.gb[r] : 0.f;
  }
  const float cgam = (ln && tid < H) ? a.cg[tid] : 1.f, cbet = (ln && tid < H) ? a.cb[tid] : 0.f;
  __syncthreads();
  for (int t = 0; t < a.T; ++t) {
    const long row = (long)t * B + b;
    float acc[Q];
    // ---- raw gates: gx + Wp h.  The product is latency bound (T dependent steps), so it runs
    // with as many bytes in flight as the CU takes: thread (r4, kg) multiplies the float4 column
    // group r4 of W^T (rows coalesced) with its quarter of h, the k groups meet in LDS.
    {
      const int ng4 = G >> 2, kgs = max(1, nt / ng4), r4 = tid % ng4, kg = tid / ng4;
      if (kg < kgs) {
        const int ks = (H + kgs - 1) / kgs, kb = kg * ks, ke = min(H, kb + ks);
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* wcol = reinterpret_cast<const float4*>(a.wp) + r4;
        for (int k0 = kb; k0 < ke; k0 += 8) {
          float4 w[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const int k = min(k0 + kk, ke - 1);
            w[kk] = wcol[(long)k * ng4];
          }
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            const float hk = k0 + kk < ke ? hv[k0 + kk] : 0.f;
            s4.x = fmaf(w[kk].x, hk, s4.x); s4.y = fmaf(w[kk].y, hk, s4.y);
            s4.z = fmaf(w[kk].z, hk, s4.z); s4.w = fmaf(w[kk].w, hk, s4.w);
          }
        }
        reinterpret_cast<float4*>(part + (long)kg * G)[r4] = s4;
      }
      __syncthreads();
      float accs[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int r = tid + q * nt;
        float v = 0.f;
        if (r < G) {
          v = a.gx[row * G + r];
          for (int g = 0; g < kgs; ++g) v += part[(long)g * G + r];
        }
        accs[q] = v;
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[q] = accs[q];
    }
    // ---- g_norm
    float rs_g = 1.f;
    if (ln) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (tid + q * nt < G) s1 += acc[q];
      float dummy = 0.f;
      block_sum2(s1, dummy, red);
      const float mean = s1 / (float)G;
#pragma unroll
      for (int q = 0; q < Q; ++q)
        if (tid + q * nt < G) {
          acc[q] -= mean;
          s2 += acc[q] * acc[q];
        }
      dummy = 0.f;
      block_sum2(s2, dummy, red);
      rs_g = rsqrtf(s2 / (float)G + a.eps);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int r = tid + q * nt;
      if (r < G) {
        const float xh = ln ? acc[q] * rs_g : acc[q];
        a.ghat[row * G + r] = xh;
        gate[r] = wg[q] * xh + wb[q];
      }
    }
    __syncthreads();
    // ---- cell: thread j < H
    float cpre = 0.f, og = 0.f;
    if (tid < H) {
      const float ig = sigm(gate[tid]), fg = sigm(gate[H + tid]), zg = tanh_f(gate[2 * H + tid]);
      og = sigm(gate[3 * H + tid]);
      cpre = fg * c_prev + ig * zg;
    }
    float rs_c = 1.f, ch = cpre;
    if (ln) {
      float s1 = tid < H ? cpre : 0.f, dummy = 0.f;
      block_sum2(s1, dummy, red);
      const float mean = s1 / (float)H;
      ch = cpre - mean;
      float s2 = tid < H ? ch * ch : 0.f;
      dummy = 0.f;
      block_sum2(s2, dummy, red);
      rs_c = rsqrtf(s2 / (float)H + a.eps);
      ch *= rs_c;
    } else {
      __syncthreads();              // every thread is past its reads of hv / gate
    }
    if (tid < H) {
      const float cn = cgam * ch + cbet;
      const float h = og * tanh_f(cn);
      a.chat[row * H + tid] = ch;
      a.hs[row * H + tid] = h;
      hv[tid] = h;
      c_prev = cn;
      if (t == a.T - 1) {
        a.hT[(long)b * H + tid] = h;
        a.cT[(long)b * H + tid] = cn;
      }
    }
    if (tid == 0) {
      a.rstd[2 * row] = rs_g;
      a.rstd[2 * row + 1] = rs_c;
    }
    __syncthreads();
  }
}

template <int Q>
__global__ __launch_bounds__(MAXT) void lnlstm_bwd_kernel(LstmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = a.H, G = 4 * a.H, B = a.B, b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  float* dhr = smem;              // [H]    recurrent gradient w.r.t. h_{t-1}
  float* da = dhr + H;            // [4H]   gradient w.r.t. the (affine) gate inputs, then dg_raw
  float* red = da + G;            // [32]
  float* part = red + 32;         // [nparts][H] partial W^T dg products (16-byte aligned)
  const int nparts = max(1, nt / (H >> 2));
  const bool ln = a.gg != nullptr;
  if (tid < H) dhr[tid] = 0.f;
  float wg[Q], wb[Q], agg[Q], agb[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int r = tid + q * nt;
    wg[q] = (ln && r < G) ? a.gg[r] : 1.f;
    wb[q] = (ln && r < G) ? a.gb[r] : 0.f;
    agg[q] = agb[q] = 0.f;
  }
  const float cgam = (ln && tid < H) ? a.cg[tid] : 1.f, cbet = (ln && tid < H) ? a.cb[tid] : 0.f;
  // hidden unit j also needs the affine parameters of ITS four gate rows j, H+j, 2H+j, 3H+j
  float ug[4] = {1.f, 1.f, 1.f, 1.f}, ub[4] = {0.f, 0.f, 0.f, 0.f};
  if (ln && tid < H) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ug[q] = a.gg[q * H + tid];
      ub[q] = a.gb[q * H + tid];
    }
  }
  float acg = 0.f, acb = 0.f, dc_carry = 0.f;
  __syncthreads();
  for (int t = a.T - 1; t >= 0; --t) {
    const long row = (long)t * B + b;
    const float rs_g = a.rstd[2 * row], rs_c = a.rstd[2 * row + 1];
    // ---- through h = o tanh(c_n), c_norm, and the cell update: thread j < H
    float dchat = 0.f, ch = 0.f, ig = 0.f, fg = 0.f, zg = 0.f, og = 0.f, dcn = 0.f, dog = 0.f;
    if (tid < H) {
      const float* gh = a.ghat + row * G;
      ig = sigm(ug[0] * gh[tid] + ub[0]);
      fg = sigm(ug[1] * gh[H + tid] + ub[1]);
      zg = tanh_f(ug[2] * gh[2 * H + tid] + ub[2]);
      og = sigm(ug[3] * gh[3 * H + tid] + ub[3]);
      ch = a.chat[row * H + tid];
      const float cn = cgam * ch + cbet;
      const float tc = tanh_f(cn);
      const float dh = a.dhs[row * H + tid] + dhr[tid];
      dog = dh * tc * og * (1.f - og);
      dcn = dh * og * (1.f - tc * tc) + dc_carry;
      acg += dcn * ch;
      acb += dcn;
      dchat = dcn * cgam;
    }
    float dcpre = dcn;
    if (ln) {
      float s1 = tid < H ? dchat : 0.f, s2 = tid < H ? dchat * ch : 0.f;
      block_sum2(s1, s2, red);
      dcpre = rs_c * (dchat - s1 / (float)H - ch * (s2 / (float)H));
    }
    if (tid < H) {
      float cprev;
      if (t > 0) {
        const float chp = a.chat[(row - B) * H + tid];
        cprev = cgam * chp + cbet;
      } else {
        cprev = a.c0 ? a.c0[(long)b * H + tid] : 0.f;
      }
      da[tid] = dcpre * zg * ig * (1.f - ig);
      da[H + tid] = dcpre * cprev * fg * (1.f - fg);
      da[2 * H + tid] = dcpre * ig * (1.f - zg * zg);
      da[3 * H + tid] = dog;
      dc_carry = dcpre * fg;
    }
    __syncthreads();
    // ---- through g_norm: thread owns rows tid + q nt
    float dx[Q], xh[Q];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int r = tid + q * nt;
      dx[q] = xh[q] = 0.f;
      if (r < G) {
        const float d = da[r];
        xh[q] = a.ghat[row * G + r];
        agg[q] += d * xh[q];
        agb[q] += d;
        dx[q] = d * wg[q];
        s1 += dx[q];
        s2 += dx[q] * xh[q];
      }
    }
    if (ln) {
      block_sum2(s1, s2, red);                // (its barriers also order the da reads above)
      s1 /= (float)G;
      s2 /= (float)G;
    } else {
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int r = tid + q * nt;
      if (r < G) {
        const float d = ln ? rs_g * (dx[q] - s1 - xh[q] * s2) : dx[q];
        a.dgx[row * G + r] = d;
        da[r] = d;
      }
    }
    __syncthreads();
    // ---- recurrent gradient: dh_{t-1}[k] = sum_r Wp[r][k] dg[r]: thread (k4, row group) walks
    // every nparts-th row of Wp with float4 loads (rows are k-contiguous), the groups meet in LDS
    {
      const int nh4 = H >> 2, k4 = tid % nh4, p = tid / nh4;
      if (p < nparts) {
        float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* wk = reinterpret_cast<const float4*>(a.wp) + k4;
        for (int r0 = p; r0 < G; r0 += 8 * nparts) {
          float4 w[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = min(r0 + u * nparts, G - 1);
            w[u] = wk[(long)r * nh4];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int r = r0 + u * nparts;
            const float d = r < G ? da[r] : 0.f;
            s4.x = fmaf(w[u].x, d, s4.x); s4.y = fmaf(w[u].y, d, s4.y);
            s4.z = fmaf(w[u].z, d, s4.z); s4.w = fmaf(w[u].w, d, s4.w);
          }
        }
        reinterpret_cast<float4*>(part + (long)p * H)[k4] = s4;
      }
    }
    __syncthreads();
    if (tid < H) {
      float s = 0.f;
      for (int p = 0; p < nparts; ++p) s += part[p * H + tid];
      dhr[tid] = s;
    }
    __syncthreads();
  }
  if (ln) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int r = tid + q * nt;
      if (r < G) {
        atomicAdd(a.dgg + r, agg[q]);
        atomicAdd(a.dgb + r, agb[q]);
      }
    }
    if (tid < H) {
      atomicAdd(a.dcg + tid, acg);
      atomicAdd(a.dcb + tid, acb);
    }
  }
}

int lstm_threads(int H) {
  const int G = 4 * H;
  int nt = ((G + 63) / 64) * 64;
  return nt > MAXT ? MAXT : nt;
}

}  // namespace

extern "C" {

int s2t_lnlstm_fwd(const float* gx, const float* wp_t, const float* g_gamma, const float* g_beta,
                   const float* c_gamma, const float* c_beta, const float* h0, const float* c0,
                   int T, int B, int H, float eps, float* hs, float* ghat, float* chat,
                   float* rstd, float* hT, float* cT, void* stream) {
  if (T <= 0 || B <= 0) return 0;
  if (H <= 0 || (H & 3) || 4 * H > RPT * MAXT) return -2;
  if ((g_gamma == nullptr) != (c_gamma == nullptr)) return -1;
  LstmArgs a{gx, wp_t, g_gamma, g_beta, c_gamma, c_beta, h0, c0, T, B, H, eps, hs, ghat, chat,
             rstd, hT, cT, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const int nt = lstm_threads(H);
  const int kgs = nt / H > 0 ? nt / H : 1;               // k groups of the forward product
  const size_t smem = sizeof(float) * (5 * (size_t)H + 32 + (size_t)kgs * 4 * H);
  switch ((4 * H + nt - 1) / nt) {
    case 1: hipLaunchKernelGGL(lnlstm_fwd_kernel<1>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(lnlstm_fwd_kernel<2>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL(lnlstm_fwd_kernel<3>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(lnlstm_fwd_kernel<4>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_lnlstm_bwd(const float* wp, const float* g_gamma, const float* g_beta,
                   const float* c_gamma, const float* c_beta, const float* c0, int T, int B, int H,
                   const float* ghat, const float* chat, const float* rstd, const float* dhs,
                   float* dgx, float* d_g_gamma, float* d_g_beta, float* d_c_gamma,
                   float* d_c_beta, void* stream) {
  if (T <= 0 || B <= 0) return 0;
  if (H <= 0 || (H & 3) || 4 * H > RPT * MAXT) return -2;
  if ((g_gamma == nullptr) != (c_gamma == nullptr)) return -1;
  LstmArgs a{nullptr, wp, g_gamma, g_beta, c_gamma, c_beta, nullptr, c0, T, B, H, 0.f, nullptr,
             const_cast<float*>(ghat), const_cast<float*>(chat), const_cast<float*>(rstd), nullptr,
             nullptr, dhs, dgx, d_g_gamma, d_g_beta, d_c_gamma, d_c_beta};
  const int nt = lstm_threads(H);
  const int nparts = nt / (H / 4) > 0 ? nt / (H / 4) : 1;
  const size_t smem = sizeof(float) * ((5 + (size_t)nparts) * H + 32);
  switch ((4 * H + nt - 1) / nt) {
    case 1: hipLaunchKernelGGL(lnlstm_bwd_kernel<1>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
    case 2: hipLaunchKernelGGL(lnlstm_bwd_kernel<2>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
    case 3: hipLaunchKernelGGL(lnlstm_bwd_kernel<3>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(lnlstm_bwd_kernel<4>, dim3(B), dim3(nt), smem, (hipStream_t)stream, a); break;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
