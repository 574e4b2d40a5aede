// Multi-head self-attention core of the conformer block (nn.MultiheadAttention inside
// torchaudio.models.Conformer; call site model/encoder/conformer.py:170-178,193) for gfx950:
//     O = softmax(Q K^T / sqrt(dh) + key_padding_mask) V         per (utterance, head)
// forward and backward, fp32 on v_mfma_f32_32x32x2_f32, flash-style (the T x T weights are never
// written to HBM; the backward recomputes them from the saved row log-sum-exp).
//
// Layout: q, k, v are column blocks of the in-projection's output (T, B, 3D) time-major -- no
// (B, H, T, dh) permute copies; O / dO are (T, B, D).  A wave owns 32 query rows (forward, dQ)
// or 32 keys (dK / dV); the other operand streams through LDS in tiles of 64 rows, staged
// global -> registers -> LDS with the next tile's loads in flight under the MFMAs.
//
// Orientation (guide: "an accumulator tile as the next MFMA's operand"): every product is
// arranged so that the wave's own rows sit on the LANES of the accumulator and the streamed
// rows in its registers.  S^T = K Q^T has lane = query, regs = keys, so the softmax statistics
// are per lane (no cross-lane row reductions beyond one half-wave exchange), and P feeds the next
// MFMA (O^T = V^T P^T) straight from the accumulator registers: no LDS round trip for P.
#include "common.h"
#include "../../include/s2t_mi355.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MhsaArgs {
  const float* qkv;     // (T*B, ld): row t*B + b
  long ld;
  int qoff, koff, voff;
  const long* lens;     // [B] valid keys (frames), or NULL = all T
  int T, B, H;
  float scale;
  float* o;             // forward out (T*B, ldo)
  const float* o_in;    // backward: saved forward output
  const float* d_o;     // backward: gradient w.r.t. o
  long ldo;
  float* lse;           // [B][H][T]
  float* delta;         // [B][H][T]   rowsum(dO * O)
  float* dqkv;          // (T*B, ld) same column layout as qkv
  unsigned drop_thr;    // attention-probability dropout: drop where hash < thr (= p * 2^32)
  float inv_keep;       // 1 / (1 - p)
  unsigned long long seed;
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Dropout on the attention probabilities (nn.MultiheadAttention(dropout=p) in training): the keep
// decision of element (b, h, q, k) is a stateless hash of (seed, index), so the backward kernels
// regenerate exactly the forward's mask instead of storing a T x T tensor (splitmix64 finaliser).
__device__ __forceinline__ bool keep_elem(unsigned long long seed, long idx, unsigned thr) {
  unsigned long long z = seed + (unsigned long long)idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32) >= thr;
}

// row index inside a 32-row MFMA tile held in accumulator register r of a lane in half `hi`
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---- a 64-row x DH tile: global -> registers (clamped rows) -> LDS [64][DH + 4] (zero beyond T)
template <int DH>
struct RowTile {
  static constexpr int LD = DH + 4;
  static constexpr int NV = DH / 16;          // float4 per thread (256 threads)
  static constexpr int VPR = DH / 4;          // float4 per row
  __device__ static __forceinline__ void load(float4 (&v)[NV], const float* __restrict__ base,
                                              long ld, int B, int b, int t0, int T) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int r = idx / VPR, c = idx % VPR;
      const int t = min(t0 + r, T - 1);
      v[i] = *reinterpret_cast<const float4*>(base + ((long)t * B + b) * ld + 4 * c);
    }
  }
  __device__ static __forceinline__ void store(float* __restrict__ s, const float4 (&v)[NV],
                                               int t0, int T, float scale) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int r = idx / VPR, c = idx % VPR;
      float4 x = v[i];
      if (t0 + r >= T) x = make_float4(0.f, 0.f, 0.f, 0.f);
      else { x.x *= scale; x.y *= scale; x.z *= scale; x.w *= scale; }
      *reinterpret_cast<float4*>(s + r * LD + 4 * c) = x;
    }
  }
};

// row fragments of a wave's own 32 rows (clamped to T-1): element j of group g = col 8g + 4hi + j
template <int DH>
__device__ __forceinline__ void load_frags(float4 (&f)[DH / 8], const float* __restrict__ base,
                                           long ld, int B, int b, int row, int T, int hi,
                                           float scale) {
  const int t = min(row, T - 1);
  const float* p = base + ((long)t * B + b) * ld + 4 * hi;
#pragma unroll
  for (int g = 0; g < DH / 8; ++g) {
    float4 x = *reinterpret_cast<const float4*>(p + 8 * g);
    f[g] = make_float4(x.x * scale, x.y * scale, x.z * scale, x.w * scale);
  }
}

// acc (lane = own row, regs = streamed rows of sub-tile `sub`) = tile_rows . own_rows^T
template <int DH>
__device__ __forceinline__ f32x16 dot_tile(const float* __restrict__ s, int sub, int lo, int hi,
                                           const float4 (&own)[DH / 8]) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* kr = s + (32 * sub + lo) * (DH + 4) + 4 * hi;
#pragma unroll
  for (int g = 0; g < DH / 8; ++g) {
    const float4 kf = *reinterpret_cast<const float4*>(kr + 8 * g);
    acc = MFMA(kf.x, own[g].x, acc);
    acc = MFMA(kf.y, own[g].y, acc);
    acc = MFMA(kf.z, own[g].z, acc);
    acc = MFMA(kf.w, own[g].w, acc);
  }
  return acc;
}

// out^T (DH x own rows) += tile^T (DH x streamed rows) . w (streamed rows x own rows), w held in
// accumulator layout (lane = own row, regs = streamed rows)
template <int DH>
__device__ __forceinline__ void accum_tile(f32x16 (&out)[(DH + 31) / 32],
                                           const float* __restrict__ s, int sub, int lo, int hi,
                                           const f32x16& w) {
  constexpr int NDT = (DH + 31) / 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* row = s + (32 * sub + acc_row(r, hi)) * (DH + 4);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = 32 * dt + lo;
      const float a = (DH % 32 == 0 || d < DH) ? row[d] : 0.f;
      out[dt] = MFMA(a, w[r], out[dt]);
    }
  }
}

// write out^T (lane = own row `t`, regs = d) to dst[(t*B + b)*ld + d], d = 32 dt + 8 c + 4 hi + j
template <int DH>
__device__ __forceinline__ void write_rows(const f32x16 (&acc)[(DH + 31) / 32],
                                           float* __restrict__ dst, long ld, int B, int b, int t,
                                           int T, int hi, float mul) {
  if (t >= T) return;
  float* p = dst + ((long)t * B + b) * ld;
#pragma unroll
  for (int dt = 0; dt < (DH + 31) / 32; ++dt)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = 32 * dt + 8 * c + 4 * hi;
      if (DH % 32 == 0 || d < DH)
        *reinterpret_cast<float4*>(p + d) =
            make_float4(acc[dt][4 * c] * mul, acc[dt][4 * c + 1] * mul, acc[dt][4 * c + 2] * mul,
                        acc[dt][4 * c + 3] * mul);
    }
}

// ------------------------------------------------------------------ forward
template <int DH, bool DROP>
__global__ __launch_bounds__(256) void mhsa_fwd_kernel(MhsaArgs a) {
  using TL = RowTile<DH>;
  constexpr int NDT = (DH + 31) / 32;
  __shared__ __attribute__((aligned(16))) float Ks[64 * TL::LD];
  __shared__ __attribute__((aligned(16))) float Vs[64 * TL::LD];
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 31, hi = lane >> 5;
  const int q = blockIdx.x * 128 + wave * 32 + lo;
  const int T = a.T, B = a.B;
  const int len = a.lens ? (int)min((long)T, max(0L, a.lens[b])) : T;
  const float* kbase = a.qkv + a.koff + h * DH;
  const float* vbase = a.qkv + a.voff + h * DH;
  float4 qf[DH / 8];
  load_frags<DH>(qf, a.qkv + a.qoff + h * DH, a.ld, B, b, q, T, hi, a.scale);
  f32x16 oacc[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
  float m = S2T_NEG_INF, l = 0.f;
  float4 pk[TL::NV], pv[TL::NV];
  if (len > 0) {
    TL::load(pk, kbase, a.ld, B, b, 0, T);
    TL::load(pv, vbase, a.ld, B, b, 0, T);
  }
  for (int k0 = 0; k0 < len; k0 += 64) {
    __syncthreads();
    TL::store(Ks, pk, k0, T, 1.f);
    TL::store(Vs, pv, k0, T, 1.f);
    __syncthreads();
    if (k0 + 64 < len) {
      TL::load(pk, kbase, a.ld, B, b, k0 + 64, T);
      TL::load(pv, vbase, a.ld, B, b, k0 + 64, T);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kb = k0 + 32 * sub;
      if (kb >= len) break;
      f32x16 s = dot_tile<DH>(Ks, sub, lo, hi, qf);
      float tmax = S2T_NEG_INF;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kb + acc_row(r, hi) >= len) s[r] = S2T_NEG_INF;
        tmax = fmaxf(tmax, s[r]);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float mn = fmaxf(m, tmax);
      const float ms = mn == S2T_NEG_INF ? 0.f : mn;
      const float alpha = __expf(m - ms);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = __expf(s[r] - ms);
        psum += s[r];
      }
      l = l * alpha + psum;
      m = mn;
      if (DROP) {
        const long rowbase = (((long)b * a.H + h) * T + min(q, T - 1)) * T + kb;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          s[r] = keep_elem(a.seed, rowbase + acc_row(r, hi), a.drop_thr) ? s[r] * a.inv_keep : 0.f;
      }
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
      accum_tile<DH>(oacc, Vs, sub, lo, hi, s);
    }
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  const float inv = lt > 0.f ? __fdividef(1.f, lt) : 0.f;
  write_rows<DH>(oacc, a.o + h * DH, a.ldo, B, b, q, T, hi, inv);
  if (hi == 0 && q < T) a.lse[((long)b * a.H + h) * T + q] = lt > 0.f ? m + __logf(lt) : 0.f;
}

// ------------------------------------------------------------------ backward: dQ (+ delta)
template <int DH, bool DROP>
__global__ __launch_bounds__(256) void mhsa_bwd_q_kernel(MhsaArgs a) {
  using TL = RowTile<DH>;
  constexpr int NDT = (DH + 31) / 32;
  __shared__ __attribute__((aligned(16))) float Ks[64 * TL::LD];
  __shared__ __attribute__((aligned(16))) float Vs[64 * TL::LD];
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 31, hi = lane >> 5;
  const int q = blockIdx.x * 128 + wave * 32 + lo;
  const int T = a.T, B = a.B;
  const int len = a.lens ? (int)min((long)T, max(0L, a.lens[b])) : T;
  const float* kbase = a.qkv + a.koff + h * DH;
  const float* vbase = a.qkv + a.voff + h * DH;
  float4 qf[DH / 8], dof[DH / 8];
  load_frags<DH>(qf, a.qkv + a.qoff + h * DH, a.ld, B, b, q, T, hi, a.scale);
  load_frags<DH>(dof, a.d_o + h * DH, a.ldo, B, b, q, T, hi, 1.f);
  float delta = 0.f;
  {
    float4 of[DH / 8];
    load_frags<DH>(of, a.o_in + h * DH, a.ldo, B, b, q, T, hi, 1.f);
#pragma unroll
    for (int g = 0; g < DH / 8; ++g)
      delta += (of[g].x * dof[g].x + of[g].y * dof[g].y) + (of[g].z * dof[g].z + of[g].w * dof[g].w);
    delta += __shfl_xor(delta, 32, 64);
  }
  const long sidx = ((long)b * a.H + h) * T + min(q, T - 1);
  if (hi == 0 && q < T) a.delta[sidx] = delta;
  const float lse = a.lse[sidx];
  f32x16 dq[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
  float4 pk[TL::NV], pv[TL::NV];
  if (len > 0) {
    TL::load(pk, kbase, a.ld, B, b, 0, T);
    TL::load(pv, vbase, a.ld, B, b, 0, T);
  }
  for (int k0 = 0; k0 < len; k0 += 64) {
    __syncthreads();
    TL::store(Ks, pk, k0, T, 1.f);
    TL::store(Vs, pv, k0, T, 1.f);
    __syncthreads();
    if (k0 + 64 < len) {
      TL::load(pk, kbase, a.ld, B, b, k0 + 64, T);
      TL::load(pv, vbase, a.ld, B, b, k0 + 64, T);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int kb = k0 + 32 * sub;
      if (kb >= len) break;
      f32x16 s = dot_tile<DH>(Ks, sub, lo, hi, qf);
      const f32x16 dp = dot_tile<DH>(Vs, sub, lo, hi, dof);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = (kb + acc_row(r, hi) < len) ? __expf(s[r] - lse) : 0.f;
        float g = dp[r];
        if (DROP)
          g = keep_elem(a.seed, (((long)b * a.H + h) * T + min(q, T - 1)) * T + kb + acc_row(r, hi),
                        a.drop_thr) ? g * a.inv_keep : 0.f;
        s[r] = p * (g - delta);
      }
      accum_tile<DH>(dq, Ks, sub, lo, hi, s);
    }
  }
  write_rows<DH>(dq, a.dqkv + a.qoff + h * DH, a.ld, B, b, q, T, hi, a.scale);
}

// ------------------------------------------------------------------ backward: dK, dV
template <int DH, bool DROP>
__global__ __launch_bounds__(256) void mhsa_bwd_kv_kernel(MhsaArgs a) {
  using TL = RowTile<DH>;
  constexpr int NDT = (DH + 31) / 32;
  __shared__ __attribute__((aligned(16))) float Qs[64 * TL::LD];
  __shared__ __attribute__((aligned(16))) float Ds[64 * TL::LD];
  __shared__ __attribute__((aligned(16))) float lse_s[64];
  __shared__ __attribute__((aligned(16))) float del_s[64];
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lo = lane & 31, hi = lane >> 5;
  const int key = blockIdx.x * 128 + wave * 32 + lo;
  const int T = a.T, B = a.B;
  const int len = a.lens ? (int)min((long)T, max(0L, a.lens[b])) : T;
  f32x16 dk[NDT], dv[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;
  if ((int)blockIdx.x * 128 < len) {             // (workgroup-uniform) some key of this block is valid
    const float* qbase = a.qkv + a.qoff + h * DH;
    const float* dobase = a.d_o + h * DH;
    float4 kf[DH / 8], vf[DH / 8];
    load_frags<DH>(kf, a.qkv + a.koff + h * DH, a.ld, B, b, key, T, hi, 1.f);
    load_frags<DH>(vf, a.qkv + a.voff + h * DH, a.ld, B, b, key, T, hi, 1.f);
    const bool key_ok = key < len;
    const long sbase = ((long)b * a.H + h) * T;
    float4 pq[TL::NV], pd[TL::NV];
    TL::load(pq, qbase, a.ld, B, b, 0, T);
    TL::load(pd, dobase, a.ldo, B, b, 0, T);
    float pl = 0.f, pdl = 0.f;
    if (threadIdx.x < 64) {
      const int t = min((int)threadIdx.x, T - 1);
      pl = a.lse[sbase + t];
      pdl = a.delta[sbase + t];
    }
    for (int q0 = 0; q0 < T; q0 += 64) {
      __syncthreads();
      TL::store(Qs, pq, q0, T, a.scale);
      TL::store(Ds, pd, q0, T, 1.f);
      if (threadIdx.x < 64) {
        const bool ok = q0 + (int)threadIdx.x < T;
        lse_s[threadIdx.x] = ok ? pl : __builtin_huge_valf();     // exp(s - inf) = 0
        del_s[threadIdx.x] = ok ? pdl : 0.f;
      }
      __syncthreads();
      if (q0 + 64 < T) {
        TL::load(pq, qbase, a.ld, B, b, q0 + 64, T);
        TL::load(pd, dobase, a.ldo, B, b, q0 + 64, T);
        if (threadIdx.x < 64) {
          const int t = min(q0 + 64 + (int)threadIdx.x, T - 1);
          pl = a.lse[sbase + t];
          pdl = a.delta[sbase + t];
        }
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        if (q0 + 32 * sub >= T) break;
        f32x16 s = dot_tile<DH>(Qs, sub, lo, hi, kf);        // lane = key, regs = queries
        f32x16 dp = dot_tile<DH>(Ds, sub, lo, hi, vf);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 ls = *reinterpret_cast<const float4*>(lse_s + 32 * sub + 8 * c + 4 * hi);
          const float4 dl = *reinterpret_cast<const float4*>(del_s + 32 * sub + 8 * c + 4 * hi);
          const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * c + j;
            const float p = key_ok ? __expf(s[r] - lsv[j]) : 0.f;
            float pd = p, g = dp[r];
            if (DROP) {
              const int qq = min(q0 + 32 * sub + acc_row(r, hi), T - 1);
              const bool keep = keep_elem(a.seed, (((long)b * a.H + h) * T + qq) * T + min(key, T - 1),
                                          a.drop_thr);
              pd = keep ? p * a.inv_keep : 0.f;
              g = keep ? g * a.inv_keep : 0.f;
            }
            s[r] = pd;
            dp[r] = p * (g - dlv[j]);
          }
        }
        accum_tile<DH>(dv, Ds, sub, lo, hi, s);
        accum_tile<DH>(dk, Qs, sub, lo, hi, dp);
      }
    }
  }
  write_rows<DH>(dk, a.dqkv + a.koff + h * DH, a.ld, B, b, key, T, hi, 1.f);
  write_rows<DH>(dv, a.dqkv + a.voff + h * DH, a.ld, B, b, key, T, hi, 1.f);
}

bool mhsa_ok(const MhsaArgs& a, int dh) {
  if (a.T <= 0 || a.B <= 0 || a.H <= 0) return false;
  if (dh != 16 && dh != 32 && dh != 64) return false;
  if ((a.ld & 3) || (a.ldo & 3) || ((a.qoff | a.koff | a.voff) & 3)) return false;
  if (reinterpret_cast<uintptr_t>(a.qkv) & 15) return false;
  if (a.B > 65535 || a.H > 65535) return false;
  return true;
}

#define MHSA_DISPATCH2(KERNEL, DROP, dh, grid, st, a)                                       \
  switch (dh) {                                                                             \
    case 16: hipLaunchKernelGGL((KERNEL<16, DROP>), grid, dim3(256), 0, st, a); break;      \
    case 32: hipLaunchKernelGGL((KERNEL<32, DROP>), grid, dim3(256), 0, st, a); break;      \
    default: hipLaunchKernelGGL((KERNEL<64, DROP>), grid, dim3(256), 0, st, a); break;      \
  }
#define MHSA_DISPATCH(KERNEL, dh, grid, st, a)                                              \
  if (a.drop_thr) { MHSA_DISPATCH2(KERNEL, true, dh, grid, st, a) }                         \
  else { MHSA_DISPATCH2(KERNEL, false, dh, grid, st, a) }

bool set_dropout(MhsaArgs& a, float p, unsigned long long seed) {
  if (!(p >= 0.f && p < 1.f)) return false;
  a.drop_thr = p > 0.f ? (unsigned)((double)p * 4294967296.0) : 0u;
  if (p > 0.f && a.drop_thr == 0u) a.drop_thr = 1u;
  a.inv_keep = 1.f / (1.f - p);
  a.seed = seed;
  return true;
}

}  // namespace

extern "C" {

int s2t_mhsa_fwd(const float* qkv, long ld, int qoff, int koff, int voff, const long* lens, int T,
                 int B, int H, int dh, float scale, float dropout_p, unsigned long seed, float* o,
                 long ldo, float* lse, void* stream) {
  MhsaArgs a{qkv, ld, qoff, koff, voff, lens, T, B, H, scale, o, nullptr, nullptr, ldo, lse,
             nullptr, nullptr, 0u, 1.f, 0ull};
  if (!mhsa_ok(a, dh) || (reinterpret_cast<uintptr_t>(o) & 15)) return -2;
  if (!set_dropout(a, dropout_p, seed)) return -1;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((T + 127) / 128, H, B);
  MHSA_DISPATCH(mhsa_fwd_kernel, dh, grid, st, a);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_mhsa_bwd(const float* qkv, long ld, int qoff, int koff, int voff, const long* lens, int T,
                 int B, int H, int dh, float scale, float dropout_p, unsigned long seed,
                 const float* o, const float* d_o, long ldo, const float* lse, float* delta,
                 float* dqkv, void* stream) {
  MhsaArgs a{qkv, ld, qoff, koff, voff, lens, T, B, H, scale, nullptr, o, d_o, ldo,
             const_cast<float*>(lse), delta, dqkv, 0u, 1.f, 0ull};
  if (!set_dropout(a, dropout_p, seed)) return -1;
  if (!mhsa_ok(a, dh) || (reinterpret_cast<uintptr_t>(o) & 15) ||
      (reinterpret_cast<uintptr_t>(d_o) & 15) || (reinterpret_cast<uintptr_t>(dqkv) & 15))
    return -2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((T + 127) / 128, H, B);
  MHSA_DISPATCH(mhsa_bwd_q_kernel, dh, grid, st, a);
  S2T_CHECK_LAUNCH();
  MHSA_DISPATCH(mhsa_bwd_kv_kernel, dh, grid, st, a);
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
