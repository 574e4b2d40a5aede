// Relative-position multi-head attention weights for gfx950 (zipformer), fused:
//   s[h,b,i,j] = q[i].k[j] + p[i].pos[(T-1) - i + j]      (rel->abs shift by index arithmetic)
//   s = -1000 where attn_mask[i][j] or key j is padding ;  W = softmax_j(s)
// Reference: model/encoder/zipformer.py:1966-2066 (q/k/p split of in_proj, matmul, pos matmul,
// as_strided rel->abs, two masked_fill(-1000), softmax).  The reference materialises the
// (H,B,T,T) scores, the (H,B,T,2T-1) position scores and several masked copies; here the only
// (H,B,T,T) HBM traffic is ONE write of W in forward and ONE read of W and of dW per backward
// kernel.
//
// Layout: qkp (T,B,Dp) = in_proj output, Dp = H*(2*qd+pd): [q: H*qd | k: H*qd | p: H*pd];
// pos (2T-1, H*pd) = linear_pos(pos_emb); W (H,B,T,T).
// Forward: workgroup = (64 query rows, b, h); the key tile (<=512 keys) is staged TRANSPOSED
// in LDS (Kt[d][j], conflict-free across lanes), lanes own keys j = lane + 64 m, each wave
// register-blocks 4 query rows x 8 keys so every LDS operand feeds 4-8 FMAs; softmax statistics
// are wave reductions.  One key tile (T <= 512, the 10 s utterances of the benchmark) is a
// single pass with the scores held in registers; longer sequences use two passes
// (statistics, then recompute + write) so LDS use is independent of T.
#include "common.h"
#include <cstdint>
#include <cstdlib>

namespace {

// keys per tile = 64 * NM (NM = keys per lane, chosen from T); LDS rows are padded by one
// float so that the transposed staging writes are bank-conflict free
#define JT (64 * NM)
#define JTP (JT + 1)
#define PWP (JT + 64 + 1)
constexpr int ROWS = 64;      // query rows per workgroup
constexpr int MAXQD = 32;
constexpr int MAXPD = 8;

struct AttnArgs {
  const float* qkp;   // (T,B,Dp)
  const float* pos;   // (2T-1, H*pd) or null
  const unsigned char* kpm;    // (B,T) 1 = padded key, or null
  const unsigned char* amask;  // (T,T) 1 = masked, or null
  int T, B, H, qd, pd;
  // gradient of W, supplied either materialised (dW) and/or as factors that are contracted
  // on the fly: dW[h,b,i,j] = dW[...] + (h==0) dW0[b,i,j] + sum_c sum_d dO_c[i,b,h,d] V_c[j,b,h,d]
  const float* dW;     // (H,B,T,T) or null
  const float* dW0;    // (B,T,T) head-0 extra or null
  const float* pdO[2]; // (T,B,H*dv_c) each, or null
  const float* pV[2];
  int pdv[2];
  // forward only: *pen_flag = 1 when any raw score (before masking) exceeds pen_limit in absolute
  // value -- tells the caller whether penalize_abs_values_gt has a non-zero gradient this call
  float pen_limit;
  float* pen_flag;
};

__device__ __forceinline__ const float* q_row(const AttnArgs& a, int t, int b, int h) {
  return a.qkp + ((long)t * a.B + b) * (a.H * (2 * a.qd + a.pd)) + h * a.qd;
}
__device__ __forceinline__ const float* k_row(const AttnArgs& a, int t, int b, int h) {
  return a.qkp + ((long)t * a.B + b) * (a.H * (2 * a.qd + a.pd)) + a.H * a.qd + h * a.qd;
}
__device__ __forceinline__ const float* p_row(const AttnArgs& a, int t, int b, int h) {
  return a.qkp + ((long)t * a.B + b) * (a.H * (2 * a.qd + a.pd)) + 2 * a.H * a.qd + h * a.pd;
}

// LDS carve-up shared by the kernels
struct Smem {
  float* Kt;    // [qd][JT]
  float* Pt;    // [pd][JT + ROWS]   pos window, transposed
  float* Q;     // [ROWS][MAXQD]
  float* P;     // [ROWS][MAXPD]
};
template <int NM>
__device__ __forceinline__ Smem carve(unsigned char* raw, int qd, int pd) {
  Smem s;
  s.Kt = reinterpret_cast<float*>(raw);
  s.Pt = s.Kt + qd * JTP;
  s.Q = s.Pt + pd * PWP;
  s.P = s.Q + ROWS * MAXQD;
  return s;
}
template <int NM>
inline size_t attn_smem(int qd, int pd) {
  return sizeof(float) * ((size_t)qd * JTP + (size_t)pd * PWP + ROWS * MAXQD + ROWS * MAXPD);
}

// stage K^T tile (keys j0..j0+JT) and the pos window for (i0, j0).  (j,d) are advanced
// incrementally: no per-element integer division.
template <int NM>
__device__ __forceinline__ void stage_tile(const AttnArgs& a, const Smem& s, int b, int h, int i0,
                                           int j0) {
  const int qd = a.qd, pd = a.pd;
  {
    int j = threadIdx.x / qd, d = threadIdx.x % qd;
    const int dj = 256 / qd, dd = 256 % qd;
    const long rs = (long)a.B * (a.H * (2 * qd + pd));
    const float* kb = k_row(a, 0, b, h);
    while (j < JT) {
      const int t = j0 + j;
      s.Kt[d * JTP + j] = t < a.T ? kb[(long)t * rs + d] : 0.f;
      j += dj;
      d += dd;
      if (d >= qd) {
        d -= qd;
        ++j;
      }
    }
  }
  if (a.pos) {
    // window entry w <-> rel index (T-1) - (i0 + ROWS-1) + j0 + w
    const int base = (a.T - 1) - (i0 + ROWS - 1) + j0;
    int w = threadIdx.x / pd, d = threadIdx.x % pd;
    const int dw = 256 / pd, dd = 256 % pd;
    while (w < JT + ROWS) {
      const int r = base + w;
      s.Pt[d * PWP + w] = (r >= 0 && r < 2 * a.T - 1) ? a.pos[(long)r * a.H * pd + h * pd + d] : 0.f;
      w += dw;
      d += dd;
      if (d >= pd) {
        d -= pd;
        ++w;
      }
    }
  }
}

// scores of 4 rows (ib..ib+3) x NM keys per lane for the staged tile
template <int NM>
__device__ __forceinline__ void tile_scores(const AttnArgs& a, const Smem& s, int b, int i0, int ib,
                                            int j0, int lane, float acc[4][NM]) {
  const int qd = a.qd, pd = a.pd;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[r][m] = 0.f;
  for (int d = 0; d < qd; ++d) {
    float kv[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) kv[m] = s.Kt[d * JTP + lane + 64 * m];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float qv = s.Q[(ib - i0 + r) * MAXQD + d];
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[r][m] = fmaf(qv, kv[m], acc[r][m]);
    }
  }
  if (a.pos) {
    for (int d = 0; d < pd; ++d) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = s.P[(ib - i0 + r) * MAXPD + d];
        const int off = (ROWS - 1) - (ib - i0 + r) + lane;
#pragma unroll
        for (int m = 0; m < NM; ++m)
          acc[r][m] = fmaf(pv, s.Pt[d * PWP + off + 64 * m], acc[r][m]);
      }
    }
  }
  // masks (-1000 replaces the score) and out-of-range keys (-inf: no contribution)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = ib + r;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int j = j0 + lane + 64 * m;
      if (j >= a.T || i >= a.T) {
        acc[r][m] = S2T_NEG_INF;
      } else if ((a.kpm && a.kpm[(long)b * a.T + j]) || (a.amask && a.amask[(long)i * a.T + j])) {
        acc[r][m] = -1000.f;
      }
    }
  }
}

constexpr int MAXCD = 32;  // concatenated value dims of the deferred-dW factors

__device__ __forceinline__ int pair_dims(const AttnArgs& a) {
  return (a.pdO[0] ? a.pdv[0] : 0) + (a.pdO[1] ? a.pdv[1] : 0);
}
// dst[r][k] = src_c[t0 + r, b, h*dv_c + d] for the concatenated (c,d) index k
__device__ __forceinline__ void stage_pairs(const AttnArgs& a, const float* const src[2], int b,
                                            int h, int t0, int nrows, float (*dst)[MAXCD + 1]) {
  int k0 = 0;
  for (int c = 0; c < 2; ++c) {
    if (!a.pdO[c]) continue;
    const int dv = a.pdv[c];
    const long ld = (long)a.H * dv;
    for (int idx = threadIdx.x; idx < nrows * dv; idx += blockDim.x) {
      const int r = idx / dv, d = idx % dv;
      const int t = t0 + r;
      dst[r][k0 + d] = (t < a.T) ? src[c][((long)t * a.B + b) * ld + h * dv + d] : 0.f;
    }
    k0 += dv;
  }
}

__device__ __forceinline__ void stage_rows(const AttnArgs& a, const Smem& s, int b, int h, int i0) {
  for (int idx = threadIdx.x; idx < ROWS * a.qd; idx += blockDim.x) {
    const int r = idx / a.qd, d = idx % a.qd;
    s.Q[r * MAXQD + d] = (i0 + r < a.T) ? q_row(a, i0 + r, b, h)[d] : 0.f;
  }
  for (int idx = threadIdx.x; idx < ROWS * a.pd; idx += blockDim.x) {
    const int r = idx / a.pd, d = idx % a.pd;
    s.P[r * MAXPD + d] = (i0 + r < a.T) ? p_row(a, i0 + r, b, h)[d] : 0.f;
  }
}

template <int NM>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs a, float* __restrict__ W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Smem s = carve<NM>(smem_raw, a.qd, a.pd);
  const int i0 = blockIdx.x * ROWS, b = blockIdx.y, h = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (a.T + JT - 1) / JT;
  stage_rows(a, s, b, h, i0);
  float* Wb = W + ((long)h * a.B + b) * a.T * a.T;
  float rmax[4][4], rsum[4][4];   // [group][row] running statistics (multi-tile case)
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      rmax[g][r] = S2T_NEG_INF;
      rsum[g][r] = 0.f;
    }
  for (int pass = 0; pass < (ntiles == 1 ? 1 : 2); ++pass) {
    for (int tile = 0; tile < ntiles; ++tile) {
      const int j0 = tile * JT;
      __syncthreads();
      stage_tile<NM>(a, s, b, h, i0, j0);
      __syncthreads();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ib = i0 + wave * 16 + g * 4;
        if (ib >= a.T) continue;
        float acc[4][NM];
        tile_scores<NM>(a, s, b, i0, ib, j0, lane, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (pass == 0) {
            float m = acc[r][0];
#pragma unroll
            for (int q = 1; q < NM; ++q) m = fmaxf(m, acc[r][q]);
            m = wave_max(m);
            const float mnew = fmaxf(rmax[g][r], m);
            float e = 0.f;
#pragma unroll
            for (int q = 0; q < NM; ++q) e += (acc[r][q] == S2T_NEG_INF) ? 0.f : __expf(acc[r][q] - mnew);
            e = wave_sum(e);
            rsum[g][r] = rsum[g][r] * ((rmax[g][r] == S2T_NEG_INF) ? 0.f : __expf(rmax[g][r] - mnew)) + e;
            rmax[g][r] = mnew;
          }
          if (pass == 1 || ntiles == 1) {
            const int i = ib + r;
            if (i < a.T) {
              const float inv = 1.f / rsum[g][r], mx = rmax[g][r];
#pragma unroll
              for (int q = 0; q < NM; ++q) {
                const int j = j0 + lane + 64 * q;
                if (j < a.T) Wb[(long)i * a.T + j] = __expf(acc[r][q] - mx) * inv;
              }
            }
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------- forward on the f32 MFMA
// Workgroup = 128 query rows of one (b,h): 4 waves x one 32-row strip each.  The whole key
// matrix of the (b,h) (T <= 512 keys x qd) is staged ONCE in LDS, k-contiguous with a 4-float
// pad (conflict-free ds_read_b128 fragments); each wave keeps its strip of scores
// (32 rows x T) in MFMA accumulators (NT tiles x 16 registers), adds the position term from an
// LDS window of linear_pos rows (one ds_read_b128 = the 4 position dims of one offset), applies
// the masks, does the softmax with half-wave shuffles and writes W once.  Lane = key column
// (lane & 31), register r = query row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the strip.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int MF_ROWS = 128;
constexpr int MF_KLD = MAXQD;   // unpadded; 16-byte chunks are XOR-swizzled by (row & 7)

// SPLIT = 2: two waves share a strip (each NT of the 2 NT key tiles) and combine their row
// statistics through LDS -- keeps the accumulators of a 512-key strip within the register file.
template <int NT, int SPLIT, bool HAS_POS, bool HAS_AM>
__global__ __launch_bounds__(256 * SPLIT, (NT >= 8 && SPLIT == 1) ? 2 : 1)
void attn_fwd_mfma_kernel(AttnArgs a, float* __restrict__ W) {
  constexpr int KT = NT * SPLIT;                                // key tiles staged
  constexpr int NTH = 256 * SPLIT;                              // SPLIT waves per 32-row strip
  constexpr int WG_ROWS = MF_ROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sK = reinterpret_cast<float*>(smem_raw);               // [KT*32][MF_KLD]
  float4* sPos = reinterpret_cast<float4*>(sK + KT * 32 * MF_KLD);   // [KT*32 + MF_ROWS]
  float4* sP = sPos + (KT * 32 + MF_ROWS);                      // [MF_ROWS]
  float* sStat = reinterpret_cast<float*>(sP + MF_ROWS);        // [waves][32 rows]
  unsigned* sMask = reinterpret_cast<unsigned*>(sStat + 8 * 32);   // AM_BITS: [MF_ROWS][KT] bit rows
  // (short sequences keep the byte reads: at T = 124 the staging pass cost more than it saved)
  constexpr bool AM_BITS = HAS_AM && KT >= 8;
  const int T = a.T, qd = a.qd, pd = a.pd;
  const int i0 = blockIdx.x * WG_ROWS, b = blockIdx.y, h = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = lane & 31, hi = lane >> 5;
  const long rs = (long)a.B * (a.H * (2 * qd + pd));           // floats between frames
  // ---- stage K (zero rows past T, zero dims past qd are never read), pos window, p rows
  {
    // all of a thread's global loads are issued before the first LDS store (the key rows of one
    // (b,h) are 128-byte pieces B*Dp floats apart: latency-bound unless many are in flight)
    const float* kb = k_row(a, 0, b, h);
    const int v4 = qd >> 2;                                     // float4 per key row (<= 8)
    const int jr = threadIdx.x >> 3, c = threadIdx.x & 7;
    constexpr int JR = NTH / 8, KI = KT * 32 / JR;              // key rows per pass, passes
    float4 kv[KI];
#pragma unroll
    for (int it = 0; it < KI; ++it) {
      const int j = jr + JR * it;
      kv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < T && c < v4) kv[it] = *reinterpret_cast<const float4*>(kb + (long)j * rs + 4 * c);
    }
    // window entry w <-> relative index rel = (T-1) - (i0 + MF_ROWS-1) + w  (w = 127 - il + j)
    const int base = (T - 1) - (i0 + MF_ROWS - 1);
    constexpr int NW = (KT * 32 + MF_ROWS + NTH - 1) / NTH;
    float4 wv[NW];
    const bool vec4 = pd == 4 && ((reinterpret_cast<uintptr_t>(a.pos) & 15) == 0);
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int w = threadIdx.x + NTH * it;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = base + w;
      if (HAS_POS && w < KT * 32 + MF_ROWS && r >= 0 && r < 2 * T - 1) {
        const float* pp = a.pos + (long)r * a.H * pd + h * pd;
        if (vec4) {
          v = *reinterpret_cast<const float4*>(pp);
        } else {
          v.x = pp[0];
          if (pd > 1) v.y = pp[1];
          if (pd > 2) v.z = pp[2];
          if (pd > 3) v.w = pp[3];
        }
      }
      wv[it] = v;
    }
    float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_POS && threadIdx.x < MF_ROWS && i0 + (int)threadIdx.x < T) {
      const float* pp = p_row(a, i0 + threadIdx.x, b, h);
      pq.x = pp[0];
      if (pd > 1) pq.y = pp[1];
      if (pd > 2) pq.z = pp[2];
      if (pd > 3) pq.w = pp[3];
    }
#pragma unroll
    for (int it = 0; it < KI; ++it)
      if (c < v4)
        *reinterpret_cast<float4*>(sK + (jr + JR * it) * MF_KLD + 4 * (c ^ (jr & 7))) = kv[it];
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int w = threadIdx.x + NTH * it;
      if (w < KT * 32 + MF_ROWS) sPos[w] = wv[it];
    }
    if (threadIdx.x < MF_ROWS) sP[threadIdx.x] = pq;
    if (AM_BITS) {
      // the attention mask of the workgroup's 128 rows as bits (a word per 32-key tile): the byte
      // gathers used to sit INSIDE the softmax -- 16 x NT dependent global loads per lane (chunked
      // training: 281 us against 155 unmasked at T = 495); here they are coalesced row reads +
      // ballots before the barrier, and the softmax reads LDS words
      constexpr int NWV = NTH / 64, NQ = (KT + 1) / 2;
      for (int il = wave; il < MF_ROWS; il += 2 * NWV) {        // two rows' loads in flight
        const unsigned char* am0 = a.amask + (long)min(i0 + il, T - 1) * T;
        const unsigned char* am1 = a.amask + (long)min(i0 + il + NWV, T - 1) * T;
        unsigned char mb0[NQ], mb1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          mb0[q] = am0[min(64 * q + lane, T - 1)];
          mb1[q] = am1[min(64 * q + lane, T - 1)];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const unsigned long long b0 = __ballot(mb0[q] != 0 && 64 * q + lane < T);
          const unsigned long long b1 = __ballot(mb1[q] != 0 && 64 * q + lane < T);
          if (lane == 0) {
            sMask[il * KT + 2 * q] = (unsigned)b0;
            sMask[(il + NWV) * KT + 2 * q] = (unsigned)b1;
            if (2 * q + 1 < KT) {
              sMask[il * KT + 2 * q + 1] = (unsigned)(b0 >> 32);
              sMask[(il + NWV) * KT + 2 * q + 1] = (unsigned)(b1 >> 32);
            }
          }
        }
      }
    }
  }
  // ---- this lane's query fragments: row (strip row lo), dims 8 s + 4 hi .. + 3
  const int strip = wave / SPLIT, half = wave % SPLIT;
  const int iw = i0 + strip * 32;                               // first row of the strip
  float4 qf[MAXQD / 8];
  {
    const int i = min(iw + lo, T - 1);
    const float* qp = q_row(a, i, b, h);
#pragma unroll
    for (int s8 = 0; s8 < MAXQD / 8; ++s8)
      qf[s8] = (8 * s8 < qd) ? *reinterpret_cast<const float4*>(qp + 8 * s8 + 4 * hi)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  if (SPLIT == 1 && iw >= T) return;                            // strip entirely past the end
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s8 = 0; s8 < MAXQD / 8; ++s8) {
      if (8 * s8 < qd) {
        const float4 kf = *reinterpret_cast<const float4*>(
            sK + (32 * (half * NT + t) + lo) * MF_KLD + 4 * ((2 * s8 + hi) ^ (lo & 7)));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s8].x, kf.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s8].y, kf.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s8].z, kf.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[s8].w, kf.w, acc[t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // keep the next tile's fragment reads behind these MFMAs
  }
  // ---- position term + masks, row statistics
  const unsigned char* kpm = a.kpm ? a.kpm + (long)b * T : nullptr;
  unsigned pad_bits = 0;                                        // bit t: key 32 t + lo is padding
  if (kpm) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = 32 * (half * NT + t) + lo;
      if (j < T && kpm[j]) pad_bits |= 1u << t;
    }
  }
  float* Wb = W + ((long)h * a.B + b) * T * T;
  // row by row: the row's position vector is read once, stores walk one row pointer
  const int ilb = strip * 32 + 4 * hi;                          // row of register 0
  float rmax[16], rsum[16];
  bool big = false;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int il = ilb + (r & 3) + 8 * (r >> 2);
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_POS) pv = sP[il];
    const float4* prow = sPos + (MF_ROWS - 1) - il + 32 * half * NT + lo;
    const unsigned* mrow = AM_BITS ? sMask + il * KT + half * NT : nullptr;
    const unsigned char* am = (HAS_AM && !AM_BITS) ? a.amask + (long)min(i0 + il, T - 1) * T : nullptr;
    float m = S2T_NEG_INF;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = 32 * (half * NT + t) + lo;
      float x = acc[t][r];
      if (HAS_POS) {
        const float4 e = prow[32 * t];
        x = fmaf(pv.x, e.x, x);
        x = fmaf(pv.y, e.y, x);
        x = fmaf(pv.z, e.z, x);
        x = fmaf(pv.w, e.w, x);
      }
      big |= (fabsf(x) > a.pen_limit) && j < T && (i0 + il) < T;
      bool masked = (pad_bits >> t) & 1u;
      if (AM_BITS) masked = masked || ((mrow[t] >> lo) & 1u);
      else if (HAS_AM) masked = masked || am[min(j, T - 1)] != 0;
      x = masked ? -1000.f : x;
      x = (j >= T) ? S2T_NEG_INF : x;
      acc[t][r] = x;
      m = fmaxf(m, x);
    }
    rmax[r] = m;
  }
  if (a.pen_flag != nullptr && big)
    __hip_atomic_store(a.pen_flag, 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) rmax[r] = fmaxf(rmax[r], __shfl_xor(rmax[r], o, 64));
  }
  if (SPLIT == 2) {                                             // combine with the partner wave
    if (lo == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sStat[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = rmax[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
      rmax[r] = fmaxf(rmax[r], sStat[(wave ^ 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]);
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float e = (32 * (half * NT + t) + lo >= T) ? 0.f : __expf(acc[t][r] - rmax[r]);
      acc[t][r] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    rsum[r] = sum;
  }
  if (SPLIT == 2) {
    if (lo == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sStat[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi] = rsum[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) rsum[r] += sStat[(wave ^ 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + ilb + (r & 3) + 8 * (r >> 2);
    if (i < T) {
      const float inv = 1.f / rsum[r];
      float* wr = Wb + (long)i * T + 32 * half * NT + lo;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (32 * (half * NT + t) + lo < T) wr[32 * t] = acc[t][r] * inv;
    }
  }
}

template <int KT>
inline size_t attn_mfma_smem() {
  return sizeof(float) * ((size_t)KT * 32 * MF_KLD + 4 * ((size_t)KT * 32 + MF_ROWS) + 4 * MF_ROWS +
                          8 * 32);
}

// ---------------------------------------------------------------- backward
// dS_ij = W_ij (dW_ij - delta_i) on unmasked entries, delta_i = sum_j W_ij dW_ij.
// bwd_q: workgroup = (64 rows, b, h): delta, dq_i = sum_j dS_ij k_j, dp_i = sum_j dS_ij pos[rel].
//        thread = (row = tid/4, dgrp = tid%4) accumulates qd/4 dims of dq and one dim of dp.
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnArgs a, const float* __restrict__ W,
                                                         int delta_given,
                                                         float* __restrict__ delta,
                                                         float* __restrict__ dqkp) {
  constexpr int JC = 64;   // key chunk
  __shared__ float s_dS[ROWS][JC + 1];
  __shared__ float s_K[JC][MAXQD + 1];
  __shared__ float s_pos[MAXPD][JC + ROWS];
  __shared__ float s_delta[ROWS];
  __shared__ float s_dO[ROWS][MAXCD + 1];
  __shared__ float s_Vc[JC][MAXCD + 1];
  const float* dW = a.dW;
  const int cd = pair_dims(a);
  const int i0 = blockIdx.x * ROWS, b = blockIdx.y, h = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* Wb = W + ((long)h * a.B + b) * a.T * a.T;
  const float* dWb = dW ? dW + ((long)h * a.B + b) * a.T * a.T : nullptr;
  const float* dW0b = (a.dW0 && h == 0) ? a.dW0 + (long)b * a.T * a.T : nullptr;
  if (delta_given) {
    for (int r = threadIdx.x; r < ROWS; r += 256)
      s_delta[r] = (i0 + r < a.T) ? delta[((long)h * a.B + b) * a.T + i0 + r] : 0.f;
  } else {
    // pass 0: delta for the 64 rows from the materialised dW (wave per row, coalesced)
    for (int r = wave; r < ROWS; r += 4) {
      const int i = i0 + r;
      float acc = 0.f;
      if (i < a.T)
        for (int j = lane; j < a.T; j += 64) acc = fmaf(Wb[(long)i * a.T + j], dWb[(long)i * a.T + j], acc);
      acc = wave_sum(acc);
      if (lane == 0) {
        s_delta[r] = acc;
        if (i < a.T) delta[((long)h * a.B + b) * a.T + i] = acc;
      }
    }
  }
  if (cd) stage_pairs(a, a.pdO, b, h, i0, ROWS, s_dO);
  __syncthreads();
  const int row = tid >> 2, dg = tid & 3;
  const int qd = a.qd, pd = a.pd;
  const int dper = (qd + 3) / 4;   // dims of dq per thread (<= 8)
  float accq[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) accq[d] = 0.f;
  float accp0 = 0.f, accp1 = 0.f;  // pos dims dg and dg+4
  for (int j0 = 0; j0 < a.T; j0 += JC) {
    __syncthreads();
    if (cd) {
      stage_pairs(a, a.pV, b, h, j0, JC, s_Vc);
      __syncthreads();
    }
    // stage dS chunk (coalesced over j), K chunk, pos window
    for (int idx = tid; idx < ROWS * JC; idx += 256) {
      const int r = idx / JC, jj = idx % JC;
      const int i = i0 + r, j = j0 + jj;
      float v = 0.f;
      if (i < a.T && j < a.T) {
        const bool masked = (a.kpm && a.kpm[(long)b * a.T + j]) || (a.amask && a.amask[(long)i * a.T + j]);
        if (!masked) {
          const float w = Wb[(long)i * a.T + j];
          float dw = dWb ? dWb[(long)i * a.T + j] : 0.f;
          if (dW0b) dw += dW0b[(long)i * a.T + j];
          for (int d = 0; d < cd; ++d) dw = fmaf(s_dO[r][d], s_Vc[jj][d], dw);
          v = w * (dw - s_delta[r]);
        }
      }
      s_dS[r][jj] = v;
    }
    for (int idx = tid; idx < JC * qd; idx += 256) {
      const int jj = idx / qd, d = idx % qd;
      s_K[jj][d] = (j0 + jj < a.T) ? k_row(a, j0 + jj, b, h)[d] : 0.f;
    }
    if (a.pos) {
      const int base = (a.T - 1) - (i0 + ROWS - 1) + j0;
      for (int idx = tid; idx < (JC + ROWS) * pd; idx += 256) {
        const int w = idx / pd, d = idx % pd;
        const int r = base + w;
        s_pos[d][w] = (r >= 0 && r < 2 * a.T - 1) ? a.pos[(long)r * a.H * pd + h * pd + d] : 0.f;
      }
    }
    __syncthreads();
    for (int jj = 0; jj < JC; ++jj) {
      const float ds = s_dS[row][jj];
#pragma unroll
      for (int d = 0; d < 8; ++d)
        if (d < dper) accq[d] = fmaf(ds, s_K[jj][dg * dper + d], accq[d]);
      if (a.pos) {
        const int w = (ROWS - 1) - row + jj;
        if (dg < pd) accp0 = fmaf(ds, s_pos[dg][w], accp0);
        if (dg + 4 < pd) accp1 = fmaf(ds, s_pos[dg + 4][w], accp1);
      }
    }
  }
  const int i = i0 + row;
  if (i < a.T) {
    float* o = dqkp + ((long)i * a.B + b) * (a.H * (2 * qd + pd));
    for (int d = 0; d < dper; ++d)
      if (dg * dper + d < qd) o[h * qd + dg * dper + d] = accq[d];
    if (dg < pd) o[2 * a.H * qd + h * pd + dg] = a.pos ? accp0 : 0.f;
    if (dg + 4 < pd) o[2 * a.H * qd + h * pd + dg + 4] = a.pos ? accp1 : 0.f;
  }
}

// bwd_k: workgroup = (64 keys, b, h): dk_j = sum_i dS_ij q_i ; dpos[rel] += sum dS_ij p_i
__global__ __launch_bounds__(256) void attn_bwd_k_kernel(AttnArgs a, const float* __restrict__ W,
                                                         const float* __restrict__ delta,
                                                         float* __restrict__ dqkp,
                                                         float* __restrict__ dpos) {
  constexpr int IC = 64;   // query chunk
  __shared__ float s_dS[IC][ROWS + 1];       // [i][j]
  __shared__ float s_Q[IC][MAXQD + 1];
  __shared__ float s_P[IC][MAXPD];
  __shared__ float s_dOc[IC][MAXCD + 1];
  __shared__ float s_Vk[ROWS][MAXCD + 1];
  const float* dW = a.dW;
  const int cd = pair_dims(a);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s_acc = reinterpret_cast<float*>(smem_raw);   // [pd][T + ROWS]: dpos for rel = j0 + u
  const int j0 = blockIdx.x * ROWS, b = blockIdx.y, h = blockIdx.z;
  const int tid = threadIdx.x;
  const float* Wb = W + ((long)h * a.B + b) * a.T * a.T;
  const float* dWb = dW ? dW + ((long)h * a.B + b) * a.T * a.T : nullptr;
  const float* dW0b = (a.dW0 && h == 0) ? a.dW0 + (long)b * a.T * a.T : nullptr;
  const float* dl = delta + ((long)h * a.B + b) * a.T;
  if (cd) stage_pairs(a, a.pV, b, h, j0, ROWS, s_Vk);
  const int col = tid >> 2, dg = tid & 3;
  const int qd = a.qd, pd = a.pd;
  const int dper = (qd + 3) / 4;
  float acck[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) acck[d] = 0.f;
  const int accw = a.T + ROWS;
  if (dpos)
    for (int idx = tid; idx < accw * pd; idx += 256) s_acc[idx] = 0.f;
  for (int i0 = 0; i0 < a.T; i0 += IC) {
    __syncthreads();
    if (cd) {
      stage_pairs(a, a.pdO, b, h, i0, IC, s_dOc);
      __syncthreads();
    }
    for (int idx = tid; idx < IC * ROWS; idx += 256) {
      const int ii = idx / ROWS, jj = idx % ROWS;
      const int i = i0 + ii, j = j0 + jj;
      float v = 0.f;
      if (i < a.T && j < a.T) {
        const bool masked = (a.kpm && a.kpm[(long)b * a.T + j]) || (a.amask && a.amask[(long)i * a.T + j]);
        if (!masked) {
          float dw = dWb ? dWb[(long)i * a.T + j] : 0.f;
          if (dW0b) dw += dW0b[(long)i * a.T + j];
          for (int d = 0; d < cd; ++d) dw = fmaf(s_dOc[ii][d], s_Vk[jj][d], dw);
          v = Wb[(long)i * a.T + j] * (dw - dl[i]);
        }
      }
      s_dS[ii][jj] = v;
    }
    for (int idx = tid; idx < IC * qd; idx += 256) {
      const int ii = idx / qd, d = idx % qd;
      s_Q[ii][d] = (i0 + ii < a.T) ? q_row(a, i0 + ii, b, h)[d] : 0.f;
    }
    for (int idx = tid; idx < IC * pd; idx += 256) {
      const int ii = idx / pd, d = idx % pd;
      s_P[ii][d] = (i0 + ii < a.T) ? p_row(a, i0 + ii, b, h)[d] : 0.f;
    }
    __syncthreads();
    for (int ii = 0; ii < IC; ++ii) {
      const float ds = s_dS[ii][col];
#pragma unroll
      for (int d = 0; d < 8; ++d)
        if (d < dper) acck[d] = fmaf(ds, s_Q[ii][dg * dper + d], acck[d]);
    }
    if (dpos) {
      // diagonals of the chunk: w = (IC-1) - ii + jj in [0, IC+ROWS-2]; thread per (w, dim)
      for (int idx = tid; idx < (IC + ROWS - 1) * pd; idx += 256) {
        const int w = idx / pd, d = idx % pd;
        float acc = 0.f;
        const int ii_lo = max(0, (IC - 1) - w), ii_hi = min(IC - 1, (IC - 1) - w + ROWS - 1);
        for (int ii = ii_lo; ii <= ii_hi; ++ii) acc = fmaf(s_dS[ii][w - (IC - 1) + ii], s_P[ii][d], acc);
        // u = rel - j0 = (T-1) - (i0 + IC-1) + w ; one thread per (u,d): no LDS atomics needed
        const int u = (a.T - 1) - (i0 + IC - 1) + w;
        if (u >= 0 && u < accw) s_acc[d * accw + u] += acc;
      }
    }
  }
  if (dpos) {
    __syncthreads();
    for (int idx = tid; idx < accw * pd; idx += 256) {
      const int d = idx / accw, u = idx % accw;
      const int r = j0 + u;
      const float v = s_acc[idx];
      if (r < 2 * a.T - 1 && v != 0.f) atomicAdd(&dpos[(long)r * a.H * pd + h * pd + d], v);
    }
  }
  const int j = j0 + col;
  if (j < a.T) {
    float* o = dqkp + ((long)j * a.B + b) * (a.H * (2 * qd + pd)) + a.H * qd + h * qd;
    for (int d = 0; d < dper; ++d)
      if (dg * dper + d < qd) o[dg * dper + d] = acck[d];
  }
}

// ---------------------------------------------------------------- backward on the matrix cores
// Per 32x32 score tile:  dW = dOcat . Vcat^T  (+ materialised terms),  dS = W o (dW - delta_i),
//   dq += dS . K   and   dk += dS^T . Q   on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate).
// Accumulator layout of a 32x32 tile: column = lane&31, rows (r&3)+8(r>>2)+4(lane>>5) in the 16
// registers.  dS^T.Q takes the dS accumulator registers directly as the A operand (k permuted
// consistently with the B rows); dS.K needs the row on the lane and goes through a wave-private
// LDS tile, which the position-gradient sums then read by diagonals.
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// element k (0..cd-1) of the concatenated factor row t (dO_c or V_c rows)
__device__ __forceinline__ float pair_elem(const AttnArgs& a, const float* const src[2], int t,
                                           int b, int h, int k) {
  if (t >= a.T) return 0.f;
  if (a.pdO[0]) {
    if (k < a.pdv[0]) return src[0][((long)t * a.B + b) * ((long)a.H * a.pdv[0]) + h * a.pdv[0] + k];
    k -= a.pdv[0];
  }
  if (a.pdO[1] && k < a.pdv[1])
    return src[1][((long)t * a.B + b) * ((long)a.H * a.pdv[1]) + h * a.pdv[1] + k];
  return 0.f;
}

// XCD-aware block order for the kernels that walk one (b, h) slab of W tile by tile: workgroups are
// dealt to the 8 XCDs round-robin by linear id, so id -> (tile, slab) is chosen such that ALL tiles
// of a slab run on one XCD (slab % 8 = id % 8).  W rows are T floats apart -- not a multiple of a
// 128-byte line -- so a tile row straddles two lines and neighbouring tiles share one of them: on
// one XCD the shared line comes out of its L2, across XCDs each fetches it from HBM (PMC: 3.2x the
// algorithmic bytes for the backward pair before this mapping).  Returns false for padding ids.
__device__ __forceinline__ bool slab_tile(int ntile, int nslab, int& tile, int& slab) {
  const int lin = blockIdx.x, q = lin >> 3;
  tile = q % ntile;
  slab = (q / ntile) * 8 + (lin & 7);
  return slab < nslab;
}
static inline unsigned slab_grid(int ntile, int nslab) { return (unsigned)(((nslab + 7) / 8) * 8 * ntile); }

// 32x32 tile of a (T,T) matrix in accumulator layout, indices clamped (validity is in tile_ok)
__device__ __forceinline__ void load_tile16(const float* __restrict__ base, int T, int i0, int j0,
                                            int lo, int hi, float (&w)[16]) {
  // 32-bit offsets from a wave-uniform base (T*T < 2^31 checked by the launcher); rows beyond
  // T clamp to the last element
  const unsigned last = (unsigned)T * (unsigned)T - 1u;
  const unsigned o0 = (unsigned)(i0 + 4 * hi) * (unsigned)T + (unsigned)min(j0 + lo, T - 1);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned off = min(o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)T, last);
    w[r] = base[off];
  }
}

// bit r set <=> element (i0 + acc_row(r,hi), j0 + lo) takes part in the softmax
// mbits (optional, LDS): the attention mask of the tile's 32 rows as bit words -- row acc_row(r, hi)
// at mbits[acc_row * mstride], bit lo = key j0 + lo -- staged once per workgroup by the MFMA
// backward kernels (attn_stage_mask_*): the 16 byte gathers per tile sat inside the tile loop
// (chunked training: +200 us per T = 495 layer)
__device__ __forceinline__ unsigned tile_ok(const AttnArgs& a, int b, int i0, int j0, int lo,
                                            int hi, const unsigned* mbits = nullptr,
                                            int mstride = 0) {
  const int j = j0 + lo;
  const int jc = min(j, a.T - 1);
  bool jok = j < a.T;
  if (a.kpm) jok = jok && !a.kpm[(long)b * a.T + jc];
  unsigned ok = 0;
  if (mbits) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ar = acc_row(r, hi);
      const bool v = jok && i0 + ar < a.T && !((mbits[ar * mstride] >> lo) & 1u);
      ok |= (v ? 1u : 0u) << r;
    }
    return ok;
  }
  const unsigned last = (unsigned)a.T * (unsigned)a.T - 1u;
  const unsigned o0 = (unsigned)(i0 + 4 * hi) * (unsigned)a.T + (unsigned)jc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + acc_row(r, hi);
    bool v = jok && i < a.T;
    if (a.amask)
      v = v && !a.amask[min(o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)a.T, last)];
    ok |= (v ? 1u : 0u) << r;
  }
  return ok;
}

template <int NS>
__device__ __forceinline__ f32x16 ds_tile(const AttnArgs& a, const float* __restrict__ dWb,
                                          const float* __restrict__ dW0b, int b, int i0, int j0,
                                          int lo, int hi, const float* af, const float* bf,
                                          const float (&w)[16], f32x16 acc,
                                          const unsigned* mbits = nullptr, int mstride = 0) {
  // acc enters as -delta_i (row constants as the initial accumulator)
#pragma unroll
  for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
  if (dWb) {
    float t[16];
    load_tile16(dWb, a.T, i0, j0, lo, hi, t);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += t[r];
  }
  if (dW0b) {
    float t[16];
    load_tile16(dW0b, a.T, i0, j0, lo, hi, t);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += t[r];
  }
  const unsigned ok = tile_ok(a, b, i0, j0, lo, hi, mbits, mstride);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ((ok >> r) & 1u) ? w[r] * acc[r] : 0.f;
  return acc;
}

constexpr int AM_MAXT = 512;      // sequences up to here get the bit-staged mask (8 KB of LDS)


// dk: workgroup = (32 keys, b, h); the 4 waves split the query blocks and their partial
// sums meet in LDS.
template <int NS>
__global__ __launch_bounds__(256, 2) void attn_bwd_k_mfma_kernel(AttnArgs a,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ delta,
                                                              float* __restrict__ dqkp) {
  constexpr int NSA = NS > 0 ? NS : 1;
  __shared__ float s_red[4][32][33];   // first the per-wave dO_cat staging, then the dk partials
  int tile_, slab_;
  if (!slab_tile((a.T + 31) / 32, a.B * a.H, tile_, slab_)) return;
  const int j0 = tile_ * 32, b = slab_ % a.B, h = slab_ / a.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
  const float* Wb = W + ((long)h * a.B + b) * a.T * a.T;
  const float* dWb = a.dW ? a.dW + ((long)h * a.B + b) * a.T * a.T : nullptr;
  const float* dW0b = (a.dW0 && h == 0) ? a.dW0 + (long)b * a.T * a.T : nullptr;
  const float* dlb = delta + ((long)h * a.B + b) * a.T;
  const int qd = a.qd;
  float bf[NSA];   // V_cat[j0+lo][hi + 2s]
#pragma unroll
  for (int s = 0; s < NS; ++s) bf[s] = pair_elem(a, a.pV, j0 + lo, b, h, hi + 2 * s);
  f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nib = (a.T + 31) / 32;
  float wn[16];
  if (wave < nib) load_tile16(Wb, a.T, wave * 32, j0, lo, hi, wn);
  for (int ib = wave; ib < nib; ib += 4) {
    const int i0 = ib * 32;
    float af[NSA], qv[16];
    f32x16 ndl;
    if (NS > 0) {
      // dO_cat rows of this query block: coalesced load -> wave-private LDS -> A fragments
      constexpr int CDP = 2 * NSA;
#pragma unroll 4
      for (int q = 0; q < (32 * CDP + 63) / 64; ++q) {
        const int idx = lane + 64 * q;
        const int rr = idx / CDP, k = idx % CDP;
        if (rr < 32) s_red[wave][rr][k] = pair_elem(a, a.pdO, i0 + rr, b, h, k);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = s_red[wave][lo][hi + 2 * s];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    const unsigned Dp = (unsigned)(a.H * (2 * qd + a.pd));
    const unsigned qcol = (unsigned)b * Dp + (unsigned)(h * qd + min(lo, qd - 1));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned i = (unsigned)min(i0 + acc_row(r, hi), a.T - 1);   // rows beyond T meet dS = 0
      ndl[r] = -dlb[i];
      const float q = a.qkp[i * ((unsigned)a.B * Dp) + qcol];
      qv[r] = lo < qd ? q : 0.f;
    }
    f32x16 ds = ds_tile<NS>(a, dWb, dW0b, b, i0, j0, lo, hi, af, bf, wn, ndl);
    if (ib + 4 < nib) load_tile16(Wb, a.T, i0 + 128, j0, lo, hi, wn);
    // dk[j][d] += sum_i dS[i][j] q[i][d]: A = dS registers (k = query row), B[k][n = d]
#pragma unroll
    for (int s = 0; s < 16; ++s) z = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[s], qv[s], z, 0, 0, 0);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) s_red[wave][acc_row(r, hi)][lo] = z[r];
  __syncthreads();
  for (int idx = tid; idx < 32 * 32; idx += 256) {
    const int jj = idx >> 5, d = idx & 31;
    const int j = j0 + jj;
    if (j < a.T && d < qd) {
      const float v = s_red[0][jj][d] + s_red[1][jj][d] + s_red[2][jj][d] + s_red[3][jj][d];
      dqkp[((long)j * a.B + b) * (a.H * (2 * qd + a.pd)) + a.H * qd + h * qd + d] = v;
    }
  }
}

// dk, shared-operand form: workgroup = (128 keys, b, h), one 32-key tile per wave, ALL waves walk
// the same query blocks.  The kernel above is bound by the volume of L1 fills, not by the matrix
// pipe (T = 495: 248 us against a 48 us MFMA floor; without its in-loop global loads 108 us, with
// them prefetched a block ahead still 259 us): each of a slab's 16 key-tile workgroups re-reads the
// query-side rows (q, dO_cat, delta -- 9 of the 13 KB a tile needs) for its own use.  Here those
// rows are fetched ONCE per query block by the whole workgroup (coalesced, a block ahead in
// registers) into LDS and feed four tiles; only the W tile (and head 0's dW tile) stay per wave.
// Every wave owns its keys' dk accumulator for the whole loop: no cross-wave reduction, the
// result leaves as 128-byte row pieces.
template <int NS>
__global__ __launch_bounds__(256, 2) void attn_bwd_k3_mfma_kernel(AttnArgs a,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ delta,
                                                               float* __restrict__ dqkp,
                                                               float* __restrict__ zero_buf,
                                                               int zero_n) {
  constexpr int NSA = NS > 0 ? NS : 1;
  constexpr int CDP = 2 * NSA;                                          // <= 32
  __shared__ float s_Q[32][33];                                         // q[i0+ii][d]
  __shared__ float s_O[32][33];                                         // dO_cat[i0+ii][k]
  __shared__ float s_dl[32];
  // the position-gradient accumulator of the reduce pass that follows this launch is cleared
  // here (it is not read by this kernel): no fill launch per layer
  for (int i = blockIdx.x * 256 + threadIdx.x; i < zero_n; i += gridDim.x * 256) zero_buf[i] = 0.f;
  int tile_, slab_;
  if (!slab_tile((a.T + 127) / 128, a.B * a.H, tile_, slab_)) return;
  const int b = slab_ % a.B, h = slab_ / a.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
  const int j0 = tile_ * 128 + wave * 32;
  const bool live = j0 < a.T;
  const float* Wb = W + ((long)h * a.B + b) * a.T * a.T;
  const float* dWb = a.dW ? a.dW + ((long)h * a.B + b) * a.T * a.T : nullptr;
  const float* dW0b = (a.dW0 && h == 0) ? a.dW0 + (long)b * a.T * a.T : nullptr;
  const float* dlb = delta + ((long)h * a.B + b) * a.T;
  const int qd = a.qd;
  float bf[NSA];   // V_cat[j0+lo][hi + 2s]
#pragma unroll
  for (int s = 0; s < NS; ++s) bf[s] = pair_elem(a, a.pV, j0 + lo, b, h, hi + 2 * s);
  f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int nib = (a.T + 31) / 32;
  // attention mask of this wave's 32 keys, one word per query row (wave-private: no barrier)
  __shared__ unsigned s_mb[4][AM_MAXT];
  const bool mbit = a.amask != nullptr && a.T <= AM_MAXT;
  if (mbit && live) {
    const int jc = min(j0 + lo, a.T - 1);
    for (int i = 0; i < a.T; i += 16) {                // 8 loads (2 rows each) in flight
      unsigned char mbv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) mbv[q] = a.amask[(long)min(i + 2 * q + hi, a.T - 1) * a.T + jc];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned long long bal = __ballot(mbv[q] != 0);
        if (lane == 0 && i + 2 * q < a.T) {
          s_mb[wave][i + 2 * q] = (unsigned)bal;
          if (i + 2 * q + 1 < a.T) s_mb[wave][i + 2 * q + 1] = (unsigned)(bal >> 32);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  // register staging of the next query block
  float rq[4], ro[4], rd = 0.f, wn[16];
  auto fetch = [&](int i0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, ii = idx >> 5, d = idx & 31;
      rq[q] = (i0 + ii < a.T && d < qd) ? q_row(a, i0 + ii, b, h)[d] : 0.f;
      ro[q] = (NS > 0 && d < CDP) ? pair_elem(a, a.pdO, i0 + ii, b, h, d) : 0.f;
    }
    if (tid < 32) rd = dlb[min(i0 + tid, a.T - 1)];
    if (live) load_tile16(Wb, a.T, i0, j0, lo, hi, wn);
  };
  fetch(0);
  for (int ib = 0; ib < nib; ++ib) {
    const int i0 = ib * 32;
    __syncthreads();                                   // previous block's tiles fully consumed
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, ii = idx >> 5, d = idx & 31;
      s_Q[ii][d] = rq[q];
      s_O[ii][d] = ro[q];
    }
    if (tid < 32) s_dl[tid] = rd;
    __syncthreads();
    f32x16 ds;
    float qv[16];
    if (live) {
      float af[NSA];
      f32x16 ndl;
#pragma unroll
      for (int s = 0; s < NS; ++s) af[s] = s_O[lo][hi + 2 * s];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ndl[r] = -s_dl[acc_row(r, hi)];
        qv[r] = s_Q[acc_row(r, hi)][lo];               // zero beyond qd / T (staged so)
      }
      ds = ds_tile<NS>(a, dWb, dW0b, b, i0, j0, lo, hi, af, bf, wn, ndl,
                       mbit ? &s_mb[wave][min(i0, AM_MAXT - 32)] : nullptr, 1);
    }
    if (ib + 1 < nib) fetch(i0 + 32);                  // lands while this block's products run
    if (!live) continue;
    // dk[j][d] += sum_i dS[i][j] q[i][d]: A = dS registers (k = query row), B[k][n = d]
#pragma unroll
    for (int s = 0; s < 16; ++s) z = __builtin_amdgcn_mfma_f32_32x32x2f32(ds[s], qv[s], z, 0, 0, 0);
  }
  if (!live) return;
  const int Dp = a.H * (2 * qd + a.pd);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = j0 + acc_row(r, hi);
    if (j < a.T && lo < qd) dqkp[((long)j * a.B + b) * Dp + a.H * qd + h * qd + lo] = z[r];
  }
}

// dq, dp and the per-(b,h,query block) partial sums of dpos: workgroup = (128 query rows, b, h),
// one 32-row block per wave, loop over key blocks.  ws: [b][h][ib][(nj+1)*32][pd]; row u of
// query block ib belongs to rel = (T-1) - (32 ib + 31) + u.
template <int NS, int PD>
__global__ __launch_bounds__(256, 2) void attn_bwd_q_mfma_kernel(AttnArgs a,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ delta,
                                                              float* __restrict__ dqkp,
                                                              float* __restrict__ ws) {
  constexpr int NSA = NS > 0 ? NS : 1;
  constexpr int NPQ = (160 * PD + 255) / 256;
  __shared__ float s_K[32][33];                                        // K[j0+jj][d]
  __shared__ float s_V[32][33];                                        // V_cat[j0+jj][k]
  __shared__ __attribute__((aligned(16))) float s_pos[160][PD];        // pos window
  __shared__ float s_t[4][32][33];                                     // per-wave dS tile [i][j]
  __shared__ __attribute__((aligned(16))) float s_P[4][32][PD];        // per-wave p rows
  int tile_, slab_;
  if (!slab_tile((a.T + 127) / 128, a.B * a.H, tile_, slab_)) return;
  const int ib0 = tile_ * 128, b = slab_ % a.B, h = slab_ / a.B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
  const int i0 = ib0 + wave * 32;
  const bool live = i0 < a.T;
  const float* Wb = W + ((long)h * a.B + b) * a.T * a.T;
  const float* dWb = a.dW ? a.dW + ((long)h * a.B + b) * a.T * a.T : nullptr;
  const float* dW0b = (a.dW0 && h == 0) ? a.dW0 + (long)b * a.T * a.T : nullptr;
  const float* dlb = delta + ((long)h * a.B + b) * a.T;
  const int qd = a.qd, pd = a.pd;
  const int nj = (a.T + 31) / 32;
  const bool want_pos = a.pos != nullptr;
  float af[NSA];
  // -delta of the wave's 32 rows lives in LDS and is re-read per key block: 16 fewer registers
  // held across the loop (the kernel sits at its 256-VGPR budget and was spilling)
  __shared__ float s_ndl[4][32];
#pragma unroll
  for (int s = 0; s < NS; ++s) af[s] = pair_elem(a, a.pdO, i0 + lo, b, h, hi + 2 * s);
  if (lane < 32) s_ndl[wave][lane] = -dlb[min(i0 + lane, a.T - 1)];
  for (int idx = lane; idx < 32 * PD; idx += 64) {
    const int rr = idx / PD, d = idx % PD;
    s_P[wave][rr][d] = (i0 + rr < a.T && d < pd) ? p_row(a, i0 + rr, b, h)[d] : 0.f;
  }
  // attention mask of this wave's 32 query rows, a word per 32-key block (wave-private)
  __shared__ unsigned s_mq[4][32][AM_MAXT / 32];
  const bool mbit = a.amask != nullptr && a.T <= AM_MAXT;
  if (mbit && live) {
    for (int rr = 0; rr < 32; rr += 2) {               // 2 rows x 8 loads of 64 keys in flight
      const unsigned char* am0 = a.amask + (long)min(i0 + rr, a.T - 1) * a.T;
      const unsigned char* am1 = a.amask + (long)min(i0 + rr + 1, a.T - 1) * a.T;
      unsigned char m0[AM_MAXT / 64], m1[AM_MAXT / 64];
#pragma unroll
      for (int q = 0; q < AM_MAXT / 64; ++q) {
        m0[q] = am0[min(64 * q + lane, a.T - 1)];
        m1[q] = am1[min(64 * q + lane, a.T - 1)];
      }
#pragma unroll
      for (int q = 0; q < AM_MAXT / 64; ++q) {
        const unsigned long long b0 = __ballot(m0[q] != 0), b1 = __ballot(m1[q] != 0);
        if (lane == 0) {
          s_mq[wave][rr][2 * q] = (unsigned)b0;
          s_mq[wave][rr][2 * q + 1] = (unsigned)(b0 >> 32);
          s_mq[wave][rr + 1][2 * q] = (unsigned)b1;
          s_mq[wave][rr + 1][2 * q + 1] = (unsigned)(b1 >> 32);
        }
      }
    }
  }
  f32x16 zq = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float dp[PD], dacc[PD];
#pragma unroll
  for (int d = 0; d < PD; ++d) dp[d] = dacc[d] = 0.f;
  // register staging of the next key block
  float rk[4], rv[4], rp[NPQ], wn[16];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, jj = idx >> 5, d = idx & 31;
      rk[q] = (j0 + jj < a.T && d < qd) ? k_row(a, j0 + jj, b, h)[d] : 0.f;
      rv[q] = (NS > 0 && d < 2 * NS) ? pair_elem(a, a.pV, j0 + jj, b, h, d) : 0.f;
    }
    if (want_pos) {
      // window entry w <-> rel (T-1) - (ib0 + 127) + j0 + w
      const int base = (a.T - 1) - (ib0 + 127) + j0;
#pragma unroll
      for (int q = 0; q < NPQ; ++q) {
        const int idx = tid + 256 * q, w = idx / PD, d = idx % PD;
        const int rel = base + w;
        rp[q] = (w < 160 && d < pd && rel >= 0 && rel < 2 * a.T - 1)
                    ? a.pos[(long)rel * a.H * pd + h * pd + d] : 0.f;
      }
    }
    if (live) load_tile16(Wb, a.T, i0, j0, lo, hi, wn);
  };
  fetch(0);
  float* wsb = nullptr;
  if (ws && live)
    wsb = ws + ((((long)b * a.H + h) * nj + (i0 >> 5)) * (nj + 1)) * 32 * pd;
  for (int jb = 0; jb < nj; ++jb) {
    const int j0 = jb * 32;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, jj = idx >> 5, d = idx & 31;
      s_K[jj][d] = rk[q];
      s_V[jj][d] = rv[q];
    }
    if (want_pos) {
#pragma unroll
      for (int q = 0; q < NPQ; ++q) {
        const int idx = tid + 256 * q;
        if (idx < 160 * PD) s_pos[idx / PD][idx % PD] = rp[q];
      }
    }
    __syncthreads();
    f32x16 ds;
    if (live) {
      float bf[NSA];
#pragma unroll
      for (int s = 0; s < NS; ++s) bf[s] = s_V[lo][hi + 2 * s];
      f32x16 ndl;
#pragma unroll
      for (int r = 0; r < 16; ++r) ndl[r] = s_ndl[wave][acc_row(r, hi)];
      ds = ds_tile<NS>(a, dWb, dW0b, b, i0, j0, lo, hi, af, bf, wn, ndl,
                       mbit ? &s_mq[wave][0][min(jb, AM_MAXT / 32 - 1)] : nullptr, AM_MAXT / 32);
    }
    if (jb + 1 < nj) fetch(j0 + 32);   // lands while this tile's products run
    if (!live) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) s_t[wave][acc_row(r, hi)][lo] = ds[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float at[16];   // dS[i = lo][j = hi + 2s]
#pragma unroll
    for (int s = 0; s < 16; ++s) at[s] = s_t[wave][lo][hi + 2 * s];
#pragma unroll
    for (int s = 0; s < 16; ++s)
      zq = __builtin_amdgcn_mfma_f32_32x32x2f32(at[s], s_K[hi + 2 * s][lo], zq, 0, 0, 0);
    if (want_pos) {
      // dp[i][d] += dS[i][j] pos[(T-1) - i + j][d]; window row = 127 - (i - ib0) + (j - j0)
      const int w0 = 127 - (wave * 32 + lo) + hi;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int d = 0; d < PD; ++d) dp[d] = fmaf(at[s], s_pos[w0 + 2 * s][d], dp[d]);
      }
      // dpos partial: lane u owns the diagonal j - i + 31 = u of this tile
#pragma unroll 4
      for (int ii = 0; ii < 32; ++ii) {
        const int jj = ii + lane - 31;
        float v = s_t[wave][ii][min(max(jj, 0), 31)];
        v = (jj >= 0 && jj < 32) ? v : 0.f;
#pragma unroll
        for (int d = 0; d < PD; ++d) dacc[d] = fmaf(v, s_P[wave][ii][d], dacc[d]);
      }
      if (wsb) {
        if (lane < 32)
          for (int d = 0; d < pd; ++d) wsb[((long)jb * 32 + lane) * pd + d] = dacc[d];
#pragma unroll
        for (int d = 0; d < PD; ++d) {
          const float up = __shfl(dacc[d], (lane + 32) & 63, 64);
          dacc[d] = lane < 32 ? up : 0.f;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  if (!live) return;
  if (want_pos && wsb && lane < 32)
    for (int d = 0; d < pd; ++d) wsb[((long)nj * 32 + lane) * pd + d] = dacc[d];
  const int Dp = a.H * (2 * qd + pd);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + acc_row(r, hi);
    if (i < a.T && lo < qd) dqkp[((long)i * a.B + b) * Dp + h * qd + lo] = zq[r];
  }
#pragma unroll
  for (int d = 0; d < PD; ++d) dp[d] += __shfl_xor(dp[d], 32, 64);
  if (hi == 0 && i0 + lo < a.T) {
    float* o = dqkp + ((long)(i0 + lo) * a.B + b) * Dp + 2 * a.H * qd + h * pd;
    for (int d = 0; d < pd; ++d) o[d] = want_pos ? dp[d] : 0.f;
  }
}

// dpos[rel][h][d] += sum over (b, query block) of the partial rows that map to rel
__global__ __launch_bounds__(256) void attn_dpos_reduce_kernel(const float* __restrict__ ws, int T,
                                                               int B, int H, int pd,
                                                               float* __restrict__ dpos) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int rel = idx / pd, d = idx % pd, h = blockIdx.y;
  if (rel >= 2 * T - 1) return;
  const int nj = (T + 31) / 32, rows = (nj + 1) * 32;
  // query blocks whose partial rows cover rel: u = rel - (T-1) + 32 ib + 31 in [0, rows).  The
  // loads of a thread are independent (no branch in the loop, unrolled): the pass used to be 128
  // dependent round trips per thread (62 us at T = 495 for 36 MB)
  const int c = rel - (T - 1) + 31;
  const int ib_lo = c >= 0 ? 0 : (-c + 31) / 32;
  const int ib_hi = min(nj - 1, (rows - 1 - c) >= 0 ? (rows - 1 - c) / 32 : -1);
  float acc = 0.f;
  for (int b = blockIdx.z; b < B; b += gridDim.z) {
    const float* base = ws + (((long)b * H + h) * nj) * rows * pd + (long)c * pd + d;
#pragma unroll 4
    for (int ib = ib_lo; ib <= ib_hi; ++ib) acc += base[((long)ib * rows + 32 * ib) * pd];
  }
  atomicAdd(&dpos[(long)rel * H * pd + h * pd + d], acc);
}

// delta[h,b,i] = sum_j W dW for a materialised dW (one wave per row)
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ W,
                                                         const float* __restrict__ dW, long rows,
                                                         int T, float* __restrict__ delta) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float acc = 0.f;
  for (int j = lane; j < T; j += 64) acc = fmaf(W[row * T + j], dW[row * T + j], acc);
  acc = wave_sum(acc);
  if (lane == 0) delta[row] = acc;
}

template <int NS>
int launch_attn_bwd_mfma(const AttnArgs& a, const float* W, const float* delta, float* dqkp,
                         float* dpos, float* ws, hipStream_t st) {
  const dim3 gq(slab_grid((a.T + 127) / 128, a.B * a.H)), gk(slab_grid((a.T + 31) / 32, a.B * a.H));
  if (a.pd <= 4)
    hipLaunchKernelGGL((attn_bwd_q_mfma_kernel<NS, 4>), gq, dim3(256), 0, st, a, W, delta, dqkp,
                       a.pos ? ws : nullptr);
  else
    hipLaunchKernelGGL((attn_bwd_q_mfma_kernel<NS, 8>), gq, dim3(256), 0, st, a, W, delta, dqkp,
                       a.pos ? ws : nullptr);
  S2T_CHECK_LAUNCH();
  static const bool k_old = getenv("S2T_ATTN_BWD_K_OLD") != nullptr;   // one 32-key tile per workgroup
  const int nz = (a.pos && a.pd > 0) ? (2 * a.T - 1) * a.H * a.pd : 0;
  if (k_old) {
    if (nz && hipMemsetAsync(dpos, 0, sizeof(float) * (size_t)nz, st) != hipSuccess) return -3;
    hipLaunchKernelGGL((attn_bwd_k_mfma_kernel<NS>), gk, dim3(256), 0, st, a, W, delta, dqkp);
  } else {
    hipLaunchKernelGGL((attn_bwd_k3_mfma_kernel<NS>), gq, dim3(256), 0, st, a, W, delta, dqkp, dpos, nz);
  }
  S2T_CHECK_LAUNCH();
  if (a.pos && a.pd > 0) {
    const int n = (2 * a.T - 1) * a.pd;
    hipLaunchKernelGGL(attn_dpos_reduce_kernel, dim3((n + 255) / 256, a.H, a.B < 32 ? a.B : 32),
                       dim3(256), 0, st, ws, a.T, a.B, a.H, a.pd, dpos);
    S2T_CHECK_LAUNCH();
  }
  return 0;
}

// ---------------------------------------------------------------- attention apply
// out[i,b,h*dv+d] = sum_j W[h,b,i,j] v[j,b,h*dv+d]  (TRANS=false, reference zipformer.py:2269)
// dv[j,b,h*dv+d]  = sum_i W[h,b,i,j] g[i,b,h*dv+d]  (TRANS=true, its gradient w.r.t. v)
// workgroup = (128 output rows, b, h); the W tile is staged in LDS with coalesced row reads;
// thread = (row pair, d-group): 8 FMAs per 6 LDS reads; dv <= 16.
template <bool TRANS>
__global__ __launch_bounds__(256) void attn_apply_kernel(const float* __restrict__ W,
                                                         const float* __restrict__ v, int T, int B,
                                                         int H, int dv, float* __restrict__ out) {
  constexpr int RT = 128, CT = 64;
  __shared__ float s_W[TRANS ? CT : RT][(TRANS ? RT : CT) + 1];
  __shared__ float s_v[CT][17];
  const int o0 = blockIdx.x * RT, b = blockIdx.y, h = blockIdx.z;
  const int tid = threadIdx.x, row = tid >> 2, dg = tid & 3;
  const float* Wb = W + ((long)h * B + b) * T * T;
  const long ld = (long)H * dv;
  float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < T; c0 += CT) {
    __syncthreads();
    // W tile: 8 loads per thread in flight (clamped addresses, validity applied on the LDS store);
    // one conditional load per trip serialises into a memory round trip each
    constexpr int WROW = TRANS ? RT : CT;              // tile row length in LDS
    for (int base = 0; base < RT * CT; base += 256 * 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int r = idx / WROW, cc = idx % WROW;
        const int i = TRANS ? c0 + r : o0 + r, j = TRANS ? o0 + cc : c0 + cc;
        t[u] = Wb[(long)min(i, T - 1) * T + min(j, T - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int r = idx / WROW, cc = idx % WROW;
        const int i = TRANS ? c0 + r : o0 + r, j = TRANS ? o0 + cc : c0 + cc;
        s_W[r][cc] = (i < T && j < T) ? t[u] : 0.f;
      }
    }
    for (int idx = tid; idx < CT * dv; idx += 256) {
      const int r = idx / dv, d = idx % dv;
      const int t = c0 + r;
      s_v[r][d] = (t < T) ? v[((long)t * B + b) * ld + h * dv + d] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int cc = 0; cc < CT; ++cc) {
      const float w0 = TRANS ? s_W[cc][row] : s_W[row][cc];
      const float w1 = TRANS ? s_W[cc][row + 64] : s_W[row + 64][cc];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float vv = s_v[cc][dg + 4 * k];
        acc0[k] = fmaf(w0, vv, acc0[k]);
        acc1[k] = fmaf(w1, vv, acc1[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int d = dg + 4 * k;
    if (d < dv) {
      if (o0 + row < T) out[((long)(o0 + row) * B + b) * ld + h * dv + d] = acc0[k];
      if (o0 + row + 64 < T) out[((long)(o0 + row + 64) * B + b) * ld + h * dv + d] = acc1[k];
    }
  }
}

// The same product for dv % 4 == 0 (every shipped configuration), restructured
// around what bounded the kernel above (rocprof: 1.1 TB/s algorithmic): the W tile arrives as ONE
// batch of float4 row loads per thread (one HBM round trip per tile instead of four dependent
// ones), the next tile's loads are in flight in registers while the current one is multiplied, and
// the inner product reads LDS in 16-byte pieces (W along the contraction for TRANS = false, the
// value rows always): 3 LDS instructions per 16 FMAs instead of 6 per 8.
template <bool TRANS>
__global__ __launch_bounds__(256) void attn_apply4_kernel(const float* __restrict__ W,
                                                          const float* __restrict__ v, int T, int B,
                                                          int H, int dv, float* __restrict__ out) {
  constexpr int RT = 128, CT = 64;
  constexpr int WROW = TRANS ? RT : CT, WROWS = TRANS ? CT : RT, WP = WROW + 4;   // pitch (floats)
  constexpr int NV = RT * CT / 4 / 256;                                          // float4 / thread
  __shared__ __attribute__((aligned(16))) float s_W[WROWS * WP];
  __shared__ __attribute__((aligned(16))) float s_v[CT * 16];
  const int o0 = blockIdx.x * RT, b = blockIdx.y, h = blockIdx.z;
  const int tid = threadIdx.x, row = tid >> 2, dg = tid & 3;
  const float* Wb = W + ((long)h * B + b) * T * T;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Wb), 0, T * T * 4, 0x00020000);
  const long ld = (long)H * dv;
  const int ndg = dv >> 2;
  float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
  float4 tw[NV], tv;
  auto load = [&](int c0) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = u * 256 + tid;
      const int r = idx / (WROW / 4), c4 = idx % (WROW / 4);
      const int i = TRANS ? c0 + r : o0 + r, j = (TRANS ? o0 : c0) + 4 * c4;
      // buffer load: 16 bytes from a dword-aligned offset (rows start on 4-byte boundaries when
      // T % 4 != 0 -- T = 495 at C3), dwords beyond the (b, h) slab come back as 0
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i * T + j) * 4, 0, 0);
      tw[u] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z),
                          __uint_as_float(q.w));
    }
    const int r = tid >> 2, t = c0 + r;               // 64 value rows x 4 groups of 4 channels
    tv = (dg < ndg) ? *reinterpret_cast<const float4*>(v + ((long)min(t, T - 1) * B + b) * ld +
                                                        h * dv + 4 * dg)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store = [&](int c0) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int idx = u * 256 + tid;
      const int r = idx / (WROW / 4), c4 = idx % (WROW / 4);
      const int i = TRANS ? c0 + r : o0 + r, j = (TRANS ? o0 : c0) + 4 * c4;
      float4 q = tw[u];
      if (i >= T || j >= T) q.x = 0.f;
      if (i >= T || j + 1 >= T) q.y = 0.f;
      if (i >= T || j + 2 >= T) q.z = 0.f;
      if (i >= T || j + 3 >= T) q.w = 0.f;
      *reinterpret_cast<float4*>(&s_W[r * WP + 4 * c4]) = q;
    }
    const int r = tid >> 2;
    *reinterpret_cast<float4*>(&s_v[r * 16 + 4 * dg]) =
        (c0 + r < T) ? tv : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  load(0);
  for (int c0 = 0; c0 < T; c0 += CT) {
    __syncthreads();                                   // previous tile fully consumed
    store(c0);
    __syncthreads();
    if (c0 + CT < T) load(c0 + CT);                    // flies under the multiply below
    if (!TRANS) {
#pragma unroll 4
      for (int cc = 0; cc < CT; cc += 4) {
        const float4 w0 = *reinterpret_cast<const float4*>(&s_W[row * WP + cc]);
        const float4 w1 = *reinterpret_cast<const float4*>(&s_W[(row + 64) * WP + cc]);
        const float w0a[4] = {w0.x, w0.y, w0.z, w0.w}, w1a[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 vv = *reinterpret_cast<const float4*>(&s_v[(cc + q) * 16 + 4 * dg]);
          acc0.x = fmaf(w0a[q], vv.x, acc0.x); acc0.y = fmaf(w0a[q], vv.y, acc0.y);
          acc0.z = fmaf(w0a[q], vv.z, acc0.z); acc0.w = fmaf(w0a[q], vv.w, acc0.w);
          acc1.x = fmaf(w1a[q], vv.x, acc1.x); acc1.y = fmaf(w1a[q], vv.y, acc1.y);
          acc1.z = fmaf(w1a[q], vv.z, acc1.z); acc1.w = fmaf(w1a[q], vv.w, acc1.w);
        }
      }
    } else {
#pragma unroll 8
      for (int cc = 0; cc < CT; ++cc) {
        const float w0 = s_W[cc * WP + row], w1 = s_W[cc * WP + row + 64];
        const float4 vv = *reinterpret_cast<const float4*>(&s_v[cc * 16 + 4 * dg]);
        acc0.x = fmaf(w0, vv.x, acc0.x); acc0.y = fmaf(w0, vv.y, acc0.y);
        acc0.z = fmaf(w0, vv.z, acc0.z); acc0.w = fmaf(w0, vv.w, acc0.w);
        acc1.x = fmaf(w1, vv.x, acc1.x); acc1.y = fmaf(w1, vv.y, acc1.y);
        acc1.z = fmaf(w1, vv.z, acc1.z); acc1.w = fmaf(w1, vv.w, acc1.w);
      }
    }
  }
  if (dg < ndg) {
    if (o0 + row < T)
      *reinterpret_cast<float4*>(out + ((long)(o0 + row) * B + b) * ld + h * dv + 4 * dg) = acc0;
    if (o0 + row + 64 < T)
      *reinterpret_cast<float4*>(out + ((long)(o0 + row + 64) * B + b) * ld + h * dv + 4 * dg) = acc1;
  }
}

}  // namespace

extern "C" int s2t_attn_apply(const float* W, const float* v, int T, int B, int H, int dv,
                              int transpose, float* out, void* stream) {
  if (T <= 0 || B <= 0 || H <= 0) return 0;
  if (dv <= 0 || dv > 16) return -1;
  dim3 grid((T + 127) / 128, B, H);
  static int old = -1;
  if (old < 0) { const char* e = getenv("S2T_ATTN_APPLY_OLD"); old = e ? atoi(e) : 0; }
  const bool wide = !old && T >= 4 && T <= 16384 && (dv & 3) == 0 &&
                    ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(W) & 3) == 0;
  if (wide) {
    if (transpose)
      hipLaunchKernelGGL(attn_apply4_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, W, v, T,
                         B, H, dv, out);
    else
      hipLaunchKernelGGL(attn_apply4_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, W, v, T,
                         B, H, dv, out);
    S2T_CHECK_LAUNCH();
    return 0;
  }
  if (transpose)
    hipLaunchKernelGGL(attn_apply_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, W, v, T,
                       B, H, dv, out);
  else
    hipLaunchKernelGGL(attn_apply_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, W, v, T,
                       B, H, dv, out);
  S2T_CHECK_LAUNCH();
  return 0;
}

template <int NT, int SPLIT, bool HAS_POS, bool HAS_AM>
int launch_fwd_mfma_v(const AttnArgs& a, float* W, hipStream_t st) {
  auto kern = attn_fwd_mfma_kernel<NT, SPLIT, HAS_POS, HAS_AM>;
  const size_t smem = attn_mfma_smem<NT * SPLIT>() +
                      ((HAS_AM && NT * SPLIT >= 8) ? sizeof(unsigned) * MF_ROWS * NT * SPLIT : 0);
  static bool attr = false;                       // per instantiation
  if (!attr && smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
  dim3 grid((a.T + MF_ROWS - 1) / MF_ROWS, a.B, a.H);
  hipLaunchKernelGGL(kern, grid, dim3(256 * SPLIT), smem, st, a, W);
  return (int)hipGetLastError();
}
template <int NT, int SPLIT>
int launch_fwd_mfma(int variant, const AttnArgs& a, float* W, hipStream_t st) {
  switch (variant) {
    case 0: return launch_fwd_mfma_v<NT, SPLIT, false, false>(a, W, st);
    case 1: return launch_fwd_mfma_v<NT, SPLIT, true, false>(a, W, st);
    case 2: return launch_fwd_mfma_v<NT, SPLIT, false, true>(a, W, st);
    default: return launch_fwd_mfma_v<NT, SPLIT, true, true>(a, W, st);
  }
}

extern "C" int s2t_relpos_attn_fwd_flag(const float* qkp, const float* pos,
                                        const unsigned char* kpm, const unsigned char* amask,
                                        int T, int B, int H, int qd, int pd, float* W,
                                        float pen_limit, float* pen_flag, void* stream);

extern "C" int s2t_relpos_attn_fwd(const float* qkp, const float* pos, const unsigned char* kpm,
                                   const unsigned char* amask, int T, int B, int H, int qd, int pd,
                                   float* W, void* stream) {
  return s2t_relpos_attn_fwd_flag(qkp, pos, kpm, amask, T, B, H, qd, pd, W, 0.f, nullptr, stream);
}

extern "C" int s2t_relpos_attn_fwd_flag(const float* qkp, const float* pos,
                                        const unsigned char* kpm, const unsigned char* amask,
                                        int T, int B, int H, int qd, int pd, float* W,
                                        float pen_limit, float* pen_flag, void* stream) {
  if (T <= 0 || B <= 0 || H <= 0) return 0;
  if (qd <= 0 || qd > MAXQD || pd < 0 || pd > MAXPD) return -1;
  AttnArgs a{qkp, pos, kpm, amask, T, B, H, qd, pd, nullptr, nullptr, {nullptr, nullptr},
             {nullptr, nullptr}, {0, 0}, pen_limit, pen_flag};
  hipStream_t st = (hipStream_t)stream;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<8>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e1 != hipSuccess) return (int)e1;
    attr_done = true;
  }
  // MFMA path: whole key matrix of a (b,h) in LDS, scores of a 32-row strip in accumulators
  static const bool no_mfma = getenv("S2T_ATTN_FWD_OLD") != nullptr;
  const bool aligned = ((reinterpret_cast<uintptr_t>(qkp) & 15) == 0) && (qd % 8 == 0) &&
                       ((H * (2 * qd + pd)) % 4 == 0) && ((H * qd) % 4 == 0);
  if (!no_mfma && T <= 512 && pd <= 4 && aligned) {
    const int v = (pos ? 1 : 0) + (amask ? 2 : 0);
    int rc;
    if (T > 256) rc = launch_fwd_mfma<8, 2>(v, a, W, st);
    else if (T > 128) rc = launch_fwd_mfma<8, 1>(v, a, W, st);
    else if (T > 64) rc = launch_fwd_mfma<4, 1>(v, a, W, st);
    else rc = launch_fwd_mfma<2, 1>(v, a, W, st);
    return rc;
  }
  if (pen_flag) return -3;                        // only the MFMA kernel reports the score limit
  dim3 grid((T + ROWS - 1) / ROWS, B, H);
  if (T <= 128)
    hipLaunchKernelGGL(attn_fwd_kernel<2>, grid, dim3(256), attn_smem<2>(qd, pd), st, a, W);
  else if (T <= 256)
    hipLaunchKernelGGL(attn_fwd_kernel<4>, grid, dim3(256), attn_smem<4>(qd, pd), st, a, W);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<8>, grid, dim3(256), attn_smem<8>(qd, pd), st, a, W);
  S2T_CHECK_LAUNCH();
  return 0;
}

// Gradient of W is given materialised (dW) and/or as factors (see AttnArgs): dW0 (B,T,T) for
// head 0, and up to two (dO_c, V_c) pairs of (T,B,H*dv_c) tensors.  When any factor is used,
// `delta_ws` must hold delta[h,b,i] = sum_j W dW on entry (delta_given = 1; for the pairs it is
// sum_d dO_c * O_c, O_c = the apply's forward output).  dqkp (T,B,Dp) is fully written; dpos
// (2T-1, H*pd) must be zeroed by the caller (accumulated) or NULL when pos was skipped.
extern "C" long s2t_relpos_attn_bwd_workspace_floats(int T, int B, int H, int pd) {
  const long nj = (T + 31) / 32;
  return (long)B * H * nj * (nj + 1) * 32 * pd;
}

extern "C" int s2t_relpos_attn_bwd(const float* qkp, const float* pos, const unsigned char* kpm,
                                   const unsigned char* amask, int T, int B, int H, int qd, int pd,
                                   const float* W, const float* dW, const float* dW0,
                                   const float* dO1, const float* V1, int dv1, const float* dO2,
                                   const float* V2, int dv2, int delta_given, float* delta_ws,
                                   float* dqkp, float* dpos, float* workspace, void* stream) {
  if (T <= 0 || B <= 0 || H <= 0) return 0;
  if (qd <= 0 || qd > MAXQD || pd < 0 || pd > MAXPD) return -1;
  if ((dO1 ? dv1 : 0) + (dO2 ? dv2 : 0) > MAXCD) return -1;
  if ((long)T * T >= (1L << 31) || (long)T * B * H * (2 * qd + pd) >= (1L << 31)) return -1;
  if (!dW && !delta_given) return -1;
  AttnArgs a{qkp, pos, kpm, amask, T, B, H, qd, pd, dW, dW0, {dO1, dO2}, {V1, V2}, {dv1, dv2}};
  static const bool use_valu = getenv("S2T_ATTN_BWD_VALU") != nullptr;
  if (!use_valu) {
    hipStream_t st = (hipStream_t)stream;
    if (pos && pd > 0 && !workspace) return -1;
    if (!delta_given) {
      const long rows = (long)H * B * T;
      hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, W,
                         dW, rows, T, delta_ws);
      S2T_CHECK_LAUNCH();
    }
    const int cd = (dO1 ? dv1 : 0) + (dO2 ? dv2 : 0);
    if (cd == 0) return launch_attn_bwd_mfma<0>(a, W, delta_ws, dqkp, dpos, workspace, st);
    if (cd <= 12) return launch_attn_bwd_mfma<6>(a, W, delta_ws, dqkp, dpos, workspace, st);
    if (cd <= 24) return launch_attn_bwd_mfma<12>(a, W, delta_ws, dqkp, dpos, workspace, st);
    return launch_attn_bwd_mfma<16>(a, W, delta_ws, dqkp, dpos, workspace, st);
  }
  dim3 grid((T + ROWS - 1) / ROWS, B, H);
  if (pos && pd > 0 &&
      hipMemsetAsync(dpos, 0, sizeof(float) * (size_t)(2 * T - 1) * H * pd, (hipStream_t)stream) != hipSuccess)
    return -3;
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, W, delta_given,
                     delta_ws, dqkp);
  S2T_CHECK_LAUNCH();
  const size_t sm = pos ? sizeof(float) * (size_t)(T + ROWS) * pd : 0;
  if (sm > 64 * 1024) return -1;
  hipLaunchKernelGGL(attn_bwd_k_kernel, grid, dim3(256), sm, (hipStream_t)stream, a, W, delta_ws,
                     dqkp, pos ? dpos : nullptr);
  S2T_CHECK_LAUNCH();
  return 0;
}
