// fp32 GEMM family on v_mfma_f32_32x32x2_f32 with fused prologues / epilogues, for the dense
// layers of the training path (reference: every nn.Linear / ScaledLinear /
// ActivationDropoutAndLinear of model/encoder/zipformer.py:1924-2695 and
// model/layer/scaling.py:1512-1583, plus their gradients under loss.backward()).
//
//   mode NT:  C[M,N]  = pro_a(A[M,K]) . B[N,K]^T          forward  y = act(x) W^T (+bias, +residual)
//   mode NN:  C[M,N]  = A[M,K] . B[K,N]                   dgrad    dx = g W  (* act'(h), + residual)
//   mode TN:  C[M,N] += A[K,M]^T . pro_b(B[K,N])          wgrad    dW += g^T act(x), db += colsum(g)
//
// Workgroup = 4 waves in a 2x2 grid, each wave TM x TN MFMA tiles of 32x32 (block tile
// 64TM x 64TN), K chunks of 32 staged global -> registers -> LDS (the next chunk's global loads
// are in flight while the current one is multiplied).  One lane reads 4 consecutive k of its
// row with a single ds_read_b128 (k-contiguous operands) or 4 ds_read_b32 (k-major operands);
// the two k slots of each MFMA take k and k+4, which is a valid summation order because both
// operands use the same assignment.  f32 MFMA is exact fp32 (fmaf chain), so results match a
// torch fp32 GEMM to rounding-order differences only.
// TN splits the (long) contraction dimension over gridDim.z and accumulates with fp32 atomics
// straight into the caller's gradient buffer: no partial-sum pass, no separate "+=" kernel.
#include "common.h"
#include "../../include/s2t_mi355.h"
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// (x0, x1) -> three packed bf16 pairs, exact: x = p0 + p1 + p2 to 24 bits
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p0, unsigned& p1,
                                           unsigned& p2) {
  f32x2 x = {x0, x1};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  f32x2 h = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
  x = x - h;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  f32x2 h1 = {__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xFFFF0000u)};
  x = x - h1;
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}

// the two leading pieces only (S2T_GEMM_ARITH=2, csrc/gemm_x3p.hip: x = p0 + p1 + O(2^-18 |x|))
__device__ __forceinline__ void split_pair2(float x0, float x1, unsigned& p0, unsigned& p1) {
  f32x2 x = {x0, x1};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  f32x2 h = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
  x = x - h;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}

constexpr int BK0 = 32;   // contraction chunk of the big tiles; the 64x64 tile uses 64
enum { MODE_NT = 0, MODE_NN = 1, MODE_TN = 2 };
enum { ACT_NONE = 0, ACT_SWOOSH_L = 1, ACT_SWOOSH_R = 2 };

// Implicit im2col of a 3x3 convolution on a channel-last map x (B,H,W,C): row r = (b, ho, wo) of
// the patch matrix starts at x + b sb + ho sh + wo sw (floats) and its 9C columns are three runs
// of seg = 3C contiguous floats, s1 = W C apart -- the operand is read straight from x.
struct Patch {
  int seg, hw, wo, sb, sh, sw, s1;
};
__device__ __forceinline__ int patch_row(const Patch& p, int r) {
  const int b = r / p.hw, q = r - b * p.hw, ho = q / p.wo, wo = q - ho * p.wo;
  return b * p.sb + ho * p.sh + wo * p.sw;
}
__device__ __forceinline__ int patch_col(const Patch& p, int j) {
  const int kh = j / p.seg;
  return kh * p.s1 + (j - kh * p.seg);
}

struct GemmArgs {
  const float* A;
  long lda;
  const float* B;
  long ldb;
  float* C;
  long ldc;
  int M, N, K;
  const float* bias;      // [N] added to every row (NT / NN), or NULL
  const float* resid;     // [M][N] added, or NULL
  long ldr;
  const float* act_src;   // epilogue: C *= act'(act_src[m][n]) (dgrad through the activation)
  long lds;
  int act_kind;
  int pro_a, pro_b;       // activation applied to A / B elements while staging
  float* colsum;          // TN: colsum[m] += sum_k A[k][m]   (bias gradient), or NULL
  int accumulate;         // NT / NN: C += result
  int kper;               // TN: contraction rows per z-slice (multiple of BK)
  int tiles_m, tiles_n;
  int splits;
  int debug;
  float alpha;            // TN: scale of the accumulated product / column sums
  int sym_cg;             // TN with A == B (x^T x): > 0 = only the 64x64 tiles on / above the
                          // diagonal that hold same-group pairs (groups of sym_cg channels)
  int tile_force;         // NT / NN: 0 = dispatch's choice, else "tm tn" digits (11 12 21 22 23)
  int wide_ep;            // NT / NN: rows of C / residual / bias 16-byte aligned -> float4 epilogue
  Patch pt;               // PATCH kernels: A (NT) or B (TN) is the implicit patch matrix of pt
  // NT / NN, 16-byte epilogue only (s2t_gemm_f32_sq): sq_sums[0] += sum of squares of the (M, N)
  // matrix sq_other (rows ld_sq apart), sq_sums[1] += sum of squares of C as stored -- the two norms
  // Whiten's backward needs (scaling.py:1024-1027), taken while C leaves the accumulators
  const float* sq_other;
  long ld_sq;
  float* sq_sums;
  // bf16 pieces per fp32 operand of the matrix-core paths: 2 = the three leading products of the
  // two-piece split, anything else = three pieces, six products (s2t_gemm_arith(), read per call
  // by the entry points; uniform over the launch)
  int np;
};

__device__ __forceinline__ float log1p_fast(float e) {   // as zip_elem.hip
  const float u = 1.f + e;
  return u == 1.f ? e : __logf(u) * __fdividef(e, u - 1.f);
}
__device__ __forceinline__ float swoosh(float x, int kind) {
  // log(1 + exp(x - off)) - 0.08 x - c   (scaling.py:1340-1343, 1418-1423)
  const float off = kind == ACT_SWOOSH_L ? 4.f : 1.f;
  const float c = kind == ACT_SWOOSH_L ? 0.035f : 0.313261687f;
  const float z = x - off;
  return fmaxf(z, 0.f) + log1p_fast(__expf(-fabsf(z))) - 0.08f * x - c;   // as zip_elem.hip swoosh_f
}
__device__ __forceinline__ float swoosh_deriv(float x, int kind) {
  const float off = kind == ACT_SWOOSH_L ? 4.f : 1.f;
  return __fdividef(1.f, 1.f + __expf(off - x)) - 0.08f;
}
// One operand tile in LDS.  KC: [ROWS][BK + 4] (rows = output index, k contiguous);
// KM: [BK][ROWS + 4] (k-major).  ROWS = 64 * T.  Loads are unconditional (clamped addresses) so
// that all of a chunk's global loads are in flight together; validity and the activation are
// applied one iteration later, when the registers are written to LDS.
template <int ROWS, bool KC, int ACT, int BK, bool PATCH = false>
struct Tile {
  static constexpr int LD = KC ? (BK + 4) : (ROWS + 4);
  static constexpr int SIZE = KC ? ROWS * LD : BK * LD;
  static constexpr int NV = ROWS * BK / 4 / 256;     // float4 per thread per chunk

  // global -> registers.  KC: src[out0 + r][k0 + 4c];  KM: src[k0 + r][out0 + 4c]
  __device__ static __forceinline__ unsigned load(float4 (&v)[NV], const float* __restrict__ src,
                                                  long ld, int out0, int out_n, int k0, int k_n,
                                                  const Patch& pt) {
    unsigned ok = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + 256 * i;
      int o, k;
      if (KC) {
        o = out0 + idx / (BK / 4);                     // BK / 4 float4 per row of BK k
        k = k0 + 4 * (idx % (BK / 4));
      } else {
        constexpr int V = ROWS / 4;                    // float4 per k row
        k = k0 + idx / V;
        o = out0 + 4 * (idx % V);
      }
      const bool valid = o < out_n && k < k_n;
      ok |= (valid ? 1u : 0u) << i;
      const int oc = min(o, KC ? out_n - 1 : out_n - 4), kc = min(k, KC ? k_n - 4 : k_n - 1);
      const float* p;
      if (PATCH)   // patch matrix row = output index (KC, forward) or contraction index (k-major)
        p = KC ? src + patch_row(pt, oc) + patch_col(pt, kc) : src + patch_row(pt, kc) + patch_col(pt, oc);
      else
        p = KC ? src + (long)oc * ld + kc : src + (long)kc * ld + oc;
      v[i] = *reinterpret_cast<const float4*>(p);
    }
    return ok;
  }
  __device__ static __forceinline__ void store(float* __restrict__ s, const float4 (&v)[NV],
                                               unsigned ok) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + 256 * i;
      float4 x = v[i];
      if (ACT != ACT_NONE)
        x = make_float4(swoosh(x.x, ACT), swoosh(x.y, ACT), swoosh(x.z, ACT), swoosh(x.w, ACT));
      if (!((ok >> i) & 1u)) x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        *reinterpret_cast<float4*>(s + (idx / (BK / 4)) * LD + 4 * (idx % (BK / 4))) = x;
      } else {
        constexpr int V = ROWS / 4;
        *reinterpret_cast<float4*>(s + (idx / V) * LD + 4 * (idx % V)) = x;
      }
    }
  }
  // fragment of MFMA tile `t0` (32 outputs starting at out index t0) for k group s (8 k):
  // element j = value at k = 8 s + 4 (lane >> 5) + j
  __device__ static __forceinline__ float4 frag(const float* __restrict__ s, int t0, int sgrp,
                                                int lane) {
    const int o = t0 + (lane & 31), kb = 8 * sgrp + 4 * (lane >> 5);
    if (KC) return *reinterpret_cast<const float4*>(s + o * LD + kb);
    return make_float4(s[kb * LD + o], s[(kb + 1) * LD + o], s[(kb + 2) * LD + o],
                       s[(kb + 3) * LD + o]);
  }
};

constexpr int bk_of(int tm, int tn, int mode) {
  return BK0;   // (a 64-deep chunk for the 64x64 TN tile measured the same as 32)
}

// ---- TN on the bf16 matrix cores with the operands split ONCE, when they are staged (round 4).
// The form below (X3) keeps fp32 tiles in LDS and splits every fragment in each of the two waves
// that read it -- 8 strided ds_read_b32 + ~50 VALU instructions per fragment against 6 MFMAs per
// 16-deep step of a 32 x 32 wave tile: the weight-gradient kernel spent more VALU cycles
// splitting than the matrix pipe spent multiplying.  Here a staging thread owns ONE output column
// and 8 consecutive contraction rows (8 coalesced 4-byte loads: a wave-instruction = 64
// consecutive floats of one row), splits them once and writes the three 16-byte pieces straight
// into the fragment-major image [32-output block][16-deep step][piece][lane][8 bf16]; the main loop
// is {3 ds_read_b128 per fragment, 6 MFMAs}, no VALU.  Same tiles, chunks, barriers, slices and
// atomics as the X3 form; the bias gradient (column sums of A) is taken from the staged registers.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// NP: pieces the LDS image holds (3: either arithmetic, chosen by g.np; 2: the three-product arithmetic
// only, in two thirds of the LDS -- its own kernels, gemm_tn_p2_kernel / gemm_tn_grouped_p2_kernel)
// PATCH: B is the implicit patch matrix of g.pt (a 3x3 convolution's weight gradient: row k = output
// pixel, 9 C columns in three runs of 3 C contiguous floats of the channel-last map)
template <int TM, int TN, int PRO, int NP = 3, bool PATCH = false>
__device__ __forceinline__ void tn_p3_body(const GemmArgs& g, const unsigned bid) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  __shared__ __attribute__((aligned(16))) unsigned char sA[BM * BK * 2 * NP];
  __shared__ __attribute__((aligned(16))) unsigned char sB[BN * BK * 2 * NP];
  const int total = g.tiles_m * g.tiles_n;
  const int q = (int)(bid >> 3);
  const int zslice = (int)(bid & 7) + 8 * (q / total);
  const int lin = q % total;
  if (zslice >= g.splits) return;
  const int tm = lin / g.tiles_n, tn = lin % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (g.sym_cg > 0) {
    if (tn < tm || n0 / g.sym_cg > (m0 + BM - 1) / g.sym_cg) return;
  }
  const int csum_tn = g.sym_cg > 0 ? tm : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int kbeg = zslice * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  if (kbeg >= kend) return;
  const bool want_csum = g.colsum != nullptr && tn == csum_tn;
  const bool np3 = NP == 3 && g.np != 2;  // (uniform: three pieces / six products, or two / three)

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: output so of each 64-column unit, contraction rows 8 kg .. 8 kg + 7 of the chunk
  const int so = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const unsigned sdst = (unsigned)(((((so >> 5) * 2 + (kg >> 1)) * NP) * 64 + (kg & 1) * 32 + (so & 31)) * 16);
  float ra[TM][8], rb[TN][8];
  float csum[TM];
#pragma unroll
  for (int u = 0; u < TM; ++u) csum[u] = 0.f;
#define TN3_LOAD(R, NU, SRC, LD, O0, ON, K0)                                                   \
  _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                             \
    const int oc_ = min((O0) + 64 * u + so, (ON) - 1);                                         \
    _Pragma("unroll") for (int e = 0; e < 8; ++e)                                              \
      R[u][e] = (SRC)[(long)min((K0) + 8 * kg + e, kend - 1) * (LD) + oc_];                    \
  }
  // registers of chunk K0 -> pieces in LDS (invalid rows / columns as zeros)
#define TN3_STORE(R, NU, S, O0, ON, K0, ACT, CSUM)                                             \
  _Pragma("unroll") for (int u = 0; u < NU; ++u) {                                             \
    const bool ov_ = (O0) + 64 * u + so < (ON);                                                \
    float v_[8];                                                                               \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
      float x_ = R[u][e];                                                                      \
      if (ACT != ACT_NONE) x_ = swoosh(x_, ACT);                                               \
      v_[e] = (ov_ && (K0) + 8 * kg + e < kend) ? x_ : 0.f;                                    \
    }                                                                                          \
    if (CSUM) csum[u] += ((v_[0] + v_[1]) + (v_[2] + v_[3])) + ((v_[4] + v_[5]) + (v_[6] + v_[7])); \
    unsigned a0_, a1_, a2_, b0_, b1_, b2_, c0_, c1_, c2_, d0_, d1_, d2_;                       \
    unsigned char* d_ = (S) + u * (2 * 2 * NP * 1024) + sdst;                                   \
    if (np3) {                                                                                 \
      split_pair(v_[0], v_[1], a0_, a1_, a2_);                                                 \
      split_pair(v_[2], v_[3], b0_, b1_, b2_);                                                 \
      split_pair(v_[4], v_[5], c0_, c1_, c2_);                                                 \
      split_pair(v_[6], v_[7], d0_, d1_, d2_);                                                 \
      const u32x4_t q2_ = {a2_, b2_, c2_, d2_};                                                \
      *reinterpret_cast<u32x4_t*>(d_ + 2048) = q2_;                                            \
    } else {                                                                                   \
      split_pair2(v_[0], v_[1], a0_, a1_);                                                     \
      split_pair2(v_[2], v_[3], b0_, b1_);                                                     \
      split_pair2(v_[4], v_[5], c0_, c1_);                                                     \
      split_pair2(v_[6], v_[7], d0_, d1_);                                                     \
    }                                                                                          \
    const u32x4_t q0_ = {a0_, b0_, c0_, d0_}, q1_ = {a1_, b1_, c1_, d1_};                      \
    *reinterpret_cast<u32x4_t*>(d_) = q0_;                                                     \
    *reinterpret_cast<u32x4_t*>(d_ + 1024) = q1_;                                              \
  }
  // the patch operand: this thread's column offsets inside a patch (fixed), and per chunk ONE
  // (image, row, column) decomposition of its first pixel, then steps of one pixel (pixels past the
  // slice's end repeat its last one; they are zeroed when stored)
  int pcol[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) pcol[u] = PATCH ? patch_col(g.pt, min(n0 + 64 * u + so, g.N - 1)) : 0;
#define TN3_LOAD_PATCH(K0)                                                                     \
  {                                                                                            \
    int k_ = min((K0) + 8 * kg, kend - 1);                                                     \
    const int rows_ = g.pt.hw / g.pt.wo;                                                       \
    const int b_ = k_ / g.pt.hw, q_ = k_ - b_ * g.pt.hw;                                       \
    int ho_ = q_ / g.pt.wo, wo_ = q_ - ho_ * g.pt.wo;                                          \
    int off_ = b_ * g.pt.sb + ho_ * g.pt.sh + wo_ * g.pt.sw;                                   \
    _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                            \
      _Pragma("unroll") for (int u = 0; u < TN; ++u) rb[u][e] = g.B[off_ + pcol[u]];           \
      if (k_ + 1 < kend) {                                                                     \
        ++k_;                                                                                  \
        if (++wo_ == g.pt.wo) {                                                                \
          wo_ = 0;                                                                             \
          off_ += g.pt.sh - (g.pt.wo - 1) * g.pt.sw;                                           \
          if (++ho_ == rows_) { ho_ = 0; off_ += g.pt.sb - rows_ * g.pt.sh; }                  \
        } else {                                                                               \
          off_ += g.pt.sw;                                                                     \
        }                                                                                      \
      }                                                                                        \
    }                                                                                          \
  }
  TN3_LOAD(ra, TM, g.A, g.lda, m0, g.M, kbeg)
  if (PATCH) { TN3_LOAD_PATCH(kbeg) } else { TN3_LOAD(rb, TN, g.B, g.ldb, n0, g.N, kbeg) }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();                       // previous chunk fully consumed
    if (want_csum) { TN3_STORE(ra, TM, sA, m0, g.M, k0, ACT_NONE, true) }
    else { TN3_STORE(ra, TM, sA, m0, g.M, k0, ACT_NONE, false) }
    TN3_STORE(rb, TN, sB, n0, g.N, k0, PRO, false)
    __syncthreads();
    if (k0 + BK < kend) {                  // next chunk's global loads fly under the MFMAs
      TN3_LOAD(ra, TM, g.A, g.lda, m0, g.M, k0 + BK)
      if (PATCH) { TN3_LOAD_PATCH(k0 + BK) } else { TN3_LOAD(rb, TN, g.B, g.ldb, n0, g.N, k0 + BK) }
    }
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 pa[TM][3], pb[TN][3];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (p < 2 || np3)
            pa[i][p] = *reinterpret_cast<const bf16x8*>(sA + ((((wm >> 5) + i) * 2 + s) * NP + p) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          if (p < 2 || np3)
            pb[j][p] = *reinterpret_cast<const bf16x8*>(sB + ((((wn >> 5) + j) * 2 + s) * NP + p) * 1024 + lane * 16);
#define S2T_P3_TERM(PA, PB)                                                                     \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][PA], pb[j][PB], acc[i][j], 0, 0, 0);
      if (np3) { S2T_P3_TERM(2, 0) S2T_P3_TERM(1, 1) S2T_P3_TERM(0, 2) }
      S2T_P3_TERM(1, 0) S2T_P3_TERM(0, 1)
      S2T_P3_TERM(0, 0)
#undef S2T_P3_TERM
    }
  }
#undef TN3_LOAD
#undef TN3_LOAD_PATCH
#undef TN3_STORE
  if (want_csum) {                         // the four row groups' partial sums -> one add per column
    __syncthreads();
    float* red = reinterpret_cast<float*>(sA);
#pragma unroll
    for (int u = 0; u < TM; ++u) red[kg * BM + 64 * u + so] = csum[u];
    __syncthreads();
    if (threadIdx.x < BM && m0 + (int)threadIdx.x < g.M) {
      const int t = threadIdx.x;
      atomicAdd(g.colsum + m0 + t, ((red[t] + red[BM + t]) + (red[2 * BM + t] + red[3 * BM + t])) * g.alpha);
    }
  }
  if (g.debug & 1) return;
  const int hi = lane >> 5, lo = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + lo;
      if (col >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row >= g.M) continue;
        atomicAdd(g.C + (long)row * g.ldc + col, acc[i][j][r] * g.alpha);
      }
    }
}

// ---- TN "W" form (round 5): wave-specialised, (64 MI) x (64 NJ) output tile per workgroup of 8 waves.
// The 64 x 64 form above spends as many VALU cycles splitting fragments (4.5 instructions per
// element) as the matrix cores spend on the products, in the SAME waves, so the two serialize.
// Here waves 4-7 (PRODUCERS) load 16-byte runs of four adjacent columns x 8 contraction rows,
// transpose in registers, split and write the piece image of chunk c+1 into one LDS buffer, while
// waves 0-3 (CONSUMERS, one per SIMD, 32 MI x 32 NJ each) run the 6 x MI x NJ x 2 MFMAs of chunk c
// out of the other -- the splitting rides under the matrix work of the same SIMD, one barrier per
// chunk.  One workgroup per CU (96-120 KB of LDS), so a launch wants ~256-512 workgroups: fewer,
// longer contraction slices, i.e. fewer atomic bytes than the 64 x 64 form's ~6000.
// The lane slots of a fragment are permuted (tnw_perm32) so that the four-column 16-byte writes
// (8-lane groups, 32 banks) and the fragment reads (16-lane groups, 64 banks) are conflict-free.
__device__ __forceinline__ int tnw_perm32(int t) {        // t = 4 a + mi  ->  ((a + 2 mi) & 7) + 8 mi
  const int a = t >> 2, mi = t & 3;
  return ((a + 2 * mi) & 7) + 8 * mi;
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
struct TnwUnit {           // a producer's unit: four columns x eight contraction rows of one operand
  const float* src;        // column-quad base (row 0 of the operand)
  long ld;
  unsigned dst;            // byte offset of (mi = 0) slot base in a buffer (without the permutation)
  int a, kg;
  bool valid, colok, isA;
};
template <int MI, int NJ, int PRO, bool PATCH = false>
__device__ __forceinline__ void tn_w_body(const GemmArgs& g, const unsigned bid, unsigned char* sm) {
  constexpr int BM = 64 * MI, BN = 64 * NJ, BK = 32;
  constexpr int IMG_A = BM * BK * 6, IMG_B = BN * BK * 6, BUF = IMG_A + IMG_B;
  constexpr int UNITS = BM + BN, NSLOT = (UNITS + 255) / 256;
  const int total = g.tiles_m * g.tiles_n;
  const int q = (int)(bid >> 3);
  const int zslice = (int)(bid & 7) + 8 * (q / total);
  const int lin = q % total;
  if (zslice >= g.splits) return;
  const int tm = lin / g.tiles_n, tn = lin % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int kbeg = zslice * g.kper;
  const int kend = min(g.K, kbeg + g.kper);
  if (kbeg >= kend) return;
  const int nchunk = (kend - kbeg + BK - 1) / BK;
  const bool want_csum = g.colsum != nullptr && tn == 0;
  const bool np3 = g.np != 2;            // (uniform: three pieces / six products, or two / three)

  // ---- producer state
  TnwUnit un[NSLOT];
  f32x4_t r[NSLOT][8];
  float cs[NSLOT][4];
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int u = (tid & 255) + 256 * j;
    TnwUnit& t = un[j];
    t.valid = producer && u < UNITS;
    t.isA = u < BM;
    const int idx = t.isA ? u : u - BM;
    const int Q = t.isA ? BM / 4 : BN / 4;
    const int kg = idx / Q, mq = idx - kg * Q;
    const int col = (t.isA ? m0 : n0) + 4 * mq;
    t.colok = col < (t.isA ? g.M : g.N);                  // (column counts are multiples of 4)
    t.ld = t.isA ? g.lda : g.ldb;
    // PATCH: B is the implicit patch matrix of g.pt (row k = output pixel, 9C columns in three runs
    // of 3C contiguous floats: a column quad never straddles a run, seg % 4 == 0)
    t.src = (t.isA ? g.A : g.B) + (t.colok ? ((PATCH && !t.isA) ? patch_col(g.pt, col) : col) : 0);
    t.kg = kg;
    t.a = mq & 7;
    t.dst = (unsigned)((t.isA ? 0 : IMG_A) + (((mq >> 3) * 2 + (kg >> 1)) * 3) * 1024 + (kg & 1) * 512);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) cs[j][mi] = 0.f;
  }
  auto load = [&](int c) {
    const int k0 = kbeg + c * BK;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if (un[j].valid) {
        if (PATCH && !un[j].isA) {
          // the unit's eight consecutive output pixels: one (image, row, column) decomposition, then
          // steps of one pixel; pixels past the slice's end repeat its last one (zeroed in store)
          int k = min(k0 + 8 * un[j].kg, kend - 1);
          const int rows = g.pt.hw / g.pt.wo;
          int b = k / g.pt.hw, q = k - b * g.pt.hw, ho = q / g.pt.wo, wo = q - ho * g.pt.wo;
          int off = b * g.pt.sb + ho * g.pt.sh + wo * g.pt.sw;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            r[j][e] = *reinterpret_cast<const f32x4_t*>(un[j].src + off);
            if (k + 1 < kend) {
              ++k;
              if (++wo == g.pt.wo) {
                wo = 0;
                off += g.pt.sh - (g.pt.wo - 1) * g.pt.sw;
                if (++ho == rows) { ho = 0; off += g.pt.sb - rows * g.pt.sh; }
              } else {
                off += g.pt.sw;
              }
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            r[j][e] = *reinterpret_cast<const f32x4_t*>(un[j].src + (long)min(k0 + 8 * un[j].kg + e, kend - 1) * un[j].ld);
        }
      }
  };
  // (columns past M / N need no zeroing: they only reach output rows / columns that are never
  // written; contraction rows past kend do, in the slice's last chunk only)
  auto store_t = [&](int c, auto tailc) {
    constexpr bool TAIL = decltype(tailc)::value;
    const int k0 = kbeg + c * BK;
    unsigned char* const buf = sm + (c & 1) * BUF;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if (un[j].valid) {
        const TnwUnit& t = un[j];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = r[j][e][mi];
            if (PRO != ACT_NONE && !t.isA) x = swoosh(x, PRO);
            v[e] = (!TAIL || k0 + 8 * t.kg + e < kend) ? x : 0.f;
          }
          if (want_csum && t.isA)
            cs[j][mi] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
          unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
          unsigned char* const d = buf + t.dst + ((((t.a + 2 * mi) & 7) + 8 * mi) << 4);
          if (np3) {
            split_pair(v[0], v[1], a0, a1, a2);
            split_pair(v[2], v[3], b0, b1, b2);
            split_pair(v[4], v[5], c0, c1, c2);
            split_pair(v[6], v[7], d0, d1, d2);
            const u32x4_t q2 = {a2, b2, c2, d2};
            *reinterpret_cast<u32x4_t*>(d + 2048) = q2;
          } else {
            split_pair2(v[0], v[1], a0, a1);
            split_pair2(v[2], v[3], b0, b1);
            split_pair2(v[4], v[5], c0, c1);
            split_pair2(v[6], v[7], d0, d1);
          }
          const u32x4_t q0 = {a0, b0, c0, d0}, q1 = {a1, b1, c1, d1};
          *reinterpret_cast<u32x4_t*>(d) = q0;
          *reinterpret_cast<u32x4_t*>(d + 1024) = q1;
        }
      }
  };

  auto store = [&](int c) {
    if (kbeg + (c + 1) * BK <= kend) store_t(c, std::false_type{});
    else store_t(c, std::true_type{});
  };

  // ---- consumer state
  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wm = (wave >> 1) * 32 * MI, wn = (wave & 1) * 32 * NJ;    // (consumers: wave < 4)
  const unsigned fslot = (unsigned)(((lane >> 5) * 32 + tnw_perm32(lane & 31)) * 16);

  if (producer) {
    load(0);
    store(0);
    if (nchunk > 1) load(1);
  }
  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    if (producer) {
      if (c + 1 < nchunk) {
        if (!(g.debug & 2)) store(c + 1);
        if (c + 2 < nchunk && !(g.debug & 8)) load(c + 2);
      }
    } else if (!(g.debug & 4)) {
      const unsigned char* const sA = sm + (c & 1) * BUF;
      const unsigned char* const sB = sA + IMG_A;
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        bf16x8 pa[MI][3], pb[NJ][3];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            if (p < 2 || np3)
              pa[i][p] = *reinterpret_cast<const bf16x8*>(sA + ((((wm >> 5) + i) * 2 + s) * 3 + p) * 1024 + fslot);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            if (p < 2 || np3)
              pb[j][p] = *reinterpret_cast<const bf16x8*>(sB + ((((wn >> 5) + j) * 2 + s) * 3 + p) * 1024 + fslot);
#define S2T_W_TERM(PA, PB)                                                                       \
  _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NJ; ++j)  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][PA], pb[j][PB], acc[i][j], 0, 0, 0);
        if (np3) { S2T_W_TERM(2, 0) S2T_W_TERM(1, 1) S2T_W_TERM(0, 2) }
        S2T_W_TERM(1, 0) S2T_W_TERM(0, 1) S2T_W_TERM(0, 0)
#undef S2T_W_TERM
      }
    }
    __syncthreads();
  }
  if (want_csum) {                         // the four row groups' partial sums -> one add per column
    float* red = reinterpret_cast<float*>(sm);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j)
      if (un[j].valid && un[j].isA) {
        const int idx = (tid & 255) + 256 * j;              // = kg (BM / 4) + mq
        const int kg = idx / (BM / 4), mq = idx - kg * (BM / 4);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) red[kg * BM + 4 * mq + mi] = cs[j][mi];
      }
    __syncthreads();
    if (tid < BM && m0 + tid < g.M)
      atomicAdd(g.colsum + m0 + tid, ((red[tid] + red[BM + tid]) + (red[2 * BM + tid] + red[3 * BM + tid])) * g.alpha);
  }
  if (producer || (g.debug & 1)) return;
  const int hi = lane >> 5, lo = lane & 31;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = n0 + wn + 32 * j + lo;
      if (c >= g.N) continue;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = m0 + wm + 32 * i + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
        if (row >= g.M) continue;
        atomicAdd(g.C + (long)row * g.ldc + c, acc[i][j][rr] * g.alpha);
      }
    }
}
// tile shape of a problem on the W form: 0 = 128 x 128, 1 = 128 x 192, 2 = 192 x 128
template <int PRO, bool PATCH = false>
__device__ __forceinline__ void tn_w_shape(const GemmArgs& g, int shape, unsigned bid, unsigned char* sm) {
  if (shape == 1) tn_w_body<2, 3, PRO, PATCH>(g, bid, sm);
  else if (shape == 2) tn_w_body<3, 2, PRO, PATCH>(g, bid, sm);
  else tn_w_body<2, 2, PRO, PATCH>(g, bid, sm);
}
constexpr int TNW_LDS = 2 * (128 + 192) * 32 * 6;   // 120 KB: the widest shape's two buffers
constexpr int TNW_LDS0 = 2 * (128 + 128) * 32 * 6;  // 96 KB: 128 x 128 tiles

// X3 (TN only): the contraction runs on the bf16 matrix cores -- both operand fragments are split
// exactly into three bf16 pieces when a wave reads them from LDS (8 k-strided reads per
// fragment: the tiles are k-major) and every 16-deep step is six v_mfma_f32_32x32x16_bf16 products
// (fp32-level error); 6 x 8 passes instead of 8 x 16 passes of the f32 MFMA.
template <int TM, int TN, int MODE, int PRO, bool X3 = false, bool PATCH = false, bool P3 = false>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const unsigned bid) {
  if constexpr (P3 && X3 && MODE == MODE_TN && !PATCH) {
    tn_p3_body<TM, TN, PRO>(g, bid);
    return;
  }
  constexpr bool A_KC = MODE != MODE_TN, B_KC = MODE == MODE_NT;
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int BK = bk_of(TM, TN, MODE);
  // the staged activation applies to A in NT (y = act(x) W^T) and to B in TN (dW = g^T act(x))
  using TA = Tile<BM, A_KC, (MODE == MODE_NT ? PRO : ACT_NONE), BK, (PATCH && MODE == MODE_NT)>;
  using TB = Tile<BN, B_KC, (MODE == MODE_TN ? PRO : ACT_NONE), BK, (PATCH && MODE == MODE_TN)>;
  __shared__ __attribute__((aligned(16))) float sA[TA::SIZE];
  __shared__ __attribute__((aligned(16))) float sB[TB::SIZE];

  // XCD-aware tile order: blocks that land on one XCD (block id % 8) walk a contiguous range of
  // tiles, n fastest, so the n-tiles of one m-panel share that XCD's L2 copy of the A panel
  const int total = g.tiles_m * g.tiles_n;
  const int per_xcd = (total + 7) / 8;
  // TN: 1-D grid of tiles x slices.  Workgroups are dealt to the XCDs round-robin, so XCD x gets
  // blocks x, x + 8, ...: those walk the tiles of slice x, then of slice x + 8, ... -- all tiles
  // of one contraction slice run on ONE XCD and share its L2 copy of that slice's operand rows.
  int lin, zslice = 0;
  if (MODE == MODE_TN) {
    const int q = (int)(bid >> 3);
    zslice = (int)(bid & 7) + 8 * (q / total);
    lin = q % total;
    if (zslice >= g.splits) return;
  } else {
    lin = (int)((bid & 7) * per_xcd + (bid >> 3));
    if (lin >= total) return;
  }
  const int tm = lin / g.tiles_n, tn = lin % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (MODE == MODE_TN && g.sym_cg > 0) {
    // symmetric product: the tile below the diagonal is the transpose of one above it, and a tile
    // whose first column group lies beyond its last row group holds no pair the caller reads
    if (tn < tm || n0 / g.sym_cg > (m0 + BM - 1) / g.sym_cg) return;
  }
  const int csum_tn = (MODE == MODE_TN && g.sym_cg > 0) ? tm : 0;   // the tile that owns colsum
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

  int kbeg = 0, kend = g.K;
  if (MODE == MODE_TN) {
    kbeg = zslice * g.kper;
    kend = min(g.K, kbeg + g.kper);
    if (kbeg >= kend) return;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[TA::NV], rb[TB::NV];
  unsigned oka = TA::load(ra, g.A, g.lda, m0, g.M, kbeg, kend, g.pt);
  unsigned okb = TB::load(rb, g.B, g.ldb, n0, g.N, kbeg, kend, g.pt);
  float csum = 0.f;   // TN bias gradient: thread t < BM owns column m0 + t
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();                       // previous chunk fully consumed
    TA::store(sA, ra, oka);
    TB::store(sB, rb, okb);
    __syncthreads();
    if (k0 + BK < kend) {                  // next chunk's global loads fly under the MFMAs
      oka = TA::load(ra, g.A, g.lda, m0, g.M, k0 + BK, kend, g.pt);
      okb = TB::load(rb, g.B, g.ldb, n0, g.N, k0 + BK, kend, g.pt);
    }
    if (MODE == MODE_TN && g.colsum != nullptr && tn == csum_tn && threadIdx.x < BM) {
#pragma unroll 8
      for (int r = 0; r < BK; ++r) csum += sA[r * TA::LD + threadIdx.x];
    }
    if (X3) {
      const bool np3 = g.np != 2;        // (uniform)
      const int lo3 = lane & 31, hi3 = lane >> 5;
#pragma unroll
      for (int s = 0; s < BK / 16; ++s) {
        bf16x8 pa[TM][3], pb[TN][3];
        // the lane's 8 contraction values k = 16 s + 8 hi .. + 7 of output index `col`:
        // kc tiles ([out][k], k contiguous): two 16-byte reads; k-major tiles: 8 strided reads
        auto gather = [&](const float* __restrict__ t, int ldt, int col, bool kc, bf16x8 (&out)[3]) {
          float v[8];
          if (kc) {
            const float4 u0 = *reinterpret_cast<const float4*>(t + col * ldt + 16 * s + 8 * hi3);
            const float4 u1 = *reinterpret_cast<const float4*>(t + col * ldt + 16 * s + 8 * hi3 + 4);
            v[0] = u0.x; v[1] = u0.y; v[2] = u0.z; v[3] = u0.w;
            v[4] = u1.x; v[5] = u1.y; v[6] = u1.z; v[7] = u1.w;
          } else {
            const float* q = t + (16 * s + 8 * hi3) * ldt + col;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = q[e * ldt];
          }
          uint4 q0, q1, q2;
          if (np3) {
            split_pair(v[0], v[1], q0.x, q1.x, q2.x);
            split_pair(v[2], v[3], q0.y, q1.y, q2.y);
            split_pair(v[4], v[5], q0.z, q1.z, q2.z);
            split_pair(v[6], v[7], q0.w, q1.w, q2.w);
            out[2] = __builtin_bit_cast(bf16x8, q2);
          } else {
            split_pair2(v[0], v[1], q0.x, q1.x);
            split_pair2(v[2], v[3], q0.y, q1.y);
            split_pair2(v[4], v[5], q0.z, q1.z);
            split_pair2(v[6], v[7], q0.w, q1.w);
          }
          out[0] = __builtin_bit_cast(bf16x8, q0);
          out[1] = __builtin_bit_cast(bf16x8, q1);
        };
#pragma unroll
        for (int i = 0; i < TM; ++i) gather(sA, TA::LD, wm + 32 * i + lo3, A_KC, pa[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) gather(sB, TB::LD, wn + 32 * j + lo3, B_KC, pb[j]);
#define S2T_X3_TERM(PA, PB)                                                                     \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i][PA], pb[j][PB], acc[i][j], 0, 0, 0);
        if (np3) { S2T_X3_TERM(2, 0) S2T_X3_TERM(1, 1) S2T_X3_TERM(0, 2) }
        S2T_X3_TERM(1, 0) S2T_X3_TERM(0, 1)
        S2T_X3_TERM(0, 0)
#undef S2T_X3_TERM
      }
      continue;
    }
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = TA::frag(sA, wm + 32 * i, s, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = TB::frag(sB, wn + 32 * j, s, lane);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
  }
  if (MODE == MODE_TN && g.colsum != nullptr && tn == csum_tn && threadIdx.x < BM &&
      m0 + (int)threadIdx.x < g.M)
    atomicAdd(g.colsum + m0 + threadIdx.x, csum * g.alpha);

  // ---- epilogue: lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int hi = lane >> 5, lo = lane & 31;
  if (MODE != MODE_TN && g.wide_ep) {
    // NT / NN with 16-byte-aligned rows: every 32 x 32 tile goes through a per-wave LDS scratch
    // (the operand tiles are dead; 16 rows at a time) and leaves as 16-byte stores -- 4 store
    // instructions per tile and lane instead of 16, bias and residual read as float4 too
    __syncthreads();                                   // all waves finished reading sA / sB
    float* scr = sA + wave * (16 * 36);                // 16 rows at a time: 4 x 2.3 KB fit in sA
    const int er = lane >> 3, ec = (lane & 7) * 4;     // this lane's row (of 8) and column quad
    float sq_c = 0.f, sq_o = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + 32 * j + ec;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias && col < g.N) bv = *reinterpret_cast<const float4*>(g.bias + col);
#pragma unroll
        for (int h = 0; h < 2; ++h) {                  // rows 16 h .. 16 h + 15 of the tile
          float4 rv[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int row = min(m0 + wm + 32 * i + 16 * h + er + 8 * q, g.M - 1);
            rv[q] = (g.resid && col < g.N)
                        ? *reinterpret_cast<const float4*>(g.resid + (long)row * g.ldr + col)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int r = 0; r < 8; ++r)
            scr[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + lo] = acc[i][j][8 * h + r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int row = m0 + wm + 32 * i + 16 * h + er + 8 * q;
            const float4 v = *reinterpret_cast<const float4*>(scr + (er + 8 * q) * 36 + ec);
            if (row < g.M && col < g.N) {
              const float4 o = make_float4(v.x + bv.x + rv[q].x, v.y + bv.y + rv[q].y, v.z + bv.z + rv[q].z,
                                           v.w + bv.w + rv[q].w);
              *reinterpret_cast<float4*>(g.C + (long)row * g.ldc + col) = o;
              if (g.sq_sums) {
                const float4 t = *reinterpret_cast<const float4*>(g.sq_other + (long)row * g.ld_sq + col);
                sq_c += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                sq_o += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
        }
      }
    if (g.sq_sums) {                                   // one atomic pair per workgroup
      sq_c = wave_sum(sq_c);
      sq_o = wave_sum(sq_o);
      __syncthreads();
      if (lane == 0) { sB[2 * wave] = sq_o; sB[2 * wave + 1] = sq_c; }
      __syncthreads();
      if (threadIdx.x == 0) {
        atomicAdd(g.sq_sums, (sB[0] + sB[2]) + (sB[4] + sB[6]));
        atomicAdd(g.sq_sums + 1, (sB[1] + sB[3]) + (sB[5] + sB[7]));
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + lo;
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (row >= g.M) continue;
        float v = acc[i][j][r];
        float* cp = g.C + (long)row * g.ldc + col;
        if (g.debug & 1) continue;
        if (MODE == MODE_TN) {
          atomicAdd(cp, v * g.alpha);
        } else {
          v += bv;
          if (g.act_src) v *= swoosh_deriv(g.act_src[(long)row * g.lds + col], g.act_kind);
          if (g.resid) v += g.resid[(long)row * g.ldr + col];
          if (g.accumulate) v += *cp;
          *cp = v;
        }
      }
    }
}

template <int TM, int TN, int MODE, int PRO, bool X3 = false, bool PATCH = false, bool P3 = false>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  gemm_body<TM, TN, MODE, PRO, X3, PATCH, P3>(g, blockIdx.x);
}

// ---- batched: blockIdx.y = batch item, operands `stride` floats apart (the nonlinear attention's
// W0 @ x products: 64 independent T x T x C problems that rocBLAS' strided-batched kernels run
// at 20-30 TFLOP/s at T = 248).
template <int TM, int TN, int MODE>
__global__ __launch_bounds__(256) void gemm_batched_kernel(GemmArgs g, long sA, long sB, long sC) {
  GemmArgs h = g;
  h.A += (long)blockIdx.y * sA;
  h.B += (long)blockIdx.y * sB;
  h.C += (long)blockIdx.y * sC;
  gemm_body<TM, TN, MODE, ACT_NONE, true, false, (MODE == MODE_TN)>(h, blockIdx.x);
}

// ---- grouped TN: the weight-gradient GEMMs of one layer in ONE launch.  Each problem keeps its
// own (tiles x slices) block range (a multiple of 8 blocks, so the slice -> XCD mapping of the
// single-problem launch holds); a block finds its problem by a scan of the prefix table.
constexpr int MAXG = 24;
struct TnProb {
  const float* A;
  const float* B;
  float* C;
  float* colsum;
  int lda, ldb, ldc, M, N, K, kper, tiles_m, tiles_n, splits;
  float alpha;
  int shape;               // W form: 0 = 128 x 128 tiles, 1 = 128 x 192, 2 = 192 x 128
};
struct TnGroup {
  int n;
  int debug;
  int np;                  // pieces per operand (GemmArgs::np)
  unsigned begin[MAXG + 1];
  TnProb p[MAXG];
};

template <bool X3, int TNW = 1, bool P3 = false, int TMW = 1>
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(TnGroup grp) {
  int i = 0;
  while (i + 1 < grp.n && blockIdx.x >= grp.begin[i + 1]) ++i;
  const TnProb& q = grp.p[i];
  GemmArgs g{q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, nullptr, nullptr, 0, nullptr, 0,
             0, 0, 0, q.colsum, 0, q.kper, q.tiles_m, q.tiles_n, q.splits, grp.debug, q.alpha};
  g.np = grp.np;
  gemm_body<TMW, TNW, MODE_TN, ACT_NONE, X3, false, P3>(g, blockIdx.x - grp.begin[i]);
}

// the all-waves form with a two-piece LDS image (three-product arithmetic only: 32 KB at 128 x 128)
template <int TM, int TN, int PRO>
__global__ __launch_bounds__(256) void gemm_tn_p2_kernel(GemmArgs g) {
  tn_p3_body<TM, TN, PRO, 2>(g, blockIdx.x);
}
template <int TM, int TN, int NP>
__global__ __launch_bounds__(256) void gemm_tn_patch_kernel(GemmArgs g) {
  tn_p3_body<TM, TN, ACT_NONE, NP, true>(g, blockIdx.x);
}
template <int TMW, int TNW>
__global__ __launch_bounds__(256) void gemm_tn_grouped_p2_kernel(TnGroup grp) {
  int i = 0;
  while (i + 1 < grp.n && blockIdx.x >= grp.begin[i + 1]) ++i;
  const TnProb& q = grp.p[i];
  GemmArgs g{q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, nullptr, nullptr, 0, nullptr, 0,
             0, 0, 0, q.colsum, 0, q.kper, q.tiles_m, q.tiles_n, q.splits, grp.debug, q.alpha};
  g.np = 2;
  tn_p3_body<TMW, TNW, ACT_NONE, 2>(g, blockIdx.x - grp.begin[i]);
}

// the grouped launch on the W form (tiles_m / tiles_n of the problems count that problem's tiles)
__global__ __launch_bounds__(512) void gemm_tn_grouped_w_kernel(TnGroup grp) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char tnw_sm[];
  int i = 0;
  while (i + 1 < grp.n && blockIdx.x >= grp.begin[i + 1]) ++i;
  const TnProb& q = grp.p[i];
  GemmArgs g{q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K, nullptr, nullptr, 0, nullptr, 0,
             0, 0, 0, q.colsum, 0, q.kper, q.tiles_m, q.tiles_n, q.splits, grp.debug, q.alpha};
  g.np = grp.np;
  tn_w_shape<ACT_NONE>(g, q.shape, blockIdx.x - grp.begin[i], tnw_sm);
}
template <int PRO>
__global__ __launch_bounds__(512) void gemm_tn_w_kernel(GemmArgs g, int shape) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char tnw_sm[];
  tn_w_shape<PRO>(g, shape, blockIdx.x, tnw_sm);
}
// the implicit-patch form: B = the 3x3 patch matrix of a channel-last map (conv weight gradient)
__global__ __launch_bounds__(512) void gemm_tn_w_patch_kernel(GemmArgs g, int shape) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char tnw_sm[];
  tn_w_shape<ACT_NONE, true>(g, shape, blockIdx.x, tnw_sm);
}

// Weight-gradient contractions on the bf16 matrix cores (three-way exact split, six products:
// fp32-level error, tests/test_gpu_gemm.py) -- on by default: in the training step the TN GEMMs
// share the CUs with the main stream's library GEMMs, and 2.3x fewer matrix-pipe cycles for the
// same product took the C3 step from 47.8 to 46.6 ms (same box, 3 x 3 runs).  S2T_TN_X3=0: the
// f32 MFMA form.
// the same arithmetic for the NT / NN products of s2t_gemm_f32 (default on; S2T_NN_X3=0: f32 MFMA)
static int g_nn_x3 = -1;
static bool nn_x3() {
  if (g_nn_x3 < 0) { const char* e = getenv("S2T_NN_X3"); g_nn_x3 = e ? atoi(e) : 1; }
  return g_nn_x3 == 1;
}
extern "C" int s2t_nn_x3(int set) {
  if (set >= 0) g_nn_x3 = set ? 1 : 0;
  return nn_x3() ? 1 : 0;
}
// S2T_TN_P3=0: the form that splits every fragment where it is read (A/B, tests)
static bool tn_p3() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("S2T_TN_P3"); v = e ? atoi(e) : 1; }
  return v == 1;
}
// S2T_TN_W=0: weight gradients on the 64 x 64 form instead of the wave-specialised "W" form;
// S2T_TN_W_BLOCKS: workgroups a W launch aims at (one is resident per CU)
// Round 6: with no setting the form follows the weight gradients' arithmetic -- six products: the W form
// (round 5: 0.65 ms per step better than the all-waves form); three products: the all-waves form on
// 128 x 128 tiles (the split and the MFMAs both shrank, the producer / consumer imbalance did not:
// 33.80-33.90 against 33.33-33.46 ms per step, three pairs on one box).
static int g_tn_w = -1;      // -1: not asked yet; 2: automatic; 0 / 1: forced (S2T_TN_W, s2t_tn_w)
static bool tn_w_forced() {
  if (g_tn_w < 0) { const char* e = getenv("S2T_TN_W"); g_tn_w = e ? (atoi(e) ? 1 : 0) : 2; }
  return g_tn_w != 2;
}
static bool tn_w() { return tn_w_forced() ? g_tn_w == 1 : s2t_gemm_arith_of(2) != 2; }
// the 3x3 convolution's implicit-patch weight gradient has no 128 x 128 all-waves instantiation: W unless forced off
static bool tn_w_patch() { return tn_w_forced() ? g_tn_w == 1 : true; }
static long tn_w_blocks() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("S2T_TN_W_BLOCKS"); v = e ? atol(e) : 512; }
  return v;
}
// tile shape of an (M x N) output on the W form and its tile counts
static int tn_w_shape_of(int M, int N, int& tiles_m, int& tiles_n) {
  const auto waste = [](int n, int t) { return ((n + t - 1) / t) * t - n; };
  static int wide = -1;      // S2T_TN_W_WIDE=0: 128 x 128 tiles only (96 KB of LDS instead of 120)
  if (wide < 0) { const char* e = getenv("S2T_TN_W_WIDE"); wide = e ? atoi(e) : 1; }
  int shape = 0;
  if (!wide) shape = 0;
  else if (waste(N, 192) < waste(N, 128)) shape = 1;
  else if (waste(M, 192) < waste(M, 128)) shape = 2;
  tiles_m = (M + (shape == 2 ? 191 : 127)) / (shape == 2 ? 192 : 128);
  tiles_n = (N + (shape == 1 ? 191 : 127)) / (shape == 1 ? 192 : 128);
  return shape;
}
template <typename K>
static bool tn_w_prepare(K kern) {      // the kernels take 120 KB of dynamic LDS
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             TNW_LDS) == hipSuccess;
}
static int g_tn_x3 = -1;
static bool tn_x3() {
  if (g_tn_x3 < 0) { const char* e = getenv("S2T_TN_X3"); g_tn_x3 = e ? atoi(e) : 1; }
  return g_tn_x3 == 1;
}

template <int TM, int TN, int MODE, int PRO>
int launch(GemmArgs& g, int splits, hipStream_t st) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  g.tiles_m = (g.M + BM - 1) / BM;
  g.tiles_n = (g.N + BN - 1) / BN;
  { static const int dbg = s2t_debug_env("S2T_GEMM_DEBUG"); g.debug = dbg; }
  const int total = g.tiles_m * g.tiles_n;
  g.splits = splits;
  const int grid = MODE == MODE_TN ? 8 * total * ((splits + 7) / 8) : ((total + 7) / 8) * 8;
  static const bool p2 = [] { const char* e = getenv("S2T_TN_P2"); return !e || atoi(e) != 0; }();
  if constexpr (MODE == MODE_TN && TM == 2 && TN == 2) {
    if (tn_x3() && tn_p3() && g.np == 2 && p2) {
      hipLaunchKernelGGL((gemm_tn_p2_kernel<TM, TN, PRO>), dim3(grid), dim3(256), 0, st, g);
      return (int)hipGetLastError();
    }
  }
  if (MODE == MODE_TN && tn_x3() && tn_p3())
    hipLaunchKernelGGL((gemm_kernel<TM, TN, MODE, PRO, true, false, (MODE == MODE_TN)>), dim3(grid), dim3(256), 0, st, g);
  else if (MODE == MODE_TN ? tn_x3() : nn_x3())
    hipLaunchKernelGGL((gemm_kernel<TM, TN, MODE, PRO, true>), dim3(grid), dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<TM, TN, MODE, PRO>), dim3(grid), dim3(256), 0, st, g);
  return (int)hipGetLastError();
}

template <int TM, int TN, int MODE>
int launch_p(GemmArgs& g, int pro, int splits, hipStream_t st) {
  if (MODE == MODE_NN || pro == ACT_NONE) return launch<TM, TN, MODE, ACT_NONE>(g, splits, st);
  if (pro == ACT_SWOOSH_L) return launch<TM, TN, MODE, ACT_SWOOSH_L>(g, splits, st);
  return launch<TM, TN, MODE, ACT_SWOOSH_R>(g, splits, st);
}

// tile shapes: (2,2) 128x128, (2,3) 128x192, (2,1) 128x64, (1,2) 64x128, (1,1) 64x64
template <int MODE>
int launch_t(GemmArgs& g, int tm, int tn, int pro, int splits, hipStream_t st) {
  if (tm == 2 && tn == 3) return launch_p<2, 3, MODE>(g, pro, splits, st);
  if (tm == 2 && tn == 1) return launch_p<2, 1, MODE>(g, pro, splits, st);
  if (tm == 2) return launch_p<2, 2, MODE>(g, pro, splits, st);
  if (tn >= 2) return launch_p<1, 2, MODE>(g, pro, splits, st);
  return launch_p<1, 1, MODE>(g, pro, splits, st);
}

// columns per block: 192 when that wastes less than 128 (N = 192, 384, 576, 960 ...), else 128/64
int pick_tn(int N) {
  const int w128 = ((N + 127) / 128) * 128 - N, w192 = ((N + 191) / 192) * 192 - N;
  if (N <= 64) return 1;
  return (w192 < w128) ? 3 : 2;
}

constexpr int KR = 64;   // TN slices are multiples of the deepest chunk

template <int MODE>
int dispatch(GemmArgs& g, hipStream_t st) {
  // weight gradients are class W whatever the caller's scope says; the symmetric x^T x is a statistic
  g.np = MODE == MODE_TN ? s2t_gemm_arith_of(g.sym_cg > 0 ? 3 : 2) : s2t_gemm_arith();
  int tn_sel = pick_tn(g.N);
  const int pro = MODE == MODE_NT ? g.pro_a : (MODE == MODE_TN ? g.pro_b : 0);
  long tiles_big = (long)((g.M + 127) / 128) * ((g.N + 64 * tn_sel - 1) / (64 * tn_sel));
  if (MODE == MODE_TN) {
    // The output (features x features) is small and the contraction long: split it over the chip
    // and add every slice's tile with fp32 atomics.  Atomic bytes = workgroups x tile bytes and
    // the chip adds ~1.3 TB/s, so SMALL tiles win: 64x64 tiles at ~6 workgroups per CU moved the
    // C3 shapes from ~50 to ~75 TFLOP/s against 128x192 tiles at 2 per CU (tools/bench_tn.py);
    // 64x128 once the output is wide enough to give the slices enough tiles.
    // S2T_TN_TILE ("11", "12", "21", "22", "23") / S2T_TN_BLOCKS override the choice for tuning.
    // Round 5: aligned, non-symmetric problems take the wave-specialised W form first (tn_w_body).
    // Alone on the chip it is SLOWER than the 64 x 64 form (10 C3 shapes: 700 against 535 us: one
    // workgroup per CU, 40 KB of loads in flight), inside the training step it is faster (same box:
    // 39.2 -> 38.55 ms/step): the step runs these launches beside the main stream's GEMMs, where
    // what counts is the issue slots and LDS/L2 bytes a launch takes from them, and the W form
    // takes fewer of each per flop (bigger tiles, 16-byte loads, half the split work per product).
    // S2T_GEMM_DEBUG bits (timing ablations, results wrong): 1 no output adds, 2 no split/store,
    // 4 no MFMA, 8 no loads, 16 print the launch.
    static int force = -1, user_blocks = -2;
    if (force < 0) { const char* e = getenv("S2T_TN_TILE"); force = e ? atoi(e) : 0; }
    if (user_blocks == -2) { const char* e = getenv("S2T_TN_BLOCKS"); user_blocks = e ? atoi(e) : -1; }
    if (tn_w() && tn_x3() && tn_p3() && !g.sym_cg && force <= 0 && !((g.M | g.N | g.lda | g.ldb) & 3) &&
        !((reinterpret_cast<uintptr_t>(g.A) | reinterpret_cast<uintptr_t>(g.B)) & 15)) {
      const int shape = tn_w_shape_of(g.M, g.N, g.tiles_m, g.tiles_n);
      { static const int dbg = s2t_debug_env("S2T_GEMM_DEBUG"); g.debug = dbg; }
      const long tiles = (long)g.tiles_m * g.tiles_n;
      const long target = user_blocks > 0 ? user_blocks : tn_w_blocks();
      int splits = (int)((target + tiles - 1) / tiles);
      splits = std::max(1, std::min(splits, (g.K + 2 * KR - 1) / (2 * KR)));
      int kper = (g.K + splits - 1) / splits;
      kper = ((kper + KR - 1) / KR) * KR;
      g.kper = kper;
      g.splits = (g.K + kper - 1) / kper;
      const int grid = (int)(8 * tiles * ((g.splits + 7) / 8));
      if (g.debug & 16) fprintf(stderr, "[tn_w] M %d N %d K %d shape %d splits %d kper %d grid %d pro %d\n", g.M, g.N, g.K, shape, g.splits, g.kper, grid, pro);
      static const bool ok = tn_w_prepare(gemm_tn_w_kernel<ACT_NONE>) && tn_w_prepare(gemm_tn_w_kernel<ACT_SWOOSH_L>) &&
                             tn_w_prepare(gemm_tn_w_kernel<ACT_SWOOSH_R>);
      if (!ok) return -3;
      const int lds = shape ? TNW_LDS : TNW_LDS0;
      if (pro == ACT_NONE) hipLaunchKernelGGL((gemm_tn_w_kernel<ACT_NONE>), dim3(grid), dim3(512), lds, st, g, shape);
      else if (pro == ACT_SWOOSH_L) hipLaunchKernelGGL((gemm_tn_w_kernel<ACT_SWOOSH_L>), dim3(grid), dim3(512), lds, st, g, shape);
      else hipLaunchKernelGGL((gemm_tn_w_kernel<ACT_SWOOSH_R>), dim3(grid), dim3(512), lds, st, g, shape);
      return (int)hipGetLastError();
    }
    // (no forced tile: 128 x 128 under the three-product arithmetic -- see tn_w_forced -- else 64 x 64 / 64 x 128)
    const bool big = force <= 0 && !g.sym_cg && g.np == 2 && tn_x3() && tn_p3();
    const int ttm = (force > 0 && !g.sym_cg) ? force / 10 : (big ? 2 : 1);
    const int ttn = g.sym_cg ? 1 : (force > 0 ? force % 10 : (big ? 2 : (g.N >= 512 ? 2 : 1)));
    const long tiles = (long)((g.M + 64 * ttm - 1) / (64 * ttm)) * ((g.N + 64 * ttn - 1) / (64 * ttn));
    static int xtx_blocks = -2;   // S2T_XTX_BLOCKS: workgroup target of the symmetric x^T x
    if (xtx_blocks == -2) { const char* e = getenv("S2T_XTX_BLOCKS"); xtx_blocks = e ? atoi(e) : -1; }
    const int target = (g.sym_cg && xtx_blocks > 0) ? xtx_blocks
                       : user_blocks > 0 ? user_blocks : (ttm * ttn == 1 ? 1536 : 768);
    int splits = (int)((target + tiles - 1) / tiles);
    const int maxs = (g.K + 2 * KR - 1) / (2 * KR);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int kper = (g.K + splits - 1) / splits;
    kper = ((kper + KR - 1) / KR) * KR;
    g.kper = kper;
    splits = (g.K + kper - 1) / kper;
    return launch_t<MODE>(g, ttm, ttn, pro, splits, st);
  }
  {
    static int wide = -1;       // S2T_GEMM_WIDE_EP=0: the scalar epilogue
    if (wide < 0) { const char* e = getenv("S2T_GEMM_WIDE_EP"); wide = e ? atoi(e) : 1; }
    g.wide_ep = wide && !g.act_src && !g.accumulate && (g.N & 3) == 0 && (g.ldc & 3) == 0 &&
                (reinterpret_cast<uintptr_t>(g.C) & 15) == 0 &&
                (!g.bias || (reinterpret_cast<uintptr_t>(g.bias) & 15) == 0) &&
                (!g.resid || ((g.ldr & 3) == 0 && (reinterpret_cast<uintptr_t>(g.resid) & 15) == 0));
  }
  static int nt_force = -1;     // S2T_NT_TILE = "tm tn" digits (11, 12, 21, 22, 23): tuning
  if (nt_force < 0) { const char* e = getenv("S2T_NT_TILE"); nt_force = e ? atoi(e) : 0; }
  if (g.tile_force > 0) return launch_t<MODE>(g, g.tile_force / 10, g.tile_force % 10, pro, 1, st);
  if (nt_force > 0) return launch_t<MODE>(g, nt_force / 10, nt_force % 10, pro, 1, st);
  if (tiles_big >= 384) return launch_t<MODE>(g, 2, tn_sel, pro, 1, st);
  if (tn_sel == 3) tn_sel = 2;
  return launch_t<MODE>(g, 1, tn_sel, pro, 1, st);
}

}  // namespace

extern "C" int s2t_gemm_f32(int mode, const float* A, long lda, const float* B, long ldb, float* C,
                            long ldc, int M, int N, int K, const float* bias, const float* resid,
                            long ldr, const float* act_src, long lds, int act_kind, int pro_a,
                            int pro_b, float* colsum, int accumulate, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  // float4 staging: k-contiguous operands need K % 4 == 0 and 16-byte aligned rows; k-major
  // operands need 16-byte aligned rows (ragged right edges are handled in the loader)
  const bool a_kc = mode != MODE_TN, b_kc = mode == MODE_NT;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (lda & 3) || (ldb & 3))
    return -2;
  if ((a_kc || b_kc) && (K & 3)) return -2;
  if ((!a_kc && (M & 3)) || (!b_kc && (N & 3))) return -2;    // k-major operands: 4 outputs per load
  if (M < 4 || N < 4 || K < 4) return -2;
  if (act_kind < 0 || act_kind > 2 || pro_a < 0 || pro_a > 2 || pro_b < 0 || pro_b > 2) return -1;
  if (mode == MODE_TN && (resid || act_src || bias)) return -1;
  GemmArgs g{A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, act_src, lds, act_kind, pro_a,
             pro_b, colsum, accumulate, 0, 0, 0, 0, 0, 1.f, 0};
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (mode == MODE_NT) rc = dispatch<MODE_NT>(g, st);
  else if (mode == MODE_NN) rc = dispatch<MODE_NN>(g, st);
  else if (mode == MODE_TN) rc = dispatch<MODE_TN>(g, st);
  else return -1;
  return rc;
}

// `batch` independent products in one launch (bf16x3 matrix-core arithmetic, fp32-level error):
//   mode 0 (NT): C_b[M,N]  = A_b[M,K] . B_b[N,K]^T      mode 1 (NN): C_b[M,N] = A_b[M,K] . B_b[K,N]
//   mode 2 (TN): C_b[M,N] += A_b[K,M]^T . B_b[K,N]      (ADDED with fp32 atomics: zero C first)
// X_b = X + b * sX floats.  Same alignment rules as s2t_gemm_f32 (every k-contiguous operand needs
// K % 4 == 0, every row start 16-byte aligned, strides included): -2 when they do not hold -- the
// caller keeps the library for such shapes (T = 495, 62 of the C3 stacks).
// s2t_gemm_f32 modes 0 / 1 (no residual / activation) that also ADDS sums[0] += ||other||_F^2 and
// sums[1] += ||C||_F^2 (other: an (M, N) matrix, rows ld_other apart): Whiten's backward needs both norms
// of (g, x dcov) before it can combine them (reference model/layer/scaling.py:1024-1027).
// -2: operands outside the 16-byte epilogue's rules (the caller runs the separate pass).
extern "C" int s2t_gemm_f32_sq(int mode, const float* A, long lda, const float* B, long ldb, float* C,
                               long ldc, int M, int N, int K, const float* bias, const float* other,
                               long ld_other, float* sums, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !other || !sums || (mode != MODE_NT && mode != MODE_NN)) return -1;
  const bool b_kc = mode == MODE_NT;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (lda & 3) || (ldb & 3) ||
      (K & 3) || (!b_kc && (N & 3)) || M < 4 || N < 4 || K < 4)
    return -2;
  if ((N & 3) || (ldc & 3) || (ld_other & 3) || (reinterpret_cast<uintptr_t>(C) & 15) ||
      (reinterpret_cast<uintptr_t>(other) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
    return -2;
  static const bool wide_on = [] { const char* e = getenv("S2T_GEMM_WIDE_EP"); return !e || atoi(e) != 0; }();
  if (!wide_on) return -2;
  GemmArgs g{A, lda, B, ldb, C, ldc, M, N, K, bias, nullptr, 0, nullptr, 0, 0, 0,
             0, nullptr, 0, 0, 0, 0, 0, 0, 1.f, 0};
  g.sq_other = other;
  g.ld_sq = ld_other;
  g.sq_sums = sums;
  hipStream_t st = (hipStream_t)stream;
  return mode == MODE_NT ? dispatch<MODE_NT>(g, st) : dispatch<MODE_NN>(g, st);
}

extern "C" int s2t_gemm_f32_batched(int mode, const float* A, long lda, long sA, const float* B,
                                    long ldb, long sB, float* C, long ldc, long sC, int M, int N,
                                    int K, int batch, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
  if (mode < 0 || mode > 2 || batch > 65535) return -1;
  const bool a_kc = mode != MODE_TN, b_kc = mode == MODE_NT;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (lda & 3) || (ldb & 3) || (sA & 3) || (sB & 3))
    return -2;
  // (the output needs 16-byte rows only for the float4 epilogue; T = 495 outputs take the scalar one)
  const bool c_vec = (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (sC & 3) == 0 && (ldc & 3) == 0 && (N & 3) == 0;
  if (mode == MODE_TN && !c_vec) return -2;
  if ((a_kc || b_kc) && (K & 3)) return -2;
  if ((!a_kc && (M & 3)) || (!b_kc && (N & 3))) return -2;
  if (M < 4 || N < 4 || K < 4) return -2;
  GemmArgs g{A, lda, B, ldb, C, ldc, M, N, K, nullptr, nullptr, 0, nullptr, 0, 0, 0,
             0, nullptr, 0, 0, 0, 0, 0, 0, 1.f, 0};
  hipStream_t st = (hipStream_t)stream;
  g.np = s2t_gemm_arith();              // (products of two activations: the caller's class, all three layouts)
  g.tiles_m = (M + 63) / 64;
  g.tiles_n = (N + 63) / 64;
  const int total = g.tiles_m * g.tiles_n;
  if (mode == MODE_TN) {
    // the contraction is short (K = T): at most a few slices, each a multiple of the staging depth
    int splits = std::max(1, std::min(8, K / (2 * KR)));
    int kper = (K + splits - 1) / splits;
    kper = ((kper + KR - 1) / KR) * KR;
    g.kper = kper;
    g.splits = (K + kper - 1) / kper;
    const unsigned grid = (unsigned)(8 * total * ((g.splits + 7) / 8));
    hipLaunchKernelGGL((gemm_batched_kernel<1, 1, MODE_TN>), dim3(grid, batch), dim3(256), 0, st, g, sA, sB, sC);
  } else {
    g.splits = 1;
    g.wide_ep = c_vec;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    if (mode == MODE_NT)
      hipLaunchKernelGGL((gemm_batched_kernel<1, 1, MODE_NT>), dim3(grid, batch), dim3(256), 0, st, g, sA, sB, sC);
    else
      hipLaunchKernelGGL((gemm_batched_kernel<1, 1, MODE_NN>), dim3(grid, batch), dim3(256), 0, st, g, sA, sB, sC);
  }
  return (int)hipGetLastError();
}

// s2t_gemm_f32 (modes 0 / 1) with the block tile chosen by the caller: tile = "tm tn" digits, block
// tile (64 tm) x (64 tn), one of 11 12 21 22 23; 0 = the dispatcher's choice.  The plan cache of
// s2t_linear_lt times these against the library's kernels.
extern "C" int s2t_gemm_f32_tiled(int mode, const float* A, long lda, const float* B, long ldb,
                                  float* C, long ldc, int M, int N, int K, const float* bias,
                                  const float* resid, long ldr, int tile, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (mode != MODE_NT && mode != MODE_NN)) return -1;
  if (tile != 0 && tile != 11 && tile != 12 && tile != 21 && tile != 22 && tile != 23) return -1;
  const bool b_kc = mode == MODE_NT;
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) || (lda & 3) ||
      (ldb & 3) || (K & 3) || (!b_kc && (N & 3)) || M < 4 || N < 4 || K < 4)
    return -2;
  GemmArgs g{A, lda, B, ldb, C, ldc, M, N, K, bias, resid, ldr, nullptr, 0, 0, 0,
             0, nullptr, 0, 0, 0, 0, 0, 0, 1.f, 0, tile};
  hipStream_t st = (hipStream_t)stream;
  return mode == MODE_NT ? dispatch<MODE_NT>(g, st) : dispatch<MODE_NN>(g, st);
}

// 3x3 convolution products on channel-last maps with the patch matrix read in place (see Patch):
//   mode 0: y[R, CO] = patches(x)[R, 9C] . w2[CO, 9C]^T (+ bias)          R = B Ho Wo
//   mode 2: dw2[CO, 9C] += g[R, CO]^T . patches(x)[R, 9C],  db[CO] += column sums of g
// w2 / dw2 are in (cout, kh, kw, cin) order; bf16x3 matrix-core arithmetic (fp32-level error).
extern "C" int s2t_conv3x3_gemm(int mode, const float* x, int B, int H, int W, int C, int sh, int sw,
                                int CO, const float* w2_or_g, const float* bias, float* out,
                                float* db, void* stream) {
  if (B <= 0 || H < 3 || W < 3 || C <= 0 || CO <= 0 || sh <= 0 || sw <= 0 || (mode != 0 && mode != 2))
    return -1;
  const int Ho = (H - 3) / sh + 1, Wo = (W - 3) / sw + 1;
  const long R = (long)B * Ho * Wo, nx = (long)B * H * W * C;
  if ((C & 3) || (CO & 3) || nx >= (1L << 31) || R >= (1L << 31) || R < 4) return -2;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w2_or_g) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15))
    return -2;
  const Patch pt{3 * C, Ho * Wo, Wo, H * W * C, sh * W * C, sw * C, W * C};
  const int K9 = 9 * C;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    GemmArgs g{x, 0, w2_or_g, K9, out, CO, (int)R, CO, K9, bias, nullptr, 0, nullptr, 0, 0, 0,
               0, nullptr, 0, 0, 0, 0, 0, 0, 1.f, 0, 0, 0, pt};
    g.np = s2t_gemm_arith_of(0);
    g.wide_ep = !bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0;
    g.tiles_m = (g.M + 127) / 128;
    g.tiles_n = CO > 64 ? (CO + 127) / 128 : 1;
    g.splits = 1;
    const int grid = ((g.tiles_m * g.tiles_n + 7) / 8) * 8;
    if (CO > 64)
      hipLaunchKernelGGL((gemm_kernel<2, 2, MODE_NT, ACT_NONE, true, true>), dim3(grid), dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_kernel<2, 1, MODE_NT, ACT_NONE, true, true>), dim3(grid), dim3(256), 0, st, g);
    return (int)hipGetLastError();
  }
  GemmArgs g{w2_or_g, CO, x, 0, out, K9, CO, K9, (int)R, nullptr, nullptr, 0, nullptr, 0, 0, 0,
             0, db, 0, 0, 0, 0, 0, 0, 1.f, 0, 0, 0, pt};
  g.np = s2t_gemm_arith_of(2);
  { static const int dbg = s2t_debug_env("S2T_GEMM_DEBUG"); g.debug = dbg; }
  // the wave-specialised form (tn_w_body) for outputs of >= 128 x 1024 (the conformer's 256 -> 256 conv:
  // 24 tiles of 128 x 192).  The zipformer frontend's 32 -> 128 conv (128 x 288 output over 600 k rows:
  // 2 tiles, 256 slices) lasts 1.16 ms on it against 0.48 ms on the 64 x 64 form; the STEP is the same
  // either way (37.64 / 37.66 ms, three pairs), so the shorter launch is kept.
  // Round 6: under the three-product arithmetic every size runs on the all-waves form below with the patch
  // operand (C3 33.24-33.42 -> 33.02-33.15 ms per step against the fragment-splitting 64 x 64 form, C2
  // 19.30-19.42 -> 18.89-19.10 against the W form; S2T_CONV_W_P3=0: the round-5 forms)
  static const int p3mode = [] { const char* e = getenv("S2T_CONV_W_P3"); return e ? atoi(e) : 1; }();
  const bool allw = p3mode && tn_x3() && tn_p3();
  if (!(allw && g.np == 2) && tn_w_patch() && tn_x3() && tn_p3() && CO >= 128 && K9 >= 1024) {
    const int shape = tn_w_shape_of(g.M, g.N, g.tiles_m, g.tiles_n);
    const long tiles = (long)g.tiles_m * g.tiles_n;
    int splits = (int)((tn_w_blocks() + tiles - 1) / tiles);
    splits = std::max(1, std::min(splits, (g.K + 2 * KR - 1) / (2 * KR)));
    int kper = (g.K + splits - 1) / splits;
    kper = ((kper + KR - 1) / KR) * KR;
    g.kper = kper;
    g.splits = (g.K + kper - 1) / kper;
    const int grid = (int)(8 * tiles * ((g.splits + 7) / 8));
    static const bool ok = tn_w_prepare(gemm_tn_w_patch_kernel);
    if (!ok) return -3;
    hipLaunchKernelGGL(gemm_tn_w_patch_kernel, dim3(grid), dim3(512), shape ? TNW_LDS : TNW_LDS0, st, g, shape);
    return (int)hipGetLastError();
  }
  // the all-waves form that splits once, at staging (tn_p3_body with the patch operand): 64 TM x 128 tiles
  if (allw) {
    const int tmm = CO > 64 ? 2 : 1;
    g.tiles_m = (CO + 64 * tmm - 1) / (64 * tmm);
    g.tiles_n = (K9 + 127) / 128;
    const long tiles = (long)g.tiles_m * g.tiles_n;
    int splits = (int)((768 + tiles - 1) / tiles);
    splits = std::max(1, std::min(splits, (g.K + 2 * KR - 1) / (2 * KR)));
    int kper = (g.K + splits - 1) / splits;
    kper = ((kper + KR - 1) / KR) * KR;
    g.kper = kper;
    g.splits = (g.K + kper - 1) / kper;
    const int grid = (int)(8 * tiles * ((g.splits + 7) / 8));
    if (g.np == 2 && tmm == 2) hipLaunchKernelGGL((gemm_tn_patch_kernel<2, 2, 2>), dim3(grid), dim3(256), 0, st, g);
    else if (g.np == 2) hipLaunchKernelGGL((gemm_tn_patch_kernel<1, 2, 2>), dim3(grid), dim3(256), 0, st, g);
    else if (tmm == 2) hipLaunchKernelGGL((gemm_tn_patch_kernel<2, 2, 3>), dim3(grid), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_tn_patch_kernel<1, 2, 3>), dim3(grid), dim3(256), 0, st, g);
    return (int)hipGetLastError();
  }
  g.tiles_m = (CO + 63) / 64;
  g.tiles_n = (K9 + 63) / 64;
  const long tiles = (long)g.tiles_m * g.tiles_n;
  int splits = (int)((1536 + tiles - 1) / tiles);
  const int maxs = (g.K + 2 * KR - 1) / (2 * KR);
  splits = std::max(1, std::min(splits, maxs));
  int kper = (g.K + splits - 1) / splits;
  kper = ((kper + KR - 1) / KR) * KR;
  g.kper = kper;
  g.splits = (g.K + kper - 1) / kper;
  const int grid = (int)(8 * tiles * ((g.splits + 7) / 8));
  hipLaunchKernelGGL((gemm_kernel<1, 1, MODE_TN, ACT_NONE, true, true>), dim3(grid), dim3(256), 0, st, g);
  return (int)hipGetLastError();
}

extern "C" int s2t_tn_w(int set) {
  if (set >= 0) g_tn_w = set >= 2 ? 2 : (set ? 1 : 0);
  return tn_w() ? 1 : 0;
}
extern "C" int s2t_tn_x3(int set) {
  if (set >= 0) g_tn_x3 = set ? 1 : 0;
  return tn_x3() ? 1 : 0;
}

extern "C" int s2t_gemm_xtx(const float* x, long ldx, int R, int C, int cg, float* xtx, long ldc,
                            float* colsum, void* stream) {
  if (R <= 0 || C <= 0 || cg <= 0 || C % cg) return -1;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (ldx & 3) || (C & 3) || C < 4 || R < 4) return -2;
  GemmArgs g{x, ldx, x, ldx, xtx, ldc, C, C, R, nullptr, nullptr, 0, nullptr, 0, 0, 0,
             0, colsum, 0, 0, 0, 0, 0, 0, 1.f, cg};
  return dispatch<MODE_TN>(g, (hipStream_t)stream);
}

extern "C" int s2t_gemm_tn_grouped(int n, const S2tTnProblem* probs, void* stream) {
  if (n <= 0) return 0;
  if (!probs) return -1;
  hipStream_t st = (hipStream_t)stream;
  // Contraction slices per problem.  Every slice's tile is ADDED to the output with fp32 atomics,
  // so the atomic traffic is (slices x output bytes); a slice count chosen per problem to fill
  // the chip on its own (the single-problem rule: ~1536 blocks each) made a 17-problem layer
  // launch write 22x its output.  The problems of a group run concurrently, so the GROUP has to
  // fill the chip: slices = the multiple of 8 (one residue per XCD, see gemm_body) that brings
  // the group's total to ~S2T_TN_GROUP_BLOCKS blocks (default 6144 = 4 rounds of 6 per CU).
  static long target = -1;
  if (target < 0) {
    const char* e = getenv("S2T_TN_GROUP_BLOCKS");
    target = e ? atol(e) : 6144;
    if (target < 8) target = 8;
  }
  // output tile (64 tmw) x (64 tnw): S2T_TN_GROUP_TILE = 11 | 12 | 22; unset: 22 under the three-product
  // arithmetic (1536 workgroups per group unless S2T_TN_GROUP_BLOCKS says otherwise), else 11
  static int gtile = -1;
  static bool user_target = false;
  if (gtile < 0) {
    const char* e = getenv("S2T_TN_GROUP_TILE");
    gtile = e ? atoi(e) : 0;
    user_target = getenv("S2T_TN_GROUP_BLOCKS") != nullptr;
  }
  const int gt = gtile > 0 ? gtile : ((s2t_gemm_arith_of(2) == 2 && tn_x3() && tn_p3()) ? 22 : 11);
  const int tmw = gt == 22 ? 2 : 1, tnw = (gt == 12 || gt == 22) ? 2 : 1;
  const long gtarget = (gt == 22 && !user_target) ? 1536 : target;
  const long qtarget = tn_w_blocks();
  bool useq = tn_w() && tn_x3() && tn_p3();
  for (int i = 0; i < n && useq; ++i) {
    const S2tTnProblem& s = probs[i];
    useq = !((s.M & 3) || (s.N & 3) || (s.lda & 3) || (s.ldb & 3) ||
             (reinterpret_cast<uintptr_t>(s.A) & 15) || (reinterpret_cast<uintptr_t>(s.B) & 15));
  }
  if (useq) {
    static const bool ok = tn_w_prepare(gemm_tn_grouped_w_kernel);
    if (!ok) return -3;
  }
  for (int base = 0; base < n && useq; base += MAXG) {
    TnGroup grp;
    { static const int dbg = s2t_debug_env("S2T_GEMM_DEBUG"); grp.debug = dbg; }
    grp.np = s2t_gemm_arith_of(2);
    grp.n = std::min(MAXG, n - base);
    long total_tiles = 0;
    for (int i = 0; i < grp.n; ++i) {
      const S2tTnProblem& s = probs[base + i];
      if (s.M <= 0 || s.N <= 0 || s.K <= 0 || s.lda > INT32_MAX || s.ldb > INT32_MAX || s.ldc > INT32_MAX)
        return -2;
      int tmm, tnn;
      tn_w_shape_of(s.M, s.N, tmm, tnn);
      total_tiles += (long)tmm * tnn;
    }
    int want = (int)((qtarget + total_tiles - 1) / total_tiles);
    want = std::max(8, ((want + 4) / 8) * 8);
    unsigned blocks = 0;
    int lds = TNW_LDS0;
    for (int i = 0; i < grp.n; ++i) {
      const S2tTnProblem& s = probs[base + i];
      TnProb& q = grp.p[i];
      q = TnProb{s.A, s.B, s.C, s.colsum, (int)s.lda, (int)s.ldb, (int)s.ldc, s.M, s.N, s.K,
                 0, 0, 0, 0, s.alpha, 0};
      q.shape = tn_w_shape_of(s.M, s.N, q.tiles_m, q.tiles_n);
      if (q.shape) lds = TNW_LDS;
      const long tiles = (long)q.tiles_m * q.tiles_n;
      const int maxs = (s.K + 2 * KR - 1) / (2 * KR);
      int splits = std::max(1, std::min(want, maxs));
      int kper = (s.K + splits - 1) / splits;
      kper = ((kper + KR - 1) / KR) * KR;
      q.kper = kper;
      q.splits = (s.K + kper - 1) / kper;
      grp.begin[i] = blocks;
      blocks += (unsigned)(8 * tiles * ((q.splits + 7) / 8));
    }
    grp.begin[grp.n] = blocks;
    hipLaunchKernelGGL(gemm_tn_grouped_w_kernel, dim3(blocks), dim3(512), lds, st, grp);
    if (hipGetLastError() != hipSuccess) return -3;
  }
  if (useq) return 0;
  for (int base = 0; base < n; base += MAXG) {
    TnGroup grp;
    { static const int dbg = s2t_debug_env("S2T_GEMM_DEBUG"); grp.debug = dbg; }
    grp.np = s2t_gemm_arith_of(2);
    grp.n = std::min(MAXG, n - base);
    long total_tiles = 0;
    for (int i = 0; i < grp.n; ++i) {
      const S2tTnProblem& s = probs[base + i];
      if (s.M < 4 || s.N < 4 || s.K < 4 || (s.M & 3) || (s.N & 3) || (s.lda & 3) || (s.ldb & 3) ||
          (reinterpret_cast<uintptr_t>(s.A) & 15) || (reinterpret_cast<uintptr_t>(s.B) & 15) ||
          s.lda > INT32_MAX || s.ldb > INT32_MAX || s.ldc > INT32_MAX)
        return -2;
      total_tiles += (long)((s.M + 64 * tmw - 1) / (64 * tmw)) * ((s.N + 64 * tnw - 1) / (64 * tnw));
    }
    int want = (int)((gtarget + total_tiles - 1) / total_tiles);
    want = std::max(8, ((want + 4) / 8) * 8);
    unsigned blocks = 0;
    for (int i = 0; i < grp.n; ++i) {
      const S2tTnProblem& s = probs[base + i];
      TnProb& q = grp.p[i];
      q = TnProb{s.A, s.B, s.C, s.colsum, (int)s.lda, (int)s.ldb, (int)s.ldc, s.M, s.N, s.K,
                 0, (s.M + 64 * tmw - 1) / (64 * tmw), (s.N + 64 * tnw - 1) / (64 * tnw), 0, s.alpha};
      const long tiles = (long)q.tiles_m * q.tiles_n;
      const int maxs = (s.K + 2 * KR - 1) / (2 * KR);
      int splits = std::max(1, std::min(want, maxs));
      int kper = (s.K + splits - 1) / splits;
      kper = ((kper + KR - 1) / KR) * KR;
      q.kper = kper;
      q.splits = (s.K + kper - 1) / kper;
      grp.begin[i] = blocks;
      blocks += (unsigned)(8 * tiles * ((q.splits + 7) / 8));
    }
    grp.begin[grp.n] = blocks;
    static const bool p2 = [] { const char* e = getenv("S2T_TN_P2"); return !e || atoi(e) != 0; }();
    if (tn_x3() && tn_p3() && tmw == 2 && grp.np == 2 && p2)
      hipLaunchKernelGGL((gemm_tn_grouped_p2_kernel<2, 2>), dim3(blocks), dim3(256), 0, st, grp);
    else if (tn_x3() && tn_p3() && tmw == 2)
      hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, 2, true, 2>), dim3(blocks), dim3(256), 0, st, grp);
    else if (tn_x3() && tn_p3() && tnw == 2)
      hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, 2, true>), dim3(blocks), dim3(256), 0, st, grp);
    else if (tn_x3() && tn_p3())
      hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, 1, true>), dim3(blocks), dim3(256), 0, st, grp);
    else if (tn_x3() && tnw == 2)
      hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, 2>), dim3(blocks), dim3(256), 0, st, grp);
    else if (tn_x3())
      hipLaunchKernelGGL((gemm_tn_grouped_kernel<true, 1>), dim3(blocks), dim3(256), 0, st, grp);
    else
      hipLaunchKernelGGL((gemm_tn_grouped_kernel<false, 1>), dim3(blocks), dim3(256), 0, st, grp);
    if (hipGetLastError() != hipSuccess) return -3;
  }
  return 0;
}
