// fp32 GEMM on the bf16 matrix cores: every fp32 operand is split EXACTLY into three bf16 pieces
// (a = a0 + a1 + a2, 8 + 8 + 8 = 24 significant bits, round-to-nearest at each stage), and the
// product is evaluated as the six piece products of weight <= 2,
//     a b ~ a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped
// terms (a1 b2, a2 b1, a2 b2) are <= 2^-23 |a b| with random sign: the result carries fp32-level
// error (tests: error against fp64 within 2x of the exact-fp32 MFMA GEMM), while six bf16 MFMAs
// do the work of sixteen f32 MFMAs (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate; gfx950
// has no xf32 path) -- a 2.67x higher ceiling for the dense QKV / FFN products of the training
// step (reference: every nn.Linear of model/encoder/zipformer.py:1924-2695 and of the conformer
// block, model/encoder/conformer.py:170-178).
//
//   NT:  C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]) (+ beta R[M,N])
// A = activations, fp32 in HBM, split while they are staged (v_cvt_pk_bf16_f32, ~5 VALU ops per
// element); W = a weight matrix whose three bf16 planes were written once per optimizer step by
// s2t_split_planes (forward: planes of W; data gradient dx = g W: planes of W^T, so that it is an
// NT product too).  Workgroup = 4 waves (2 x 2), block tile 128 x 128 x 32, wave tile 64 x 64 =
// 2 x 2 MFMA tiles; LDS rows are padded to 40 bf16 (conflict-free ds_read_b128 fragments).
//
// STATUS (round 3): verified (tests/test_gpu_gemm.py: error against fp64 below the fp32 library's),
// NOT on the training step's path.  On the C3 / C2 layer shapes this straightforward kernel runs
// at hipBLASLt's fp32 speed (55 us for 15872 x 256 x 768, 110 TFLOP/s-equivalent), not above it:
// rocprofv3 shows the matrix pipe 32 % busy (SQ_VALU_MFMA_BUSY_CYCLES), and ablations show the
// three phases of a tile -- operand loads + LDS staging (25 us), MFMA (20 us, = the six-product
// floor), output stores (15 us) -- adding up instead of overlapping: with the math 2.67x
// cheaper, a one-chunk register prefetch no longer covers an HBM round trip, and K = 192..960
// gives a tile only 6..30 chunks to amortise its prologue and its 64 KB of output.  Deeper
// register prefetch cost occupancy (77 us), larger wave tiles ran one wave per SIMD with exposed
// latencies (99 us), staggering co-resident workgroups changed nothing.  What it needs is the
// guide's 8-wave ping-pong template: operands DMA'd into a multi-stage LDS ring
// (global_load_lds), persistent tiles with the epilogue of one tile under the main loop of the
// next -- a next-round item; the ceiling (six bf16 MFMAs per product) is 2.67x the f32 MFMA's.
#include "common.h"
#include "../../include/s2t_mi355.h"
#include <cstdint>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BK = 32, LD = 40;   // LD in bf16 elements (80-byte rows); BN = 64 TN

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

struct X3Args {
  const float* A;
  long lda;
  const unsigned short* W;   // planes [3][plane] of bf16 bits; row n at W + p * plane + n * ldw
  long ldw, plane;
  float* C;
  long ldc;
  int M, N, K;
  const float* bias;
  const float* resid;
  long ldr;
  float beta;
  int tiles_m, tiles_n;
};

constexpr int LDA = BK + 4;    // fp32 row of the A tile in LDS (144-byte rows, conflict-free b128)

template <int TN>   // MFMA tiles per wave along N: block tile 128 x (64 TN)
__global__ __launch_bounds__(256) void gemm_x3_nt_kernel(X3Args g) {
  constexpr int BN = 64 * TN;
  // A stays fp32 in LDS and is split into its bf16 pieces when a wave reads a fragment: the
  // split's VALU work then sits in the issue shadow of the wave's own MFMAs (8 of every 32
  // cycles), and the staging phase between the barriers is a plain copy.
  __shared__ __attribute__((aligned(16))) float sA[BM * LDA];
  __shared__ __attribute__((aligned(16))) unsigned short sB[3][BN * LD];
  // XCD-aware tile order (as gemm.hip): the blocks of one XCD walk a contiguous range of tiles
  const int total = g.tiles_m * g.tiles_n;
  const int per_xcd = (total + 7) / 8;
  const int lin = (int)((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3));
  if (lin >= total) return;
  const int tm = lin / g.tiles_n, tn = lin % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lo = lane & 31, hi = lane >> 5;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (32 * TN);


  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: A: 4 float4 per thread (row = idx / 8, k = 4 (idx % 8)); W: per plane 2 x 16 bytes
  // per thread (row = idx / 4, k = 8 (idx % 4)).  The next chunk's global loads are in flight
  // while the current one is multiplied.
  float4 ra[4];
  uint4 rb[3][TN];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, kk = k0 + 4 * (idx & 7);
      const int gr = min(m0 + row, g.M - 1), gk = min(kk, g.K - 4);
      ra[i] = *reinterpret_cast<const float4*>(g.A + (long)gr * g.lda + gk);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, kk = k0 + 8 * (idx & 3);
        const int gn = min(n0 + row, g.N - 1), gk = min(kk, g.K - 8);
        rb[p][i] = *reinterpret_cast<const uint4*>(g.W + (long)p * g.plane + (long)gn * g.ldw + gk);
      }
  };
  auto store = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, c4 = idx & 7;
      float4 v = ra[i];
      if (k0 + 4 * c4 >= g.K) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(&sA[row * LDA + 4 * c4]) = v;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int idx = tid + 256 * i, row = idx >> 2, c8 = idx & 3;
        uint4 v = rb[p][i];
        if (k0 + 8 * c8 >= g.K) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(&sB[p][row * LD + 8 * c8]) = v;
      }
  };
  auto compute = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[2][3], fb[TN][3];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[j][p] = *reinterpret_cast<const bf16x8*>(&sB[p][(wn + 32 * j + lo) * LD + 16 * ks + 8 * hi]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float* ap = &sA[(wm + 32 * i + lo) * LDA + 16 * ks + 8 * hi];
        const float4 v0 = *reinterpret_cast<const float4*>(ap);
        const float4 v1 = *reinterpret_cast<const float4*>(ap + 4);
        uint4 q0, q1, q2;
        split_pair(v0.x, v0.y, q0.x, q1.x, q2.x);
        split_pair(v0.z, v0.w, q0.y, q1.y, q2.y);
        split_pair(v1.x, v1.y, q0.z, q1.z, q2.z);
        split_pair(v1.z, v1.w, q0.w, q1.w, q2.w);
        fa[i][0] = __builtin_bit_cast(bf16x8, q0);
        fa[i][1] = __builtin_bit_cast(bf16x8, q1);
        fa[i][2] = __builtin_bit_cast(bf16x8, q2);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  };

  load(0);
  for (int k0 = 0; k0 < g.K; k0 += BK) {
    __syncthreads();                       // previous chunk fully consumed
    store(k0);
    __syncthreads();
    if (k0 + BK < g.K) load(k0 + BK);
    compute();
  }

  // epilogue: lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5).  The
  // residual tile is loaded as one batch from clamped addresses (all 16 loads of a tile in flight
  // together), validity only guards the stores.
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + lo;
      const int colc = min(col, g.N - 1);
      const float bv = g.bias ? g.bias[colc] : 0.f;
      const int rbase = m0 + wm + 32 * i + 4 * hi;
      float rv[16];
      if (g.resid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
          rv[r] = g.resid[(long)row * g.ldr + colc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] + bv;
        if (g.resid) v = fmaf(g.beta, rv[r], v);
        if (row < g.M && col < g.N) g.C[(long)row * g.ldc + col] = v;
      }
    }
}

// src[n] fp32 -> planes [3][n] bf16 (n % 4 == 0)
__global__ __launch_bounds__(256) void split_planes_kernel(const float4* __restrict__ src, long n4,
                                                           uint2* __restrict__ p0,
                                                           uint2* __restrict__ p1,
                                                           uint2* __restrict__ p2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = src[i];
    unsigned a0, a1, a2, b0, b1, b2;
    split_pair(v.x, v.y, a0, a1, a2);
    split_pair(v.z, v.w, b0, b1, b2);
    p0[i] = make_uint2(a0, b0);
    p1[i] = make_uint2(a1, b1);
    p2[i] = make_uint2(a2, b2);
  }
}

// planes of the TRANSPOSES of a table of row-major matrices that live in one flat fp32 buffer:
// matrix q = (off, R, C): dst[p][off + c * R + r] = piece_p(src[off + r * C + c]).  One workgroup
// per 32 x 32 tile (tile_begin = prefix sums), transposed through LDS.
struct MatTab {
  long off;
  int R, C;
  int tile_begin;
};
__global__ __launch_bounds__(256) void split_planes_t_kernel(const float* __restrict__ src,
                                                             const MatTab* __restrict__ tab,
                                                             int ntab, long plane,
                                                             unsigned short* __restrict__ dst) {
  __shared__ unsigned short t[3][32][34];
  int q = 0;
  {                                           // binary search of the tile's matrix
    int lo = 0, hi = ntab - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tab[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    q = lo;
  }
  const MatTab m = tab[q];
  const int tile = blockIdx.x - m.tile_begin, tc = (m.C + 31) / 32;
  const int r0 = (tile / tc) * 32, c0 = (tile % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = (r < m.R && c < m.C) ? src[m.off + (long)r * m.C + c] : 0.f;
    unsigned a0, a1, a2, d0, d1, d2;
    split_pair(v, 0.f, a0, a1, a2);
    (void)d0; (void)d1; (void)d2;
    t[0][ty + 8 * i][tx] = (unsigned short)(a0 & 0xFFFFu);
    t[1][ty + 8 * i][tx] = (unsigned short)(a1 & 0xFFFFu);
    t[2][ty + 8 * i][tx] = (unsigned short)(a2 & 0xFFFFu);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < m.C && r < m.R) {
      const long o = m.off + (long)c * m.R + r;
#pragma unroll
      for (int p = 0; p < 3; ++p) dst[(long)p * plane + o] = t[p][tx][ty + 8 * i];
    }
  }
}

}  // namespace

extern "C" {

int s2t_split_planes(const float* src, long n, unsigned short* planes, long plane, void* stream) {
  if (n <= 0) return 0;
  if ((n & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(planes) & 7) ||
      (plane & 3))
    return -2;
  long grid = (n / 4 + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), n / 4, reinterpret_cast<uint2*>(planes),
                     reinterpret_cast<uint2*>(planes + plane), reinterpret_cast<uint2*>(planes + 2 * plane));
  S2T_CHECK_LAUNCH();
  return 0;
}

// tab: DEVICE array of ntab {long off; int R, C; int tile_begin} (tile_begin = running sum of
// ceil(R/32) * ceil(C/32)); total_tiles = the sum.
int s2t_split_planes_t(const float* src, const void* tab, int ntab, int total_tiles,
                       unsigned short* planes_t, long plane, void* stream) {
  if (ntab <= 0 || total_tiles <= 0) return 0;
  hipLaunchKernelGGL(split_planes_t_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, src,
                     reinterpret_cast<const MatTab*>(tab), ntab, plane, planes_t);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_gemm_x3_nt(const float* A, long lda, const unsigned short* W, long ldw, long plane,
                   float* C, long ldc, int M, int N, int K, const float* bias, const float* resid,
                   long ldr, float beta, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  if ((K & 7) || (lda & 3) || (ldw & 7) || (plane & 7) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(W) & 15))
    return -2;
  static int tn_force = -1;
  if (tn_force < 0) { const char* e = getenv("S2T_X3_TN"); tn_force = e ? atoi(e) : 0; }
  const int TN = tn_force ? tn_force : 2;
  const int BN = 64 * TN;
  X3Args g{A, lda, W, ldw, plane, C, ldc, M, N, K, bias, resid, ldr, beta,
           (M + BM - 1) / BM, (N + BN - 1) / BN};
  const int total = g.tiles_m * g.tiles_n;
  if (TN == 1)
    hipLaunchKernelGGL(gemm_x3_nt_kernel<1>, dim3(((total + 7) / 8) * 8), dim3(256), 0,
                       (hipStream_t)stream, g);
  else
    hipLaunchKernelGGL(gemm_x3_nt_kernel<2>, dim3(((total + 7) / 8) * 8), dim3(256), 0,
                       (hipStream_t)stream, g);
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
