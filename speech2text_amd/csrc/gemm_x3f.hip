// fp32 GEMM on the bf16 matrix cores, pipelined form (see gemm_x3.hip for the arithmetic: exact
// three-way bf16 split of both operands, six piece products per term, fp32 accumulation).
//
//   NT:  C[M,N] = A[M,K] . B[N,K]^T (+ bias[N]) (+ beta R[M,N])
// A = fp32 activations, row-major.  B = a weight matrix (or its transpose, for the data gradient)
// whose bf16 pieces were written once per optimizer step by s2t_split_planes_frag in FRAGMENT-MAJOR
// order: for every (32-column tile nt, 16-deep k block kb, piece p) the 1 KB that ONE
// v_mfma_f32_32x32x16_bf16 B operand needs, lane-linear (lane l = column l & 31, k half l >> 5,
// 8 bf16 each).  A B fragment is then one coalesced 1 KB line group in HBM/L2, one LDS-DMA
// (global_load_lds_dwordx4, no VGPRs, no staging instructions), and one conflict-free
// ds_read_b128 -- the layout is free because the planes are ours.
//
// Workgroup = 4 waves (2 x 2), tile 128 x (64 TNW) x 16 per stage, two LDS stages:
//   stage = A tile 128 x 16 fp32 (8 KB; rows of 64 B, 16-byte chunks XOR-swizzled with
//           (row >> 2) & 3 on the SOURCE address so that b128 fragment reads are conflict-free)
//         + B pieces (2 TNW) x 3 x 1 KB.
// One barrier per stage: {wait for stage s, barrier, start the DMA of stage s+1 into the other
// buffer, multiply stage s}.  A is split into its bf16 pieces when a wave reads a fragment.
// Rings: A (streamed from HBM) 4 stages, B (L2-resident) 2 stages; the DMA is issued from inline
// asm and counted by hand (one `s_waitcnt vmcnt(2)` + one raw barrier per stage).  Two workgroups
// per CU (80 KB LDS, <= 256 registers).
//
// STATUS (round 3): verified (tests/test_gpu_gemm.py), NOT on the training step's path.  Measured
// on the C3 layer shapes (tools/bench_x3.py f): at hipBLASLt's fp32 speed, +-10 % (58 -> 54 us for
// 15872 x 256 x 768, 68 -> 58 us for 31744 x 192 x 512; small-M shapes lose).  What the ablations
// (tools/x3f_abl.py history in DESIGN.md section 3d) established:
//   * the MFMAs alone take 36 us of that 54 -- the six piece products run at 1.2-1.3 PFLOP/s, the
//     matrix cores' PRACTICAL bf16 rate on random data (half the nominal 2.5), so the method's
//     ceiling is ~210 TFLOP/s fp32-equivalent, 1.7-2x the fp32 library, not 2.67x;
//   * everything else (DMA waits, barriers, split, 49 MB of output stores) takes 33 us by itself
//     and the two do NOT overlap, with one or with two workgroups per CU, with the DMA one or
//     three stages ahead, with a per-workgroup rotation of the k order (L2 channel spread;
//     dropped again, it only cost summation-order accuracy): workgroups of a
//     1-2 round grid run in lockstep, so the store phase and the pipeline fill of all of them
//     coincide.  K = 192..960 gives 12..60 stages per tile: the fill and the 128 KB store are a
//     third of a tile's life.
// The remaining step is a persistent workgroup whose stores of tile t drain under the stages of
// tile t+1 (gfx950 counts stores in vmcnt too, so that needs its own accounting) and an 8-wave
// ping-pong so that split VALU and MFMA of the same SIMD interleave.
#include "common.h"
#include "../../include/s2t_mi355.h"
#include <cstdint>
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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

struct X3F {
  const float* A;
  long lda;
  const unsigned short* B;   // fragment-major pieces [NT][KB][3][64][8]
  float* C;
  long ldc;
  int M, N, K;
  const float* bias;
  const float* resid;
  long ldr;
  float beta;
  int tiles_m, tiles_n;
};

constexpr int BM = 128, A_BYTES = BM * 64;

// LDS-DMA issued from inline asm: hipcc then keeps no counter for it (it would drain vmcnt(0) before
// every LDS read that follows a DMA in program order); the kernel counts vmcnt by hand.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
               :
               : "v"(g), "s"(lds_addr)
               : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

template <int TNW>
__global__ __launch_bounds__(256, 2) void gemm_x3f_kernel(X3F g) {
  constexpr int NTB = 2 * TNW, B_BYTES = NTB * 3 * 1024;
  constexpr int NA = 4, NB = 2;          // A ring (HBM latency) 4 stages, B ring (L2) 2 stages
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int total = g.tiles_m * g.tiles_n;
  const int per_xcd = (total + 7) / 8;
  const int lin = (int)((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3));
  if (lin >= total) return;
  const int tm = lin / g.tiles_n, tn = lin % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * (64 * TNW);
  const int tid = threadIdx.x, lane = tid & 63, lo = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * 64, wt = (wave & 1) * TNW;   // wave's first row / first column tile
  const int NT = g.N >> 5, KB = g.K >> 4;
  const unsigned lds0 = lds_addr_of(smem);
  unsigned char* const sBbase = smem + NA * A_BYTES;

  // DMA sources.  A: the wave's two 16-row pieces of the tile; lane = (row r = lane >> 2, chunk
  // c' = lane & 3) fetches source chunk c' ^ ((r >> 2) & 3).
  const float* asrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int r = lane >> 2, row = (wave * 2 + q) * 16 + r;
    const int c = (lane & 3) ^ ((r >> 2) & 3);
    asrc[q] = g.A + (long)min(m0 + row, g.M - 1) * g.lda + 4 * c;
  }
  // B: pieces pb = j * 3 + p of this tile, wave w takes pb = w, w + 4, ... (past the last piece a
  // wave repeats it: same bytes to the same place, no branch)
  constexpr int NPB = (NTB * 3 + 3) / 4;
  const unsigned short* bsrc[NPB];
#pragma unroll
  for (int q = 0; q < NPB; ++q) {
    const int pb = min(wave + 4 * q, NTB * 3 - 1), j = pb / 3, p = pb % 3;
    const int nt = min((n0 >> 5) + j, NT - 1);
    bsrc[q] = g.B + (((long)nt * KB) * 3 + p) * 512 + lane * 8;
  }
  // stages past the end re-fetch the last one into a ring slot nobody reads any more: the DMA
  // count per iteration stays fixed, so one vmcnt value serves the whole loop
  auto kblock = [&](int s) { return min(s, KB - 1); };
  auto issue_a = [&](int s) {
    const int sc = kblock(s);
    const unsigned dst = lds0 + (s % NA) * A_BYTES + wave * 2048;
#pragma unroll
    for (int q = 0; q < 2; ++q) glds16(asrc[q] + 16 * sc, dst + q * 1024);
  };
  auto issue_b = [&](int s) {
    const int sc = kblock(s);
    const unsigned dst = lds0 + NA * A_BYTES + (s % NB) * B_BYTES;
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
      const int pb = min(wave + 4 * q, NTB * 3 - 1);
      glds16(bsrc[q] + (long)sc * 3 * 512, dst + pb * 1024);
    }
  };

  f32x16 acc[2][TNW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Pipeline: iteration s issues {B(s+1), A(s+3)} in that order, so at the top of iteration s the
  // only DMAs that may still be in flight are A(s+2)'s two: vmcnt(2) retires A(s), A(s+1), B(s).
  // One barrier per stage: after it every wave has finished reading stage s-1, whose slots
  // (B slot (s+1) % 2, A slot (s+3) % 4) this iteration's DMA overwrites.
  issue_a(0);
  issue_a(1);
  issue_b(0);
  issue_a(2);
  for (int s = 0; s < KB; ++s) {
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* sA = reinterpret_cast<const float*>(smem + (s % NA) * A_BYTES);
    const unsigned char* sB = sBbase + (s % NB) * B_BYTES;
    bf16x8 fa[2][3], fb[TNW][3];
    float4 av[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm + 32 * i + lo, sw = (row >> 2) & 3;
      av[i][0] = *reinterpret_cast<const float4*>(sA + row * 16 + 4 * ((2 * hi) ^ sw));
      av[i][1] = *reinterpret_cast<const float4*>(sA + row * 16 + 4 * ((2 * hi + 1) ^ sw));
    }
#pragma unroll
    for (int j = 0; j < TNW; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fb[j][p] = *reinterpret_cast<const bf16x8*>(sB + ((wt + j) * 3 + p) * 1024 + lane * 16);
    issue_b(s + 1);
    issue_a(s + 3);
    __builtin_amdgcn_sched_barrier(0);      // reads and DMA issue stay ahead of the math
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 v0 = av[i][0], v1 = av[i][1];
      uint4 q0, q1, q2;
      split_pair(v0.x, v0.y, q0.x, q1.x, q2.x);
      split_pair(v0.z, v0.w, q0.y, q1.y, q2.y);
      split_pair(v1.x, v1.y, q0.z, q1.z, q2.z);
      split_pair(v1.z, v1.w, q0.w, q1.w, q2.w);
      fa[i][0] = __builtin_bit_cast(bf16x8, q0);
      fa[i][1] = __builtin_bit_cast(bf16x8, q1);
      fa[i][2] = __builtin_bit_cast(bf16x8, q2);
    }
    // smallest terms first; consecutive MFMAs go to different accumulators
#define X3F_TERM(PA, PB)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < TNW; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0);
    X3F_TERM(2, 0) X3F_TERM(1, 1) X3F_TERM(0, 2) X3F_TERM(1, 0) X3F_TERM(0, 1) X3F_TERM(0, 0)
#undef X3F_TERM
  }
  // the ring's trailing (dummy) DMAs must have landed before this workgroup's LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // epilogue: lane holds column (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
      const int col = n0 + 32 * (wt + j) + lo;
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

// src: logical matrix Bm[n][k] (n < N, k < K) = transposed ? src[k * ld + n] : src[n * ld + k]
// -> dst[((nt * KB + kb) * 3 + p) * 512 + (hi * 32 + lo) * 8 + e], n = 32 nt + lo, k = 16 kb + 8 hi + e.
// Thread = one (n, kb, hi): 8 source elements, three 16-byte stores.
__global__ __launch_bounds__(256) void split_planes_frag_kernel(const float* __restrict__ src, long ld,
                                                                int N, int K, int transposed,
                                                                unsigned short* __restrict__ dst) {
  const int KB = K >> 4;
  const long n_items = (long)N * KB * 2;
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < n_items; it += (long)gridDim.x * 256) {
    // transposed sources are walked with n fastest (their contiguous direction)
    int n, kb, hi;
    if (transposed) {
      n = (int)(it % N);
      const long q = it / N;
      hi = (int)(q & 1);
      kb = (int)(q >> 1);
    } else {
      hi = (int)(it & 1);
      const long q = it >> 1;
      kb = (int)(q % KB);
      n = (int)(q / KB);
    }
    const int k0 = 16 * kb + 8 * hi;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = transposed ? src[(long)(k0 + e) * ld + n] : src[(long)n * ld + k0 + e];
    uint4 q0, q1, q2;
    split_pair(v[0], v[1], q0.x, q1.x, q2.x);
    split_pair(v[2], v[3], q0.y, q1.y, q2.y);
    split_pair(v[4], v[5], q0.z, q1.z, q2.z);
    split_pair(v[6], v[7], q0.w, q1.w, q2.w);
    const long o = ((((long)(n >> 5) * KB + kb) * 3) * 64 + hi * 32 + (n & 31)) * 8;
    *reinterpret_cast<uint4*>(dst + o) = q0;
    *reinterpret_cast<uint4*>(dst + o + 512) = q1;
    *reinterpret_cast<uint4*>(dst + o + 1024) = q2;
  }
}

template <int TNW>
void launch_x3f(const X3F& g, hipStream_t st) {
  constexpr int smem = 4 * A_BYTES + 2 * (2 * TNW * 3 * 1024);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_x3f_kernel<TNW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr = true;
  }
  const int total = g.tiles_m * g.tiles_n;
  hipLaunchKernelGGL(gemm_x3f_kernel<TNW>, dim3(((total + 7) / 8) * 8), dim3(256), smem, st, g);
}

}  // namespace

extern "C" {

long s2t_split_planes_frag_elems(int N, int K) { return 3L * N * K; }

int s2t_split_planes_frag(const float* src, long ld, int N, int K, int transposed,
                          unsigned short* dst, void* stream) {
  if (N <= 0 || K <= 0) return 0;
  if ((N & 31) || (K & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return -2;
  long grid = ((long)N * (K >> 4) * 2 + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(split_planes_frag_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                     src, ld, N, K, transposed, dst);
  S2T_CHECK_LAUNCH();
  return 0;
}

int s2t_gemm_x3f_nt(const float* A, long lda, const unsigned short* Bf, float* C, long ldc, int M,
                    int N, int K, const float* bias, const float* resid, long ldr, float beta,
                    int tnw, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return -1;
  if ((K & 15) || (N & 31) || (lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(Bf) & 15))
    return -2;
  if (tnw < 1 || tnw > 4) {
    // widest tile that does not leave a round of the chip mostly empty
    tnw = 2;
    double best = 1e30;
    for (int t = 4; t >= 1; --t) {
      const long tiles = (long)((M + BM - 1) / BM) * ((N + 64 * t - 1) / (64 * t));
      const double rounds = (double)((tiles + 511) / 512);
      const double cost = rounds * (t + 0.35);
      if (cost < best) { best = cost; tnw = t; }
    }
  }
  X3F g{A, lda, Bf, C, ldc, M, N, K, bias, resid, ldr, beta, (M + BM - 1) / BM,
        (N + 64 * tnw - 1) / (64 * tnw)};
  hipStream_t st = (hipStream_t)stream;
  switch (tnw) {
    case 1: launch_x3f<1>(g, st); break;
    case 2: launch_x3f<2>(g, st); break;
    case 3: launch_x3f<3>(g, st); break;
    default: launch_x3f<4>(g, st); break;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
