// Channel-last (N,H,W,C) depthwise 2-D convolution for the zipformer frontend (ConvNeXt 7x7,
// model/layer/subsampling.py:47-53,121) on gfx950.  MIOpen's fp32 depthwise path falls back to
// naive kernels here (2.5 ms fwd / 4.6 ms bwd per step at the benchmark shape); in NHWC the
// op is a pure streaming stencil: lanes = channels (coalesced), each thread slides over W.
//   y[n,h,w,c] = bias[c] + sum_{i,j} wgt[c,i,j] * x[n,h+i-ph,w+j-pw,c]     (zero padding)
// backward data = the same stencil with the taps flipped; backward weight = per-block
// partial sums [block][c][KH*KW+1] reduced by a second kernel (no contended atomics).
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int TH = 4;   // output rows per workgroup
constexpr int WT = 8;   // consecutive output columns per thread (register window)

// grid: (ceil(H/TH), N, ceil(C/64)); block 256 = 64 channels x 4 row-threads.  Each thread
// walks its row in strips of WT outputs: the (KH x (WT+KW-1)) input window is loaded once into
// registers and feeds WT*KH*KW FMAs (KH*(WT+KW-1)/WT loads per output instead of KH*KW).
template <int KH, int KW, bool FLIP>
__global__ __launch_bounds__(256) void dwconv2d_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ wgt,
                                                       const float* __restrict__ bias, int N,
                                                       int H, int W, int C,
                                                       float* __restrict__ y) {
  constexpr int PH = KH / 2, PW = KW / 2, WW = WT + KW - 1;
  __shared__ float s_w[KH * KW][64];
  const int c0 = blockIdx.z * 64, c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int n = blockIdx.y, h = blockIdx.x * TH + r;
  for (int i = threadIdx.x; i < KH * KW * 64; i += 256) {
    const int cc = i / (KH * KW), k = i % (KH * KW);
    const int kk = FLIP ? (KH * KW - 1 - k) : k;
    s_w[k][cc] = (c0 + cc < C) ? wgt[(long)(c0 + cc) * KH * KW + kk] : 0.f;
  }
  __syncthreads();
  if (h >= H || c0 + c >= C) return;
  const float b = bias ? bias[c0 + c] : 0.f;
  const float* xn = x + (long)n * H * W * C + c0 + c;
  float* yn = y + ((long)n * H + h) * W * C + c0 + c;
  for (int w0 = 0; w0 < W; w0 += WT) {
    float acc[WT];
#pragma unroll
    for (int q = 0; q < WT; ++q) acc[q] = b;
#pragma unroll
    for (int i = 0; i < KH; ++i) {
      const int hh = h + i - PH;
      if (hh < 0 || hh >= H) continue;
      float win[WW];
#pragma unroll
      for (int v = 0; v < WW; ++v) {
        const int ww = w0 + v - PW;
        win[v] = (ww >= 0 && ww < W) ? xn[((long)hh * W + ww) * C] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < KW; ++j) {
        const float wv = s_w[i * KW + j][c];
#pragma unroll
        for (int q = 0; q < WT; ++q) acc[q] = fmaf(wv, win[q + j], acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < WT; ++q)
      if (w0 + q < W) yn[(long)(w0 + q) * C] = acc[q];
  }
}

// partial weight gradients: part[blk][c][KH*KW + 1] (last = bias)
template <int KH, int KW>
__global__ __launch_bounds__(256) void dwconv2d_wgrad_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ dy, int N,
                                                             int H, int W, int C, int rows_per_blk,
                                                             float* __restrict__ part) {
  constexpr int PH = KH / 2, PW = KW / 2, NV = KH * KW + 1, WW = WT + KW - 1;
  __shared__ float s_red[4][64];
  const int c0 = blockIdx.z * 64, c = threadIdx.x & 63, r = threadIdx.x >> 6;
  const int n = blockIdx.y;
  const bool ok = c0 + c < C;
  float acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.f;
  const int h0 = blockIdx.x * rows_per_blk;
  if (ok) {
    const float* xn = x + (long)n * H * W * C + c0 + c;
    const float* gn = dy + (long)n * H * W * C + c0 + c;
    for (int h = h0 + r; h < min(H, h0 + rows_per_blk); h += 4) {
      for (int w0 = 0; w0 < W; w0 += WT) {
        float g[WT];
#pragma unroll
        for (int q = 0; q < WT; ++q) {
          g[q] = (w0 + q < W) ? gn[((long)h * W + w0 + q) * C] : 0.f;
          acc[NV - 1] += g[q];
        }
#pragma unroll
        for (int i = 0; i < KH; ++i) {
          const int hh = h + i - PH;
          if (hh < 0 || hh >= H) continue;
          float win[WW];
#pragma unroll
          for (int v = 0; v < WW; ++v) {
            const int ww = w0 + v - PW;
            win[v] = (ww >= 0 && ww < W) ? xn[((long)hh * W + ww) * C] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < KW; ++j)
#pragma unroll
            for (int q = 0; q < WT; ++q) acc[i * KW + j] = fmaf(g[q], win[q + j], acc[i * KW + j]);
        }
      }
    }
  }
  // part[c-tile][block][slot][64 channels]: lanes = channels, so every store is one 256-byte row
  const long blk = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float* dst = part + blk * NV * 64 + c;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    __syncthreads();
    s_red[r][c] = acc[k];
    __syncthreads();
    if (r == 0) dst[k * 64] = s_red[0][c] + s_red[1][c] + s_red[2][c] + s_red[3][c];
  }
}

// Rolling-window form of the stencil (the one the 7x7 ConvNeXt conv runs on): a thread owns one
// channel, WO = 7 consecutive output columns and RT consecutive rows.  It keeps the KH x (WO+KW-1)
// input window in REGISTERS (a ring of KH + RPD rows), and moving down one output row costs one
// new input row (13 coalesced loads) instead of refetching KH rows through the cache; that row is
// requested RPD steps before it is used, so ~4 rows of loads are in flight per wave -- with one
// row in flight the kernel ran at the latency-bound 620 us of its predecessors (bytes in flight,
// not FMAs or bandwidth, were the limit: 8 waves/CU x 13 x 256 B against ~2 us of latency).
//   MODE 0: y = conv(x)   MODE 1: taps flipped (= backward data)
//   MODE 2: weight / bias gradient partials of (x, dy), part[c-tile][block][tap | bias][64]
__device__ float g_zero[256];   // zero-initialised: the padding taps of the stencil read it

constexpr int RWO = 7;    // output columns per thread
constexpr int RPD = 3;    // input rows requested ahead of use (memory latency / time per row step)
constexpr int RRT = 30;   // output rows per thread (multiple of the ring length KH + RPD)
constexpr int RCT = 128;  // channels per workgroup

template <int KH, int KW, int MODE>
__global__ __launch_bounds__(256) void dwconv2d_roll_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ wgt,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ dy, int N,
                                                            int H, int W, int C, int nparts,
                                                            int nrr, float* __restrict__ out) {
  constexpr int RING = KH + RPD;
  static_assert(RRT % RING == 0, "row range must be a multiple of the ring length");
  constexpr int PH = KH / 2, PW = KW / 2, WW = RWO + KW - 1, NV = KH * KW + 1;
  // RCT channels per workgroup: sibling waves read the neighbouring 256-byte pieces of the same
  // pixels at the same time
  const int c0 = blockIdx.z * RCT, c = threadIdx.x % RCT, n = blockIdx.y;
  const int q = blockIdx.x * (256 / RCT) + (threadIdx.x / RCT);
  const int part = q % nparts, rr = q / nparts;
  const bool live = c0 + c < C && rr < nrr;
  const int cc = (c0 + c < C) ? c0 + c : 0;
  const int w0 = part * RWO, hs = rr * RRT, he = min(H, hs + RRT);
  // MODE 0/1: the taps live in LDS, [tap][channel] (conflict-free, read per FMA group): as 49
  // registers per lane next to the 130-register window they spilled to scratch, and a scratch
  // reload shares vmcnt with the prefetched rows -- every reload waited for the rows in flight,
  // which serialised the very loads the ring exists to overlap (600 us for a 120 us stream).
  // MODE 2: wv are the tap-gradient accumulators (registers).
  __shared__ float s_w[MODE == 2 ? 1 : KH * KW][MODE == 2 ? 1 : RCT];
  float wv[MODE == 2 ? KH * KW : 1];
  float bacc = 0.f;
  if (MODE == 2) {
#pragma unroll
    for (int k = 0; k < (MODE == 2 ? KH * KW : 1); ++k) wv[k] = 0.f;
  } else {
    for (int k = threadIdx.x / RCT; k < KH * KW; k += 256 / RCT)
      s_w[MODE == 2 ? 0 : k][c] = wgt[(long)cc * KH * KW + (MODE == 1 ? KH * KW - 1 - k : k)];
    __syncthreads();
  }
  const float bv = (MODE != 2 && bias) ? bias[cc] : 0.f;
  const float* xn = x + (long)n * H * W * C;
  float win[RING][WW];     // ring of input rows: slot (k + i) % RING holds row hs - PH + k + i
  // Out-of-range taps read a zero word instead of being patched after the load: the address
  // choice is wave-uniform (scalar), and nothing has to wait for the loaded value until the FMAs
  // that use it, RPD steps later.
  // (a macro, not a lambda taking the row by reference: through the reference the window array
  // stayed an alloca in scratch memory, whose loads and stores share vmcnt with the prefetched
  // rows and waited for them)
#define S2T_LOAD_ROW(SLOT, HIN)                                                        \
  {                                                                                    \
    const int hin_ = (HIN);                                                            \
    const bool rv_ = hin_ >= 0 && hin_ < H;                                            \
    const float* row_ = xn + (long)hin_ * W * C;                                       \
    _Pragma("unroll") for (int v = 0; v < WW; ++v) {                                   \
      const int ww_ = w0 + v - PW;                                                     \
      const float* p_ = (rv_ && ww_ >= 0 && ww_ < W) ? row_ + (long)ww_ * C : g_zero;  \
      win[SLOT][v] = p_[cc];                                                           \
    }                                                                                  \
  }
  if (live) {
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) S2T_LOAD_ROW(s, hs - PH + s)
    for (int k0 = 0; hs + k0 < he; k0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int h = hs + k0 + u;
        if (h < he) {                                   // uniform per wave
          // request the row that is needed RPD steps from now into the slot the oldest row left:
          // RPD + 1 rows of loads are in flight while a step's FMAs run
          S2T_LOAD_ROW((u + RING - 1) % RING, h + PH + RPD)
          if (MODE != 2) {
            // an index the compiler cannot see through (always 0): the 49 tap reads stay inside this
            // row step -- hoisted out of the row loop they are 49 live registers again
            int opaque0;
            asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
            float acc[RWO];
            if (dy) {          // MODE 0/1: dy is an optional addend with the output's layout
              const float* ar = dy + (((long)n * H + h) * W + w0) * C + cc;
#pragma unroll
              for (int o = 0; o < RWO; ++o)
                acc[o] = bv + ((w0 + o < W) ? ar[(long)min(o, W - 1 - w0) * C] : 0.f);
            } else {
#pragma unroll
              for (int o = 0; o < RWO; ++o) acc[o] = bv;
            }
#pragma unroll
            for (int i = 0; i < KH; ++i)
#pragma unroll
              for (int j = 0; j < KW; ++j) {
                const float tap = s_w[MODE == 2 ? 0 : i * KW + j][c + opaque0];
#pragma unroll
                for (int o = 0; o < RWO; ++o)
                  acc[o] = fmaf(tap, win[(u + i) % RING][o + j], acc[o]);
              }
            float* yr = out + (((long)n * H + h) * W + w0) * C + cc;
#pragma unroll
            for (int o = 0; o < RWO; ++o)
              if (w0 + o < W) yr[(long)o * C] = acc[o];
          } else {
            float g[RWO];
            const float* gr = dy + (((long)n * H + h) * W + w0) * C + cc;
#pragma unroll
            for (int o = 0; o < RWO; ++o) {
              g[o] = (w0 + o < W) ? gr[(long)min(o, W - 1 - w0) * C] : 0.f;
              bacc += g[o];
            }
#pragma unroll
            for (int i = 0; i < KH; ++i)
#pragma unroll
              for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int o = 0; o < RWO; ++o)
                  wv[MODE == 2 ? i * KW + j : 0] = fmaf(g[o], win[(u + i) % RING][o + j], wv[MODE == 2 ? i * KW + j : 0]);
          }
        }
      }
    }
  }
#undef S2T_LOAD_ROW
  if (MODE == 2) {
    // partial layout [64-channel tile][block][slot][64]; a workgroup covers RCT / 64 tiles
    __shared__ float s_red[256 / RCT][RCT];
    const int r = threadIdx.x / RCT;
    const long nblk = (long)gridDim.y * gridDim.x;
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    const long tile = (long)blockIdx.z * (RCT / 64) + c / 64;
    float* dst = out + ((tile * nblk + blk) * NV) * 64 + (c & 63);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      __syncthreads();
      s_red[r][c] = live ? (k < NV - 1 ? wv[MODE == 2 ? (k < NV - 1 ? k : 0) : 0] : bacc) : 0.f;
      __syncthreads();
      if (r == 0) {
        float t = 0.f;
#pragma unroll
        for (int rr2 = 0; rr2 < 256 / RCT; ++rr2) t += s_red[rr2][c];
        dst[k * 64] = t;
      }
    }
  }
}

// Weight / bias gradient of the 7x7 depthwise conv, ring form (round 5).  The roll kernel's MODE 2
// kept the KH-row INPUT window (130 registers) next to the 49 tap accumulators: 256 VGPRs plus
// AGPR spill space, one wave per SIMD, 1.08 ms for a 0.6 GB stream.  Here the thread keeps a ring of
// the last KH rows of dy (7 values each) and streams the input rows through once: input row
// r = hs - PH + t meets dy row hh = t - i under tap row i.  49 accumulators + 9 x 7 dy + 3 x 13 input
// registers (GPD = 2 rows requested ahead of use) -> 3 waves per SIMD.
//   part[c-tile][block][tap | bias][64], as the roll kernel's MODE 2.
template <int KH, int KW, int GPD>
__global__ __launch_bounds__(256) void dwconv2d_wgrad_ring_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ dy, int N,
                                                                  int H, int W, int C, int nparts,
                                                                  int nrr, int rrt,
                                                                  float* __restrict__ out) {
  constexpr int GR = KH + GPD, XR = 1 + GPD;
  static_assert(GR % XR == 0, "the unrolled step loop must keep both ring slots compile-time");
  constexpr int PH = KH / 2, PW = KW / 2, WW = RWO + KW - 1, NV = KH * KW + 1;
  const int c0 = blockIdx.z * RCT, c = threadIdx.x % RCT, n = blockIdx.y;
  const int q = blockIdx.x * (256 / RCT) + (threadIdx.x / RCT);
  const int part = q % nparts, rr = q / nparts;
  const bool live = c0 + c < C && rr < nrr;
  const int cc = (c0 + c < C) ? c0 + c : 0;
  const int w0 = part * RWO, hs = rr * rrt, nrows = min(H, hs + rrt) - hs;
  float wv[KH * KW];
  float bacc = 0.f;
#pragma unroll
  for (int k = 0; k < KH * KW; ++k) wv[k] = 0.f;
  const float* xn = x + (long)n * H * W * C;
  const float* gn = dy + (long)n * H * W * C;
  float gd[GR][RWO];       // dy rows: slot hh % GR
  float xw[XR][WW];        // input rows: slot t % XR
#define S2T_LOADX(SLOT, T)                                                             \
  {                                                                                    \
    const int r_ = hs - PH + (T);                                                      \
    const bool rv_ = r_ >= 0 && r_ < H;                                                \
    const float* row_ = xn + (long)r_ * W * C;                                         \
    _Pragma("unroll") for (int v = 0; v < WW; ++v) {                                   \
      const int ww_ = w0 + v - PW;                                                     \
      const float* p_ = (rv_ && ww_ >= 0 && ww_ < W) ? row_ + (long)ww_ * C : g_zero;  \
      xw[SLOT][v] = p_[cc];                                                            \
    }                                                                                  \
  }
#define S2T_LOADG(SLOT, HH)                                                            \
  {                                                                                    \
    const int hh_ = (HH);                                                              \
    const bool rv_ = hh_ < nrows;                                                      \
    const float* row_ = gn + ((long)(hs + hh_) * W + w0) * C;                          \
    _Pragma("unroll") for (int o = 0; o < RWO; ++o) {                                  \
      const float* p_ = (rv_ && w0 + o < W) ? row_ + (long)o * C : g_zero;             \
      gd[SLOT][o] = p_[cc];                                                            \
    }                                                                                  \
  }
  if (live && nrows > 0) {
#pragma unroll
    for (int s = 0; s < GPD; ++s) {
      S2T_LOADX(s % XR, s)
      S2T_LOADG(s % GR, s)
    }
    const int nsteps = nrows + KH - 1;
    for (int k0 = 0; k0 < nsteps; k0 += GR) {
#pragma unroll
      for (int u = 0; u < GR; ++u) {
        const int t = k0 + u;
        if (t < nsteps) {                              // uniform per wave
          S2T_LOADX((u + GPD) % XR, t + GPD)
          S2T_LOADG((u + GPD) % GR, t + GPD)
          if (t < nrows) {
#pragma unroll
            for (int o = 0; o < RWO; ++o) bacc += gd[u][o];
          }
#pragma unroll
          for (int i = 0; i < KH; ++i)
            if ((unsigned)(t - i) < (unsigned)nrows) {
#pragma unroll
              for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int o = 0; o < RWO; ++o)
                  wv[i * KW + j] = fmaf(gd[(u - i + GR) % GR][o], xw[u % XR][o + j], wv[i * KW + j]);
            }
        }
      }
    }
  }
#undef S2T_LOADX
#undef S2T_LOADG
  __shared__ float s_red[256 / RCT][RCT];
  const int r = threadIdx.x / RCT;
  const long nblk = (long)gridDim.y * gridDim.x;
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  const long tile = (long)blockIdx.z * (RCT / 64) + c / 64;
  float* dst = out + ((tile * nblk + blk) * NV) * 64 + (c & 63);
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    __syncthreads();
    s_red[r][c] = live ? (k < NV - 1 ? wv[k < NV - 1 ? k : 0] : bacc) : 0.f;
    __syncthreads();
    if (r == 0) {
      float t = 0.f;
#pragma unroll
      for (int rr2 = 0; rr2 < 256 / RCT; ++rr2) t += s_red[rr2][c];
      dst[k * 64] = t;
    }
  }
}

__global__ __launch_bounds__(1024) void dwconv2d_wreduce_kernel(const float* __restrict__ part,
                                                                int nblk, int C, int NV,
                                                                float* __restrict__ dw,
                                                                float* __restrict__ db) {
  // grid (c tiles, NV slots); thread = (channel, block group): coalesced reads of 64 channels.
  // 16 block groups with 8 independent loads in flight each: the 100 workgroups of the 7x7 case
  // stream the ~40 MB of partials at HBM speed (4 groups with a dependent chain took 153 us)
  __shared__ float s_red[16][64];
  const int ct = blockIdx.x, slot = blockIdx.y;
  const int c = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const float* p = part + ((long)ct * nblk * NV + slot) * 64 + c;
  const long step = (long)NV * 64;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = kg;
  for (; k + 48 < nblk; k += 64) {
    s0 += p[k * step];
    s1 += p[(k + 16) * step];
    s2 += p[(k + 32) * step];
    s3 += p[(k + 48) * step];
  }
  for (; k < nblk; k += 16) s0 += p[k * step];
  s_red[kg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  const int cg = ct * 64 + c;
  if (kg == 0 && cg < C) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += s_red[i][c];
    if (slot < NV - 1) dw[(long)cg * (NV - 1) + slot] = s;
    else if (db) db[cg] = s;
  }
}

}  // namespace

// x,y (N,H,W,C) channel-last; wgt (C,KH,KW); only 7x7 (ConvNeXt) and 3x3 are instantiated.
// add (optional, 7x7 only): y = conv(x) + add -- the residual branch's gradient riding in the
// backward-data pass.
extern "C" int s2t_dwconv2d_nhwc_fwd_add(const float* x, const float* wgt, const float* bias,
                                         const float* add, int N, int H, int W, int C, int KH,
                                         int KW, int flip, float* y, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  if (add && !(KH == 7 && KW == 7)) return -2;
  dim3 grid((H + TH - 1) / TH, N, (C + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (KH == 7 && KW == 7) {
    const int nparts = (W + RWO - 1) / RWO, nrr = (H + RRT - 1) / RRT;
    constexpr int CPB = 256 / RCT;
    dim3 gr((nparts * nrr + CPB - 1) / CPB, N, (C + RCT - 1) / RCT);
    if (flip) hipLaunchKernelGGL((dwconv2d_roll_kernel<7, 7, 1>), gr, dim3(256), 0, st, x, wgt, bias, add, N, H, W, C, nparts, nrr, y);
    else hipLaunchKernelGGL((dwconv2d_roll_kernel<7, 7, 0>), gr, dim3(256), 0, st, x, wgt, bias, add, N, H, W, C, nparts, nrr, y);
  } else if (KH == 3 && KW == 3) {
    if (flip) hipLaunchKernelGGL((dwconv2d_kernel<3, 3, true>), grid, dim3(256), 0, st, x, wgt, bias, N, H, W, C, y);
    else hipLaunchKernelGGL((dwconv2d_kernel<3, 3, false>), grid, dim3(256), 0, st, x, wgt, bias, N, H, W, C, y);
  } else {
    return -1;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_dwconv2d_nhwc_fwd(const float* x, const float* wgt, const float* bias, int N,
                                     int H, int W, int C, int KH, int KW, int flip, float* y,
                                     void* stream) {
  return s2t_dwconv2d_nhwc_fwd_add(x, wgt, bias, nullptr, N, H, W, C, KH, KW, flip, y, stream);
}

extern "C" long s2t_dwconv2d_wgrad_workspace_floats(int N, int H, int C, int KH, int KW) {
  const long hb = (H + 7) / 8;
  return hb * N * ((C + 63) / 64) * 64 * (KH * KW + 1);
}

extern "C" int s2t_dwconv2d_nhwc_wgrad(const float* x, const float* dy, int N, int H, int W, int C,
                                       int KH, int KW, float* workspace, float* dw, float* db,
                                       void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nparts = (W + RWO - 1) / RWO, nrr = (H + RRT - 1) / RRT;
  // (the workspace is sized for (H + 7) / 8 partial blocks per (n, c-tile): holds for W <= 98)
  constexpr int CPB = 256 / RCT;
  static int ring = -1;      // S2T_DWCONV_WGRAD_RING=0: the roll kernel's MODE 2
  if (ring < 0) { const char* e = getenv("S2T_DWCONV_WGRAD_RING"); ring = e ? atoi(e) : 1; }
  if (KH == 7 && KW == 7 && ring) {
    // rows per thread: the 2 workgroups per CU that fit (210 VGPRs) in ONE round where the map is
    // tall enough (each thread pays KH - 1 extra row steps, so not below 24 rows)
    const long per_row = (long)nparts * N * ((C + RCT - 1) / RCT);   // thread groups per row range
    const int nr = std::max(1, (int)((512L * CPB) / per_row));
    const int rrt = std::max(24, (H + nr - 1) / nr);
    const int nrr2 = (H + rrt - 1) / rrt;
    if ((nparts * nrr2 + CPB - 1) / CPB <= (H + 7) / 8) {
      dim3 gr((nparts * nrr2 + CPB - 1) / CPB, N, (C + RCT - 1) / RCT);
      hipLaunchKernelGGL((dwconv2d_wgrad_ring_kernel<7, 7, 2>), gr, dim3(256), 0, st, x, dy, N, H, W, C,
                         nparts, nrr2, rrt, workspace);
      S2T_CHECK_LAUNCH();
      hipLaunchKernelGGL(dwconv2d_wreduce_kernel, dim3((C + 63) / 64, KH * KW + 1), dim3(1024), 0, st,
                         workspace, (int)(gr.x * gr.y), C, KH * KW + 1, dw, db);
      S2T_CHECK_LAUNCH();
      return 0;
    }
  }
  if (KH == 7 && KW == 7 && (nparts * nrr + CPB - 1) / CPB <= (H + 7) / 8) {
    dim3 gr((nparts * nrr + CPB - 1) / CPB, N, (C + RCT - 1) / RCT);
    const float* nf = nullptr;
    hipLaunchKernelGGL((dwconv2d_roll_kernel<7, 7, 2>), gr, dim3(256), 0, st, x, nf, nf, dy, N, H, W,
                       C, nparts, nrr, workspace);
    S2T_CHECK_LAUNCH();
    hipLaunchKernelGGL(dwconv2d_wreduce_kernel, dim3((C + 63) / 64, KH * KW + 1), dim3(1024), 0, st,
                       workspace, (int)(gr.x * gr.y), C, KH * KW + 1, dw, db);
    S2T_CHECK_LAUNCH();
    return 0;
  }
  // ~1000 workgroups: enough to fill the chip, few enough that the partial sums stay small
  int rows_per_blk = (int)(((long)H * N * ((C + 63) / 64) + 1023) / 1024);
  rows_per_blk = rows_per_blk < 8 ? 8 : ((rows_per_blk + 3) / 4) * 4;
  dim3 grid((H + rows_per_blk - 1) / rows_per_blk, N, (C + 63) / 64);
  if (KH == 7 && KW == 7)
    hipLaunchKernelGGL((dwconv2d_wgrad_kernel<7, 7>), grid, dim3(256), 0, st, x, dy, N, H, W, C, rows_per_blk, workspace);
  else if (KH == 3 && KW == 3)
    hipLaunchKernelGGL((dwconv2d_wgrad_kernel<3, 3>), grid, dim3(256), 0, st, x, dy, N, H, W, C, rows_per_blk, workspace);
  else
    return -1;
  S2T_CHECK_LAUNCH();
  hipLaunchKernelGGL(dwconv2d_wreduce_kernel, dim3((C + 63) / 64, KH * KW + 1), dim3(1024), 0, st,
                     workspace, (int)(grid.x * grid.y), C, KH * KW + 1, dw, db);
  S2T_CHECK_LAUNCH();
  return 0;
}

namespace {

// col2im of a 3x3 convolution on channel-last data, as a GATHER: every input position adds up the
// (at most 9) patch entries that cover it -- no zero fill, no strided slice-add passes.
// dc (B,Ho,Wo,3,3,C) -> dx (B,H,W,C)
__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dc, int B, int H,
                                                        int W, int C, int Ho, int Wo, int sh,
                                                        int sw, float* __restrict__ dx) {
  // one workgroup per (h, b) row of dx: the row part of the patch arithmetic is wave-uniform,
  // the rest is 32-bit (64-bit divisions per element were what this pass spent its time on)
  const int h = blockIdx.x, b = blockIdx.y;
  int ho[3];
  bool hv[3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hh = h - kh;
    hv[kh] = hh >= 0 && hh % sh == 0 && hh / sh < Ho;
    ho[kh] = hv[kh] ? hh / sh : 0;
  }
  float* row = dx + ((long)b * H + h) * W * C;
  const int n = W * C;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int w = i / C, c = i - w * C;
    // all nine taps are loaded from clamped (valid) addresses before any is used; taps that do
    // not exist for this position are masked out afterwards (loads behind `continue`s serialise)
    float t[9];
    bool ok[9];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ww = w - kw;
      const bool okw = ww >= 0 && ww % sw == 0 && ww / sw < Wo;
      const int wo = okw ? ww / sw : 0;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        ok[kh * 3 + kw] = okw && hv[kh];
        t[kh * 3 + kw] = dc[((((long)b * Ho + ho[kh]) * Wo + wo) * 9 + kh * 3 + kw) * C + c];
      }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc += ok[k] ? t[k] : 0.f;
    row[i] = acc;
  }
}

}  // namespace

extern "C" int s2t_col2im3x3_nhwc(const float* dc, int B, int H, int W, int C, int Ho, int Wo,
                                  int sh, int sw, float* dx, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  if (sh <= 0 || sw <= 0 || Ho != (H - 3) / sh + 1 || Wo != (W - 3) / sw + 1) return -1;
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(H, B), dim3(256), 0, (hipStream_t)stream, dc, B, H, W,
                     C, Ho, Wo, sh, sw, dx);
  S2T_CHECK_LAUNCH();
  return 0;
}

namespace {

// ---------------------------------------------------------------- first subsampling conv, direct
// Conv2d(1, CO, 3, padding=(0, pw)) on channel-last data (model/layer/subsampling.py:184-229): with
// one input channel the im2col GEMM is a 12-wide, 8-tall "matrix product" that the GEMM library
// runs at ~0.1 of HBM; as a stencil it is 9 loads, 9*CO FMAs and one CO-float row per output.
// x (B,H,W), y / g (B,Ho,Wo,CO) with Ho = H - 2, Wo = W + 2 pw - 2; w (CO,1,3,3), bias (CO).
template <int CO>
__global__ __launch_bounds__(256) void conv3x3_c1_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B,
                                                             int H, int W, int pw, int Ho, int Wo,
                                                             float* __restrict__ y) {
  float wv[CO * 9], bv[CO];
#pragma unroll
  for (int k = 0; k < CO * 9; ++k) wv[k] = w[k];
#pragma unroll
  for (int c = 0; c < CO; ++c) bv[c] = bias ? bias[c] : 0.f;
  const long n = (long)B * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int wo = (int)(i % Wo);
    const long r = i / Wo;
    const int ho = (int)(r % Ho), b = (int)(r / Ho);
    float xv[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = wo + kw - pw;
        const float t = x[((long)b * H + ho + kh) * W + min(max(ww, 0), W - 1)];
        xv[kh * 3 + kw] = (ww >= 0 && ww < W) ? t : 0.f;
      }
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      acc[c] = bv[c];
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[c] = fmaf(wv[c * 9 + k], xv[k], acc[c]);
    }
    float4* o = reinterpret_cast<float4*>(y + i * CO);
#pragma unroll
    for (int c = 0; c < CO; c += 4) o[c / 4] = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  }
}

// dw[co][k] += sum g[.., co] * x[tap k], db[co] += sum g: per-thread accumulators over a
// grid-stride range of output positions, wave + workgroup reduction, one atomic per value per
// workgroup (caller zeroes dw / db)
template <int CO>
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ g, int B,
                                                               int H, int W, int pw, int Ho, int Wo,
                                                               float* __restrict__ dw,
                                                               float* __restrict__ db) {
  __shared__ float s_red[4][CO * 10];
  float acc[CO * 9], bacc[CO];
#pragma unroll
  for (int k = 0; k < CO * 9; ++k) acc[k] = 0.f;
#pragma unroll
  for (int c = 0; c < CO; ++c) bacc[c] = 0.f;
  const long n = (long)B * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int wo = (int)(i % Wo);
    const long r = i / Wo;
    const int ho = (int)(r % Ho), b = (int)(r / Ho);
    float xv[9], gv[CO];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = wo + kw - pw;
        const float t = x[((long)b * H + ho + kh) * W + min(max(ww, 0), W - 1)];
        xv[kh * 3 + kw] = (ww >= 0 && ww < W) ? t : 0.f;
      }
    const float4* gp = reinterpret_cast<const float4*>(g + i * CO);
#pragma unroll
    for (int c = 0; c < CO; c += 4) {
      const float4 t = gp[c / 4];
      gv[c] = t.x;
      gv[c + 1] = t.y;
      gv[c + 2] = t.z;
      gv[c + 3] = t.w;
    }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      bacc[c] += gv[c];
#pragma unroll
      for (int k = 0; k < 9; ++k) acc[c * 9 + k] = fmaf(gv[c], xv[k], acc[c * 9 + k]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < CO * 9; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) s_red[wave][k] = v;
  }
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    const float v = wave_sum(bacc[c]);
    if (lane == 0) s_red[wave][CO * 9 + c] = v;
  }
  __syncthreads();
  if (threadIdx.x < CO * 10) {
    const float v = (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) +
                    (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
    if (threadIdx.x < CO * 9) atomicAdd(dw + threadIdx.x, v);
    else if (db) atomicAdd(db + (threadIdx.x - CO * 9), v);
  }
}

// dx[b,h,w] = sum_{co,kh,kw} g[b, h-kh, w-kw+pw, co] * w[co][kh][kw]   (gather)
template <int CO>
__global__ __launch_bounds__(256) void conv3x3_c1_dgrad_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ w, int B,
                                                               int H, int W, int pw, int Ho, int Wo,
                                                               float* __restrict__ dx) {
  float wv[CO * 9];
#pragma unroll
  for (int k = 0; k < CO * 9; ++k) wv[k] = w[k];
  const long n = (long)B * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int ww = (int)(i % W);
    const long r = i / W;
    const int h = (int)(r % H), b = (int)(r / H);
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ho = h - kh;
      if (ho < 0 || ho >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int wo = ww - kw + pw;
        if (wo < 0 || wo >= Wo) continue;
        const float4* gp = reinterpret_cast<const float4*>(g + (((long)b * Ho + ho) * Wo + wo) * CO);
#pragma unroll
        for (int c = 0; c < CO; c += 4) {
          const float4 t = gp[c / 4];
          acc = fmaf(t.x, wv[c * 9 + kh * 3 + kw], acc);
          acc = fmaf(t.y, wv[(c + 1) * 9 + kh * 3 + kw], acc);
          acc = fmaf(t.z, wv[(c + 2) * 9 + kh * 3 + kw], acc);
          acc = fmaf(t.w, wv[(c + 3) * 9 + kh * 3 + kw], acc);
        }
      }
    }
    dx[i] = acc;
  }
}

inline unsigned grid_c1(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

// mode 0: y = conv(x) + bias;  1: dw / db += (accumulated; caller zeroes);  2: dx.
// Only CO == 8 (the reference's layer1_channels) is instantiated; -2 asks the caller to use the
// im2col path.
extern "C" int s2t_conv3x3_c1(int mode, const float* x, const float* w, const float* bias,
                              const float* g, int B, int H, int W, int pw, int CO, float* y,
                              float* dw, float* db, float* dx, void* stream) {
  if (B <= 0 || H < 3 || W <= 0) return -1;
  if (CO != 8 || pw < 0 || pw > 1) return -2;
  const int Ho = H - 2, Wo = W + 2 * pw - 2;
  if (Wo <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)
    hipLaunchKernelGGL(conv3x3_c1_fwd_kernel<8>, dim3(grid_c1((long)B * Ho * Wo)), dim3(256), 0, st,
                       x, w, bias, B, H, W, pw, Ho, Wo, y);
  else if (mode == 1)
    // (1 024 workgroups, measured: every workgroup ends with 80 atomics on the same 80 words, which serialise
    //  -- 2 048 / 4 096 workgroups take 161 / 215 us against 128 -- but the scalar tap loads need the waves:
    //  512 / 256 take 164 / 250)
    hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel<8>, dim3(1024), dim3(256), 0, st, x, g, B, H, W, pw,
                       Ho, Wo, dw, db);
  else if (mode == 2)
    hipLaunchKernelGGL(conv3x3_c1_dgrad_kernel<8>, dim3(grid_c1((long)B * H * W)), dim3(256), 0, st,
                       g, w, B, H, W, pw, Ho, Wo, dx);
  else
    return -1;
  S2T_CHECK_LAUNCH();
  return 0;
}

namespace {

// ---------------------------------------------------------------- second subsampling conv, direct
// Conv2d(CI = 8, CO = 32, 3, stride 2) on channel-last data (model/layer/subsampling.py:184-229).
// As im2col + GEMM it is a 72-deep, 32-wide product over 1.2 M rows: the 357 MB patch matrix is
// written and read back and the library GEMM runs far below its rate.  Direct form: a thread owns
// one output position, keeps its 72 inputs and 32 accumulators in registers and takes the weights
// through wave-uniform (scalar) loads.  x (B,H,W,CI) -> y (B,Ho,Wo,CO), Ho = (H-3)/2+1.
template <int CI, int CO>
__global__ __launch_bounds__(256) void conv3x3_s2_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ bias, int B,
                                                             int H, int W, int Ho, int Wo,
                                                             float* __restrict__ y) {
  static_assert(CI % 4 == 0 && CO % 4 == 0, "float4 rows");
  // (measured alternatives: two positions per thread 271 us, lane = output channel with the taps
  // in registers 887 us; this form 213 us against 360 us for im2col + library GEMM)
  const long n = (long)B * Ho * Wo;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int wo = (int)(i % Wo);
  const long r = i / Wo;
  const int ho = (int)(r % Ho), b = (int)(r / Ho);
  float xv[9 * CI];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float4* p = reinterpret_cast<const float4*>(
          x + (((long)b * H + 2 * ho + kh) * W + 2 * wo + kw) * CI);
#pragma unroll
      for (int q = 0; q < CI / 4; ++q) {
        const float4 t = p[q];
        xv[(kh * 3 + kw) * CI + 4 * q] = t.x;
        xv[(kh * 3 + kw) * CI + 4 * q + 1] = t.y;
        xv[(kh * 3 + kw) * CI + 4 * q + 2] = t.z;
        xv[(kh * 3 + kw) * CI + 4 * q + 3] = t.w;
      }
    }
  float4* o = reinterpret_cast<float4*>(y + i * CO);
#pragma unroll
  for (int c0 = 0; c0 < CO; c0 += 4) {
    float acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = bias ? bias[c0 + c] : 0.f;
      // nn.Conv2d weight (CO, CI, 3, 3): element (co, ci, kh, kw) at ((co*CI + ci)*3 + kh)*3 + kw
#pragma unroll
      for (int ci = 0; ci < CI; ++ci)
#pragma unroll
        for (int k = 0; k < 9; ++k)
          acc[c] = fmaf(w[((c0 + c) * CI + ci) * 9 + k], xv[k * CI + ci], acc[c]);
    }
    o[c0 / 4] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// dx[b,h,w,ci] = sum_{kh,kw,co} g[b,(h-kh)/2,(w-kw)/2,co] w[co,ci,kh,kw] over the taps whose
// (h-kh, w-kw) are even and in range.  A thread owns the input pair (h, 2p) / (h, 2p+1): the
// even column takes kw = 0 (wo = p) and kw = 2 (wo = p-1), the odd one kw = 1 (wo = p); the row
// parity (kh in {0,2} or {1}) is uniform per workgroup row, so every weight index is wave-uniform.
template <int CI, int CO>
__global__ __launch_bounds__(256) void conv3x3_s2_dgrad_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ w, int B,
                                                               int H, int W, int Ho, int Wo,
                                                               float* __restrict__ dx) {
  // a WAVE per input row (64 column pairs), four rows of one parity per workgroup: with a whole
  // workgroup per row the 39 pairs of the 78-wide map kept 15 % of the lanes busy (302 us for a
  // 314 MB pass)
  const int b = blockIdx.z, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: h, and with
  const int p = blockIdx.x * 64 + (threadIdx.x & 63);                                 // it every weight index, stays wave-uniform)
  const int npair = (W + 1) / 2;
  const int ge = ((H + 1) / 2 + 3) / 4;      // workgroups of even rows come first
  const int h = blockIdx.y < ge ? 2 * ((int)blockIdx.y * 4 + wv) : 1 + 2 * (((int)blockIdx.y - ge) * 4 + wv);
  if (h >= H || p >= npair) return;
  float ae[CI], ao[CI];                      // gradients of columns 2p and 2p+1
#pragma unroll
  for (int c = 0; c < CI; ++c) ae[c] = ao[c] = 0.f;
  for (int kh = (h & 1); kh < 3; kh += 2) {  // uniform: h even -> 0, 2; h odd -> 1
    const int hh = h - kh;
    if (hh < 0 || hh / 2 >= Ho) continue;
    const float* grow = g + ((long)b * Ho + hh / 2) * Wo * CO;
    // g pixels wo = p (taps kw 0 and 1) and wo = p - 1 (tap kw 2), from clamped addresses
    const bool v0 = p < Wo, v1 = p - 1 >= 0 && p - 1 < Wo;
    const float4* g0 = reinterpret_cast<const float4*>(grow + (long)min(p, Wo - 1) * CO);
    const float4* g1 = reinterpret_cast<const float4*>(grow + (long)min(max(p - 1, 0), Wo - 1) * CO);
#pragma unroll
    for (int q = 0; q < CO / 4; ++q) {
      float4 t0 = g0[q], t1 = g1[q];
      if (!v0) t0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!v1) t1 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float a0[4] = {t0.x, t0.y, t0.z, t0.w}, a1[4] = {t1.x, t1.y, t1.z, t1.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int co = 4 * q + c;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const float* wp = w + ((co * CI + ci) * 3 + kh) * 3;
          ae[ci] = fmaf(a0[c], wp[0], ae[ci]);          // even column, kw = 0, wo = p
          ao[ci] = fmaf(a0[c], wp[1], ao[ci]);          // odd column,  kw = 1, wo = p
          ae[ci] = fmaf(a1[c], wp[2], ae[ci]);          // even column, kw = 2, wo = p - 1
        }
      }
    }
  }
  float* o = dx + (((long)b * H + h) * W + 2 * p) * CI;
#pragma unroll
  for (int q = 0; q < CI / 4; ++q)
    reinterpret_cast<float4*>(o)[q] = make_float4(ae[4 * q], ae[4 * q + 1], ae[4 * q + 2], ae[4 * q + 3]);
  if (2 * p + 1 < W) {
#pragma unroll
    for (int q = 0; q < CI / 4; ++q)
      reinterpret_cast<float4*>(o + CI)[q] =
          make_float4(ao[4 * q], ao[4 * q + 1], ao[4 * q + 2], ao[4 * q + 3]);
  }
}

}  // namespace

// mode 0: y = conv(x) + bias;  mode 2: dx from (g, w).  Only (CI, CO) = (8, 32), stride 2, no
// padding (the reference's second subsampling conv) is instantiated; -2 = use im2col + GEMM.
extern "C" int s2t_conv3x3_s2(int mode, const float* x, const float* w, const float* bias,
                              const float* g, int B, int H, int W, int CI, int CO, float* y,
                              float* dx, void* stream) {
  if (B <= 0 || H < 3 || W < 3) return -1;
  if (CI != 8 || CO != 32) return -2;
  const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) {
    const long n = (long)B * Ho * Wo;
    hipLaunchKernelGGL((conv3x3_s2_fwd_kernel<8, 32>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       st, x, w, bias, B, H, W, Ho, Wo, y);
  } else if (mode == 2) {
    if (B > 65535 || H > 65535) return -2;
    hipLaunchKernelGGL((conv3x3_s2_dgrad_kernel<8, 32>),
                       dim3(((W + 1) / 2 + 63) / 64, ((H + 1) / 2 + 3) / 4 + (H / 2 + 3) / 4, B),
                       dim3(256), 0, st, g, w, B, H, W, Ho, Wo, dx);
  } else {
    return -1;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}
