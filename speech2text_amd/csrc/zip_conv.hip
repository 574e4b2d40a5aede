// Fused zipformer convolution-module core for gfx950, time-major (T,B,C):
//   xg = x * sigmoid(s)   (GLU-style gate; x,s = the two halves of in_proj's output)
//   xg = 0 on padded frames
//   y  = causal_dwconv(xg) + chunkwise_dwconv(xg) * edge_scale        (causal=True)
//   y  = dwconv(xg)                                                   (plain nn.Conv1d)
// Reference: model/encoder/zipformer.py:2672-2690 (gate, permute, masked_fill, depthwise
// conv) + model/layer/scaling.py:622-681 (ChunkCausalDepthwiseConv1d.forward, _get_chunk_scale).
// The reference permutes to (B,C,T), pads, runs two cuDNN/MIOpen depthwise convs, reshapes
// chunks and permutes back; here one kernel reads the (T,B,2C) projection once and writes
// (T,B,C) once.  Tile = 64 frames x 64 channels of one utterance; gated inputs (+halo) and the
// filter taps live in LDS; each thread slides a register window over 16 consecutive frames of
// its channel, so LDS traffic is ~2 reads per 16 MACs.  Backward = one data kernel (transposed
// taps, then back through the gate) and one weight kernel (per-thread tap partials, LDS
// reduction over the 4 frame groups, one atomic per tap per block).
#include "common.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace {

constexpr int TT = 64;   // frames per tile
constexpr int FPT = 16;  // frames per thread

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// per-position scale of the chunkwise conv output (1 + left_edge + right_edge)
__device__ __forceinline__ float edge_scale(const float* __restrict__ s_le,
                                            const float* __restrict__ s_re, int c, int pos,
                                            int chunk, int K) {
  float sc = 1.f;
  if (pos < K) sc += s_le[pos * 64 + c];
  const int idx = pos - chunk + K;
  if (idx >= 0 && idx < K) sc += s_re[idx * 64 + c];
  return sc;
}

struct ConvArgs {
  const float* u;       // (T,B,ld) projection; x at [0,C), gate at [gate_off, gate_off+C)
  long ld;
  int gate_off;         // <0: no gate
  const unsigned char* mask;  // (B,T) 1 = padded frame, may be null
  int T, B, C, chunk;
  const float* wc;      // (C,Kh) causal taps or null
  const float* bc;      // (C) or null
  const float* wk;      // (C,K)
  const float* bk;      // (C) or null
  const float* scale;   // (2,C,K) or null
  // t-tile of a workgroup = blockIdx.z + z_base, + z_jump from z_split on: a launch covers either the
  // INTERIOR tiles [n_lo, n_hi) (no frame within K of a chunk edge: edge scale == 1, no edge arrays
  // and no scaled-gradient tile in LDS -- 36 instead of 52 / 76 KB at K = 31) or the others
  int z_base, z_split, z_jump;
  int nt;               // t-tiles in all (partial-sum slots of the weight kernel)
};

__device__ __forceinline__ int tile_z(const ConvArgs& a) {
  int tz = (int)blockIdx.z + a.z_base;
  if (tz >= a.z_split) tz += a.z_jump;
  return tz;
}

template <int K>
__device__ __forceinline__ void stage_weights(const ConvArgs& a, int c0, float* s_wc, float* s_wk,
                                              float* s_le, float* s_re, bool need_scale) {
  // all of a thread's tap loads are issued back to back from clamped (always valid) addresses;
  // validity is applied on the LDS store (loads behind per-lane conditions serialise)
  constexpr int Kh = (K + 1) / 2;
  constexpr int NI = (64 * K + 255) / 256, NH = (64 * Kh + 255) / 256;
  // need_scale (block-uniform): some frame this workgroup touches lies within K of a chunk edge;
  // elsewhere the edge arrays are never read and their 2 K x 64 loads + LDS stores are skipped
  const bool sc = a.scale != nullptr && need_scale;
  float wk[NI], le[NI], re[NI], wc[NH];
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int i = min((int)threadIdx.x + 256 * q, 64 * K - 1);
    const int c = min(c0 + i / K, a.C - 1), j = i % K;
    wk[q] = a.wk[(long)c * K + j];
    le[q] = sc ? a.scale[(long)c * K + j] : 0.f;
    re[q] = sc ? a.scale[((long)a.C + c) * K + j] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int i = min((int)threadIdx.x + 256 * q, 64 * Kh - 1);
    const int c = min(c0 + i / Kh, a.C - 1), j = i % Kh;
    wc[q] = a.wc ? a.wc[(long)c * Kh + j] : 0.f;
  }
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int i = threadIdx.x + 256 * q;
    if (i < 64 * K) {
      const int c = i / K, j = i % K;
      const bool ok = c0 + c < a.C;
      s_wk[j * 64 + c] = ok ? wk[q] : 0.f;
      if (need_scale) {
        s_le[j * 64 + c] = ok ? le[q] : 0.f;
        s_re[j * 64 + c] = ok ? re[q] : 0.f;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    const int i = threadIdx.x + 256 * q;
    if (i < 64 * Kh) {
      const int c = i / Kh, j = i % Kh;
      s_wc[j * 64 + c] = (c0 + c < a.C) ? wc[q] : 0.f;
    }
  }
}

// gated, masked input tile: rows r <-> frame t0 - halo + r.  Thread = (channel c, row group);
// the loads of a batch of NB rows are issued back to back from clamped (always valid) addresses
// -- padding mask byte, x and gate of every row in flight together -- and validity is applied
// afterwards.  (Loads behind per-element `if (t in range && !mask[t])` branches serialise into
// one global round trip per row, which is what bounded these kernels before.)
template <int K>
__device__ __forceinline__ void stage_xg(const ConvArgs& a, int b, int t0, int c0, float* s_x) {
  constexpr int halo = K / 2;
  constexpr int rows = TT + 2 * halo;
  constexpr int NB = 8;
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const bool cok = c0 + c < a.C;
  const int cc = cok ? c0 + c : 0;
  const bool gated = a.gate_off >= 0;
  for (int r0 = rg; r0 < rows; r0 += 4 * NB) {
    float xv[NB], gv[NB];
    unsigned char mk[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int t = min(max(t0 - halo + r0 + 4 * i, 0), a.T - 1);
      const float* row = a.u + ((long)t * a.B + b) * a.ld;
      mk[i] = a.mask ? a.mask[(long)b * a.T + t] : (unsigned char)0;
      xv[i] = row[cc];
      gv[i] = gated ? row[a.gate_off + cc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = r0 + 4 * i, t = t0 - halo + r;
      if (r < rows) {
        const bool ok = cok && t >= 0 && t < a.T && !mk[i];
        float v = xv[i];
        if (gated) v *= sigmoidf_(gv[i]);
        s_x[r * 64 + c] = ok ? v : 0.f;
      }
    }
  }
}

// The same tile through BUFFER loads (the recipe that took s2t_attn_apply from 3.2 to 4.0 TB/s):
// lane = 4 channels of one row, so a wave-instruction moves 4 whole rows (16-byte pieces, any dword
// alignment), all of a thread's loads -- x and gate of its 6 rows -- are issued as ONE batch, and
// frames outside [0, T) need no clamping: their offsets fall outside the buffer and read as 0.
// One 32-bit offset register per load instead of a 64-bit address pair.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool buf_ok(const ConvArgs& a) {
  return (long)a.T * a.B * a.ld * 4 < 0x7FFFFF00L;
}

template <int K>
__device__ __forceinline__ void stage_xg_buf(const ConvArgs& a, int b, int t0, int c0, float* s_x) {
  constexpr int halo = K / 2, rows = TT + 2 * halo, NP = (rows + 15) / 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u), 0, (int)((long)a.T * a.B * a.ld * 4), 0x00020000);
  const int rig = threadIdx.x >> 4, c4 = threadIdx.x & 15, ch = c0 + 4 * c4;
  const bool gated = a.gate_off >= 0;
  u32x4 xv[NP], gv[NP];
  unsigned char mk[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int t = t0 - halo + 16 * p + rig;
    // rows beyond the tile's last one (p = NP - 1 only) are not stored; their loads are harmless
    const int off = (((t * a.B + b) * (int)a.ld) + ch) * 4;
    const bool tin = t >= 0 && t < a.T;
    xv[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, tin ? off : 0x7FFFFFF0, 0, 0);
    gv[p] = gated ? __builtin_amdgcn_raw_buffer_load_b128(rs, tin ? off + a.gate_off * 4 : 0x7FFFFFF0, 0, 0)
                  : (u32x4){0u, 0u, 0u, 0u};
    mk[p] = a.mask ? a.mask[(long)b * a.T + min(max(t, 0), a.T - 1)] : (unsigned char)0;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int r = 16 * p + rig;
    if (r < rows) {
      float v[4] = {__uint_as_float(xv[p].x), __uint_as_float(xv[p].y), __uint_as_float(xv[p].z),
                    __uint_as_float(xv[p].w)};
      const float g[4] = {__uint_as_float(gv[p].x), __uint_as_float(gv[p].y),
                          __uint_as_float(gv[p].z), __uint_as_float(gv[p].w)};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (gated) v[e] *= sigmoidf_(g[e]);
        if (mk[p] || ch + e >= a.C) v[e] = 0.f;       // (frames outside [0,T) were read as 0)
      }
      *reinterpret_cast<float4*>(&s_x[r * 64 + 4 * c4]) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// Swoosh as zip_elem.hip swoosh_f (the same instruction sequence: the fused output equals s2t_swoosh_fwd's)
__device__ __forceinline__ float conv_swoosh(float x, float off, float c) {
  const float z = x - off;
  const float e = __expf(-fabsf(z));
  const float u = 1.f + e;
  const float l1p = u == 1.f ? e : __logf(u) * __fdividef(e, u - 1.f);
  return fmaxf(z, 0.f) + l1p - 0.08f * x - c;
}

template <int K, bool GEN, bool EDGE, int SB = 0>   // SB: see zipconv_bwd_w_kernel
__global__ __launch_bounds__(256) void zipconv_fwd_kernel(ConvArgs a, float* __restrict__ y,
                                                          float* __restrict__ y2, float act_off,
                                                          float act_c) {
  // y2 (optional): Swoosh(y) as a second output -- the conv module's activation (zipformer.py:
  // 2700-2703) leaves with the tile instead of re-reading y in a pass of its own
  constexpr int Kh = (K + 1) / 2, halo = K / 2, W = FPT + 2 * halo;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s_x = reinterpret_cast<float*>(smem_raw);
  float* s_wc = s_x + (TT + 2 * halo) * 64;
  float* s_wk = s_wc + Kh * 64;
  float* s_le = EDGE ? s_wk + K * 64 : nullptr;
  float* s_re = EDGE ? s_le + K * 64 : nullptr;
  // channel tile fastest, then utterance, then frame tile: workgroups that are dispatched together
  // read neighbouring pieces of the same (t, b) rows (rows of adjacent b are adjacent in memory)
  const int c0 = blockIdx.x * 64, b = blockIdx.y, t0 = tile_z(a) * TT;
  const int c = threadIdx.x & 63, tg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: frame indices and chunk tests stay wave-uniform)
  stage_weights<K>(a, c0, s_wc, s_wk, s_le, s_re,
                   EDGE && (GEN || t0 - K / 2 < K || t0 + TT + K / 2 > a.T - K));
  if (buf_ok(a)) stage_xg_buf<K>(a, b, t0, c0, s_x);
  else stage_xg<K>(a, b, t0, c0, s_x);
  __syncthreads();
  const bool chan_ok = c0 + c < a.C;
  const int cc = chan_ok ? c0 + c : 0;
  float win[W];
#pragma unroll
  for (int w = 0; w < W; ++w) win[w] = s_x[(tg * FPT + w) * 64 + c];
  const float bc = a.bc ? a.bc[cc] : 0.f;
  const float bk = a.bk ? a.bk[cc] : 0.f;
  const int tb = t0 + tg * FPT;
  float accc[FPT], acck[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    accc[i] = bc;
    acck[i] = bk;
  }
  const int chunk = a.chunk;
  if (a.wc) {
#pragma unroll
    for (int j = 0; j < Kh; ++j) {
      const float w = s_wc[j * 64 + c];
#pragma unroll
      for (int i = 0; i < FPT; ++i) accc[i] = fmaf(w, win[i + j], accc[i]);
    }
  }
  if (!GEN) {
    // single chunk: frames outside [0,T) are zero in the tile = the conv's zero padding
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float w = s_wk[j * 64 + c];
#pragma unroll
      for (int i = 0; i < FPT; ++i) acck[i] = fmaf(w, win[i + j], acck[i]);
    }
  } else if constexpr (SB > 0) {
    // sub-blocks of SB frames inside one chunk each: mask the window once, plain multiply-adds
#pragma unroll
    for (int sb = 0; sb < FPT / SB; ++sb) {
      constexpr int WM = SB + K - 1;
      const int i0 = sb * SB, lo = ((tb + i0) / chunk) * chunk - (tb - halo + i0);
      float wm[WM];
#pragma unroll
      for (int w = 0; w < WM; ++w) wm[w] = (w >= lo && w < lo + chunk) ? win[i0 + w] : 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float w = s_wk[j * 64 + c];
#pragma unroll
        for (int i = 0; i < SB; ++i) acck[i0 + i] = fmaf(w, wm[i + j], acck[i0 + i]);
      }
    }
  } else {
    int cs[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i) cs[i] = ((tb + i) / chunk) * chunk;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float w = s_wk[j * 64 + c];
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int tt = tb + i - halo + j;
        if (tt >= cs[i] && tt < cs[i] + chunk) acck[i] = fmaf(w, win[i + j], acck[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int t = tb + i;
    if (EDGE && a.scale && t < a.T) acck[i] *= edge_scale(s_le, s_re, c, t % chunk, chunk, K);
    acck[i] += accc[i];
  }
  if ((a.C & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
    // the tile leaves as 16-byte row pieces (lane = 4 channels; a wave-instruction = 4 whole rows
    // of the tile): per-lane 4-byte stores of 64 channels are a quarter of that per instruction
    __syncthreads();                               // every window is in registers: s_x is free
#pragma unroll
    for (int i = 0; i < FPT; ++i) s_x[(tg * FPT + i) * 64 + c] = acck[i];
    __syncthreads();
    const int rig = threadIdx.x >> 4, c4 = threadIdx.x & 15, ch = c0 + 4 * c4;
#pragma unroll
    for (int p = 0; p < TT / 16; ++p) {
      const int r = 16 * p + rig, t = t0 + r;
      if (t < a.T && ch < a.C) {
        const float4 v = *reinterpret_cast<const float4*>(&s_x[r * 64 + 4 * c4]);
        *reinterpret_cast<float4*>(y + ((long)t * a.B + b) * a.C + ch) = v;
        if (y2)
          *reinterpret_cast<float4*>(y2 + ((long)t * a.B + b) * a.C + ch) =
              make_float4(conv_swoosh(v.x, act_off, act_c), conv_swoosh(v.y, act_off, act_c),
                          conv_swoosh(v.z, act_off, act_c), conv_swoosh(v.w, act_off, act_c));
      }
    }
    return;
  }
  if (!chan_ok) return;
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int t = tb + i;
    if (t < a.T) {
      y[((long)t * a.B + b) * a.C + c0 + c] = acck[i];
      if (y2) y2[((long)t * a.B + b) * a.C + c0 + c] = conv_swoosh(acck[i], act_off, act_c);
    }
  }
}

// du[t'] : gradient w.r.t. the projection (x half and gate half)
template <int K, bool GEN, bool EDGE, int SB = 0>   // SB: see zipconv_bwd_w_kernel
__global__ __launch_bounds__(256) void zipconv_bwd_data_kernel(ConvArgs a,
                                                               const float* __restrict__ dy,
                                                               float* __restrict__ du) {
  constexpr int Kh = (K + 1) / 2, halo = K / 2, W = FPT + 2 * halo;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s_g = reinterpret_cast<float*>(smem_raw);  // dy tile with halo
  float* s_wc = s_g + (TT + 2 * halo) * 64;
  float* s_wk = s_wc + Kh * 64;
  float* s_le = EDGE ? s_wk + K * 64 : nullptr;
  float* s_re = EDGE ? s_le + K * 64 : nullptr;
  // channel tile fastest, then utterance, then frame tile: workgroups that are dispatched together
  // read neighbouring pieces of the same (t, b) rows (rows of adjacent b are adjacent in memory)
  const int c0 = blockIdx.x * 64, b = blockIdx.y, t0 = tile_z(a) * TT;
  const int c = threadIdx.x & 63, tg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: frame indices and chunk tests stay wave-uniform)
  // (block-uniform) some frame of the tile or its halo lies within K of a chunk edge
  const bool near_edge = EDGE && (GEN || t0 - K / 2 < K || t0 + TT + K / 2 > a.T - K);
  stage_weights<K>(a, c0, s_wc, s_wk, s_le, s_re, near_edge);
  __syncthreads();
  const int chunk = a.chunk;
  // ONE tile, the raw dy (causal taps); the chunkwise taps see dy * edge_scale, applied to the
  // register window of the tiles that have an edge frame (a second, scaled tile in LDS cost a
  // workgroup per CU)
  const bool dy_buf = (long)a.T * a.B * a.C * 4 < 0x7FFFFF00L;
  if (dy_buf) {
    // one batch of 16-byte buffer loads (lane = 4 channels of a row; frames outside [0,T) read 0)
    constexpr int rows = TT + 2 * halo, NP = (rows + 15) / 16;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(dy), 0, (int)((long)a.T * a.B * a.C * 4), 0x00020000);
    const int rig = threadIdx.x >> 4, c4 = threadIdx.x & 15, ch = c0 + 4 * c4;
    u32x4 gq[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int t = t0 - halo + 16 * p + rig;
      const int off = ((t * a.B + b) * a.C + ch) * 4;
      gq[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, (t >= 0 && t < a.T) ? off : 0x7FFFFFF0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int r = 16 * p + rig;
      if (r < rows) {
        float v[4] = {__uint_as_float(gq[p].x), __uint_as_float(gq[p].y), __uint_as_float(gq[p].z),
                      __uint_as_float(gq[p].w)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (ch + e >= a.C) v[e] = 0.f;
        *reinterpret_cast<float4*>(&s_g[r * 64 + 4 * c4]) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  } else {
    constexpr int rows = TT + 2 * halo, NB = 8;
    const bool cok = c0 + c < a.C;
    const int cc = cok ? c0 + c : 0;
    for (int r0 = tg; r0 < rows; r0 += 4 * NB) {
      float gv[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int t = min(max(t0 - halo + r0 + 4 * i, 0), a.T - 1);
        gv[i] = dy[((long)t * a.B + b) * a.C + cc];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int r = r0 + 4 * i, t = t0 - halo + r;
        if (r < rows) {
          const bool ok = cok && t >= 0 && t < a.T;
          s_g[r * 64 + c] = ok ? gv[i] : 0.f;
        }
      }
    }
  }
  __syncthreads();
  const int tb = t0 + tg * FPT;
  const bool chan_ok = c0 + c < a.C;
  const bool gated = a.gate_off >= 0;
  const int nout = gated ? 2 * a.C : a.C;
  // 16-byte form of everything that is not the tap loops: the projection values the gate's backward
  // needs arrive as row pieces (lane = 4 channels of a row, a wave-instruction = 4 whole rows of the
  // tile) and the result leaves the same way, through LDS -- per-lane 4-byte accesses of 64 channels
  // move a quarter of that per instruction
  const bool vec = (a.C & 3) == 0 && (a.ld & 3) == 0 && (!gated || (a.gate_off & 3) == 0) &&
                   ((reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(a.u)) & 15) == 0 && buf_ok(a);
  const int rig = threadIdx.x >> 4, c4 = threadIdx.x & 15, ch = c0 + 4 * c4;
  u32x4 qx[TT / 16], qs[TT / 16];
  unsigned char qpad[TT / 16];
  // the scalar form's operands, issued now and consumed after the tap loops
  float oxv[FPT], osv[FPT];
  unsigned char opad[FPT];
  if (vec) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.u), 0, (int)((long)a.T * a.B * a.ld * 4), 0x00020000);
#pragma unroll
    for (int p = 0; p < TT / 16; ++p) {
      const int t = t0 + 16 * p + rig;
      const bool tin = t < a.T && ch < a.C;
      const int off = (((t * a.B + b) * (int)a.ld) + ch) * 4;
      qx[p] = gated ? __builtin_amdgcn_raw_buffer_load_b128(rs, tin ? off : 0x7FFFFFF0, 0, 0)
                    : (u32x4){0u, 0u, 0u, 0u};
      qs[p] = gated ? __builtin_amdgcn_raw_buffer_load_b128(rs, tin ? off + a.gate_off * 4 : 0x7FFFFFF0, 0, 0)
                    : (u32x4){0u, 0u, 0u, 0u};
      qpad[p] = a.mask ? a.mask[(long)b * a.T + min(t, a.T - 1)] : (unsigned char)0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      const int t = min(tb + i, a.T - 1);
      const float* row = a.u + ((long)t * a.B + b) * a.ld;
      const int cc = chan_ok ? c0 + c : 0;
      opad[i] = a.mask ? a.mask[(long)b * a.T + t] : (unsigned char)0;
      oxv[i] = gated ? row[cc] : 0.f;
      osv[i] = gated ? row[a.gate_off + cc] : 0.f;
    }
  }
  float win[W];   // frames tb-halo .. tb+FPT-1+halo
  float acc[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) acc[i] = 0.f;
  // causal: y[t] += wc[j] * xg[t - halo + j]  =>  dxg[t'] += wc[j] * dy[t' + halo - j]
  if (a.wc) {
#pragma unroll
    for (int w = 0; w < W; ++w) win[w] = s_g[(tg * FPT + w) * 64 + c];
#pragma unroll
    for (int j = 0; j < Kh; ++j) {
      const float w = s_wc[j * 64 + c];
#pragma unroll
      for (int i = 0; i < FPT; ++i) acc[i] = fmaf(w, win[i + 2 * halo - j], acc[i]);
    }
  }
  // chunkwise: dxg[t'] += wk[j] * (dy*sc)[t' + halo - j] when frame t'+halo-j is in t's chunk
  if (!a.wc) {
#pragma unroll
    for (int w = 0; w < W; ++w) win[w] = s_g[(tg * FPT + w) * 64 + c];
  }
  if (near_edge && a.scale) {
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int t = tb - halo + w;
      if (t >= 0 && t < a.T) win[w] *= edge_scale(s_le, s_re, c, GEN ? t % chunk : t, chunk, K);
    }
  }
  if (!GEN) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float w = s_wk[j * 64 + c];
#pragma unroll
      for (int i = 0; i < FPT; ++i) acc[i] = fmaf(w, win[i + 2 * halo - j], acc[i]);
    }
  } else if constexpr (SB > 0) {
#pragma unroll
    for (int sb = 0; sb < FPT / SB; ++sb) {
      constexpr int WM = SB + K - 1;
      const int i0 = sb * SB, lo = ((tb + i0) / chunk) * chunk - (tb - halo + i0);
      float wm[WM];
#pragma unroll
      for (int w = 0; w < WM; ++w) wm[w] = (w >= lo && w < lo + chunk) ? win[i0 + w] : 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float w = s_wk[j * 64 + c];
#pragma unroll
        for (int i = 0; i < SB; ++i) acc[i0 + i] = fmaf(w, wm[i + 2 * halo - j], acc[i0 + i]);
      }
    }
  } else {
    int cs[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i) cs[i] = ((tb + i) / chunk) * chunk;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const float w = s_wk[j * 64 + c];
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int t = tb + i + halo - j;
        if (t >= cs[i] && t < cs[i] + chunk) acc[i] = fmaf(w, win[i + 2 * halo - j], acc[i]);
      }
    }
  }
  if (vec) {
    __syncthreads();                               // every window is in registers: s_g is free
#pragma unroll
    for (int i = 0; i < FPT; ++i) s_g[(tg * FPT + i) * 64 + c] = acc[i];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < TT / 16; ++p) {
      const int r = 16 * p + rig, t = t0 + r;
      if (t >= a.T || ch >= a.C) continue;
      const float4 d = *reinterpret_cast<const float4*>(&s_g[r * 64 + 4 * c4]);
      const float dv[4] = {d.x, d.y, d.z, d.w};
      float* o = du + ((long)t * a.B + b) * nout + ch;
      const bool pad = qpad[p] != 0;
      if (gated) {
        const float xv[4] = {__uint_as_float(qx[p].x), __uint_as_float(qx[p].y), __uint_as_float(qx[p].z),
                             __uint_as_float(qx[p].w)};
        const float sv[4] = {__uint_as_float(qs[p].x), __uint_as_float(qs[p].y), __uint_as_float(qs[p].z),
                             __uint_as_float(qs[p].w)};
        float ox[4], og[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sg = sigmoidf_(sv[e]);
          ox[e] = pad ? 0.f : dv[e] * sg;
          og[e] = pad ? 0.f : dv[e] * xv[e] * sg * (1.f - sg);
        }
        *reinterpret_cast<float4*>(o) = make_float4(ox[0], ox[1], ox[2], ox[3]);
        *reinterpret_cast<float4*>(o + a.C) = make_float4(og[0], og[1], og[2], og[3]);
      } else {
        *reinterpret_cast<float4*>(o) = pad ? make_float4(0.f, 0.f, 0.f, 0.f) : d;
      }
    }
    return;
  }
  if (!chan_ok) return;
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int t = tb + i;
    if (t < a.T) {
      const bool pad = opad[i] != 0;
      float* o = du + ((long)t * a.B + b) * nout;
      if (gated) {
        const float xv = oxv[i], sg = sigmoidf_(osv[i]);
        o[c0 + c] = pad ? 0.f : acc[i] * sg;
        o[a.C + c0 + c] = pad ? 0.f : acc[i] * xv * sg * (1.f - sg);
      } else {
        o[c0 + c] = pad ? 0.f : acc[i];
      }
    }
  }
}

// weight / bias gradients; block = (t-tile, group of BB utterances, c-tile).  Each block
// writes its partial sums to part[block][NV][64 ch] (NV = Kh + 1 + K + 1); zipconv_reduce_w_kernel
// sums over blocks -- no atomics (contended float atomics on a few KB run ~14x slower).
// GEN with SB > 0 (chunk a power of two, SB = min(chunk, FPT)): the FPT frames of a thread fall
// into FPT / SB sub-blocks that lie in ONE chunk each, so "tap j of frame i is inside the chunk"
// does not depend on (i, j) separately -- it is a property of the WINDOW element i + j.  Each
// sub-block masks its SB + K - 1 window elements once and runs the plain multiply-adds; the
// per-(frame, tap) test (SB = 0, any chunk) cost three instructions per multiply-add: 218 us
// against 55 for the unchunked kernel at T = 495.
template <int K, bool GEN, bool EDGE, int SB = 0>
__global__ __launch_bounds__(256) void zipconv_bwd_w_kernel(ConvArgs a,
                                                            const float* __restrict__ dy, int BB,
                                                            float* __restrict__ part,
                                                            float* __restrict__ dscale,
                                                            float* __restrict__ dscale_t = nullptr,
                                                            int cpad = 0) {
  constexpr int Kh = (K + 1) / 2, halo = K / 2, W = FPT + 2 * halo, NV = Kh + K + 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s_x = reinterpret_cast<float*>(smem_raw);
  float* s_wc = s_x + (TT + 2 * halo) * 64;
  float* s_wk = s_wc + Kh * 64;
  float* s_le = EDGE ? s_wk + K * 64 : nullptr;
  float* s_re = EDGE ? s_le + K * 64 : nullptr;
  float* s_red = s_wk + (EDGE ? 3 : 1) * K * 64;  // [4][64] reduction scratch
  const int tz = tile_z(a);
  const int c0 = blockIdx.x * 64, t0 = tz * TT;
  const int c = threadIdx.x & 63, tg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: frame indices and chunk tests stay wave-uniform)
  const int tb = t0 + tg * FPT;
  const int chunk = a.chunk;
  stage_weights<K>(a, c0, s_wc, s_wk, s_le, s_re,
                   EDGE && (GEN || t0 - K / 2 < K || t0 + TT + K / 2 > a.T - K));
  float pwc[Kh], pwk[K];
#pragma unroll
  for (int j = 0; j < Kh; ++j) pwc[j] = 0.f;
#pragma unroll
  for (int j = 0; j < K; ++j) pwk[j] = 0.f;
  float pbc = 0.f, pbk = 0.f;
  const bool chan_ok = c0 + c < a.C;
  int cs[FPT];
  float sc[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    cs[i] = GEN ? ((tb + i) / chunk) * chunk : 0;
    sc[i] = 1.f;
  }
  __syncthreads();
  if (EDGE && a.scale) {
#pragma unroll
    for (int i = 0; i < FPT; ++i)
      if (tb + i < a.T) sc[i] = edge_scale(s_le, s_re, c, tb + i - cs[i], chunk, K);
  }
  // d(edge scale)[c, pos] = sum dy * (unscaled chunkwise conv output) over the frames within K of
  // a chunk edge: only tiles that contain such a frame compute it (block-uniform), summed over
  // this block's utterances in registers, one atomic per (frame, side) at the end
  bool edge = false;
  if (EDGE && dscale != nullptr) {
    const int p0 = t0 % chunk;
    edge = (p0 < K) || (p0 + TT - 1 >= chunk - K) || (p0 + TT > chunk);
  }
  const float bkv = (a.bk && chan_ok) ? a.bk[c0 + c] : 0.f;
  float ds[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) ds[i] = 0.f;
  const int b_end = min(a.B, (int)(blockIdx.y + 1) * BB);
  for (int b = blockIdx.y * BB; b < b_end; ++b) {
    __syncthreads();
    // this thread's 16 gradient values travel with the tile's loads (one round trip, not two)
    float gpre[FPT];
#pragma unroll
    for (int i = 0; i < FPT; ++i)
      gpre[i] = dy[((long)min(tb + i, a.T - 1) * a.B + b) * a.C + (chan_ok ? c0 + c : 0)];
    if (buf_ok(a)) stage_xg_buf<K>(a, b, t0, c0, s_x);
    else stage_xg<K>(a, b, t0, c0, s_x);
    __syncthreads();
    if (!chan_ok) continue;
    float win[W];
#pragma unroll
    for (int w = 0; w < W; ++w) win[w] = s_x[(tg * FPT + w) * 64 + c];
    if constexpr (GEN && SB > 0) {
      // causal-conv taps and the bias sums first (they see the unmasked window) ...
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const float g = (tb + i < a.T) ? gpre[i] : 0.f;
        pbc += g;
        pbk += g * sc[i];
        if (a.wc) {
#pragma unroll
          for (int j = 0; j < Kh; ++j) pwc[j] = fmaf(g, win[i + j], pwc[j]);
        }
      }
      // ... then, sub-block by sub-block, the window with everything outside the sub-block's chunk
      // zeroed (wave-uniform tests: tb, cs are scalar)
#pragma unroll
      for (int sb = 0; sb < FPT / SB; ++sb) {
        constexpr int WM = SB + K - 1;
        const int i0 = sb * SB, lo = cs[i0] - (tb - halo + i0);   // window index of the chunk's first frame
        float wm[WM];
#pragma unroll
        for (int w = 0; w < WM; ++w) wm[w] = (w >= lo && w < lo + chunk) ? win[i0 + w] : 0.f;
        if (edge) {
          float ak[SB];
#pragma unroll
          for (int i = 0; i < SB; ++i) ak[i] = bkv;
#pragma unroll
          for (int j = 0; j < K; ++j) {
            const float w = s_wk[j * 64 + c];                 // one LDS read per tap, SB uses
#pragma unroll
            for (int i = 0; i < SB; ++i) ak[i] = fmaf(w, wm[i + j], ak[i]);
          }
#pragma unroll
          for (int i = 0; i < SB; ++i)
            if (tb + i0 + i < a.T) ds[i0 + i] = fmaf(gpre[i0 + i], ak[i], ds[i0 + i]);
        }
#pragma unroll
        for (int i = 0; i < SB; ++i) {
          const float gs = ((tb + i0 + i < a.T) ? gpre[i0 + i] : 0.f) * sc[i0 + i];
#pragma unroll
          for (int j = 0; j < K; ++j) pwk[j] = fmaf(gs, wm[i + j], pwk[j]);
        }
      }
    } else {
    if (edge) {
      float ak[FPT];
#pragma unroll
      for (int i = 0; i < FPT; ++i) ak[i] = bkv;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const float w = s_wk[j * 64 + c];
#pragma unroll
        for (int i = 0; i < FPT; ++i) {
          const int tt = tb + i - halo + j;
          if (!GEN || (tt >= cs[i] && tt < cs[i] + chunk)) ak[i] = fmaf(w, win[i + j], ak[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < FPT; ++i)
        if (tb + i < a.T) ds[i] = fmaf(gpre[i], ak[i], ds[i]);
    }
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      const int t = tb + i;
      const float g = (t < a.T) ? gpre[i] : 0.f;
      const float gs = g * sc[i];
      pbc += g;
      pbk += gs;
      if (a.wc) {
#pragma unroll
        for (int j = 0; j < Kh; ++j) pwc[j] = fmaf(g, win[i + j], pwc[j]);
      }
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (!GEN) {
          pwk[j] = fmaf(gs, win[i + j], pwk[j]);
        } else {
          const int tt = t - halo + j;
          if (tt >= cs[i] && tt < cs[i] + chunk) pwk[j] = fmaf(gs, win[i + j], pwk[j]);
        }
      }
    }
    }
  }
  if (edge && chan_ok) {
#pragma unroll
    for (int i = 0; i < FPT; ++i) {
      const int t = tb + i;
      if (t >= a.T) continue;
      const int pos = t - cs[i], idx = pos - chunk + K;
      if (dscale_t) {
        // chunked launches (every tile has edge frames): into a [side][pos][channel] scratch, so
        // that the 64 lanes of an atomic instruction add to 64 CONSECUTIVE words -- in the
        // parameter's own (channel, pos) layout they are K words apart, one L2 transaction per
        // lane, and that was 105 of the launch's 230 us; zipconv_dscale_commit_kernel folds the
        // scratch into the parameter gradient
        if (pos < K) atomicAdd(&dscale_t[(long)pos * cpad + c0 + c], ds[i]);
        if (idx >= 0 && idx < K) atomicAdd(&dscale_t[((long)K + idx) * cpad + c0 + c], ds[i]);
      } else {
        if (pos < K) atomicAdd(&dscale[(long)(c0 + c) * K + pos], ds[i]);
        if (idx >= 0 && idx < K) atomicAdd(&dscale[((long)a.C + c0 + c) * K + idx], ds[i]);
      }
    }
  }
  // reduce the 4 frame groups of each channel through LDS, then one store per value
  const long blk = ((long)blockIdx.x * gridDim.y + blockIdx.y) * a.nt + tz;
  float* dst = part + blk * NV * 64 + c;          // [block][slot][64 channels]: coalesced stores
  auto reduce_store = [&](float v, int slot) {
    __syncthreads();
    s_red[tg * 64 + c] = v;
    __syncthreads();
    if (tg == 0) dst[slot * 64] = s_red[c] + s_red[64 + c] + s_red[128 + c] + s_red[192 + c];
  };
#pragma unroll
  for (int j = 0; j < Kh; ++j) reduce_store(pwc[j], j);
  reduce_store(pbc, Kh);
#pragma unroll
  for (int j = 0; j < K; ++j) reduce_store(pwk[j], Kh + 1 + j);
  reduce_store(pbk, Kh + 1 + K);
}

// out[c][slot] = sum over the (t-tile, b-group) blocks of part.  Grid (c tiles, slots); thread =
// (channel, block group): every read is a 256-byte row of 64 channels.
__global__ __launch_bounds__(256) void zipconv_reduce_w_kernel(
    const float* __restrict__ part, int nblk_per_ctile, int C, int Kh, int K,
    float* __restrict__ dwc, float* __restrict__ dbc, float* __restrict__ dwk,
    float* __restrict__ dbk) {
  __shared__ float s_red[4][64];
  const int NV = Kh + K + 2;
  const int ct = blockIdx.x, slot = blockIdx.y;
  const int c = threadIdx.x & 63, kg = threadIdx.x >> 6;
  float s = 0.f;
  const float* base = part + ((long)ct * nblk_per_ctile * NV + slot) * 64 + c;
  const long stride = (long)NV * 64;
  int k = kg;
  for (; k + 28 < nblk_per_ctile; k += 32) {           // 8 loads in flight per thread
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = base[(k + 4 * u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; k < nblk_per_ctile; k += 4) s += base[k * stride];
  s_red[kg][c] = s;
  __syncthreads();
  const int cg = ct * 64 + c;
  if (kg == 0 && cg < C) {
    s = s_red[0][c] + s_red[1][c] + s_red[2][c] + s_red[3][c];
    if (slot < Kh) {
      if (dwc) dwc[(long)cg * Kh + slot] += s;
    } else if (slot == Kh) {
      if (dbc) dbc[cg] += s;
    } else if (slot < Kh + 1 + K) {
      dwk[(long)cg * K + slot - Kh - 1] += s;
    } else if (dbk) {
      dbk[cg] += s;
    }
  }
}

template <int K>
size_t conv_smem(bool with_red, int tiles, bool edge) {
  constexpr int Kh = (K + 1) / 2, halo = K / 2;
  return sizeof(float) * ((edge ? tiles : 1) * (TT + 2 * halo) * 64 + Kh * 64 + (edge ? 3 : 1) * K * 64 +
                          (with_red ? 256 : 0));
}

// t-tiles [n_lo, n_hi) are interior (single chunk: no frame of the tile or its halo within K of
// either end of the sequence)
inline void interior_tiles(int T, int K, int nt, int* n_lo, int* n_hi) {
  int lo = 0, hi = nt;
  while (lo < nt && lo * TT - K / 2 < K) ++lo;
  while (hi > lo && (hi - 1) * TT + TT + K / 2 > T - K) --hi;
  *n_lo = lo;
  *n_hi = hi;
}

}  // namespace

#define S2T_CONV_DISPATCH(K, BODY)  \
  switch (K) {                      \
    case 3: { constexpr int KK = 3; BODY; } break;   \
    case 5: { constexpr int KK = 5; BODY; } break;   \
    case 7: { constexpr int KK = 7; BODY; } break;   \
    case 15: { constexpr int KK = 15; BODY; } break; \
    case 31: { constexpr int KK = 31; BODY; } break; \
    default: return -1;             \
  }

static int conv_args_ok(int T, int B, int C, int K, int chunk) {
  return T > 0 && B > 0 && C > 0 && (K & 1) && chunk > 0;
}

// One launch: the EDGE kernel over all tiles, or -- a plain depthwise Conv1d, nothing to scale -- the
// lean one (EDGE = false: no edge arrays in LDS).
struct ZSplit {
  int nt, n_lo, n_hi;
  ZSplit(int T, int K, bool gen, bool has_scale) {
    nt = (T + TT - 1) / TT;
    if (gen) { n_lo = n_hi = 0; }                    // everything through the EDGE kernel
    else if (!has_scale) { n_lo = 0; n_hi = nt; }    // nothing to scale: everything interior
    else if (getenv("S2T_CONV_SPLIT")) interior_tiles(T, K, nt, &n_lo, &n_hi);
    else { n_lo = n_hi = 0; }
    // (measured at the C3 shapes: the interior kernel alone over ALL tiles is 20-40 % faster than
    // the edge kernel, but two launches -- each with its own tail -- are 30-45 % slower than one;
    // the split is kept for experiments: S2T_CONV_SPLIT=1)
  }
  int n_int() const { return n_hi - n_lo; }
  int n_edge() const { return nt - (n_hi - n_lo); }
  void set_int(ConvArgs& a) const { a.z_base = n_lo; a.z_split = 1 << 30; a.z_jump = 0; a.nt = nt; }
  void set_edge(ConvArgs& a) const { a.z_base = 0; a.z_split = n_lo; a.z_jump = n_hi - n_lo; a.nt = nt; }
};

// sub-block size of the chunked (GEN) kernels' masked-window form: min(chunk, FPT) when chunk is a
// power of two (every chunk_size / downsampling pair of the YAMLs), 0 = the per-tap test
static int conv_subblock(int chunk) {
  static const bool on = [] { const char* e = getenv("S2T_CONV_SUBBLOCK"); return !e || atoi(e) != 0; }();
  return (on && chunk >= 2 && (chunk & (chunk - 1)) == 0) ? std::min(chunk, FPT) : 0;
}
#define S2T_CONV_GEN_SB(SBSEL, LAUNCH) \
  switch (SBSEL) {                     \
    case 16: { constexpr int SBV = 16; LAUNCH; } break; \
    case 8: { constexpr int SBV = 8; LAUNCH; } break;   \
    case 4: { constexpr int SBV = 4; LAUNCH; } break;   \
    case 2: { constexpr int SBV = 2; LAUNCH; } break;   \
    default: { constexpr int SBV = 0; LAUNCH; } break;  \
  }

static int zipconv_fwd_impl(const float* u, long ld, int gate_off, const unsigned char* mask,
                            int T, int B, int C, int K, int chunk, const float* wc,
                            const float* bc, const float* wk, const float* bk,
                            const float* scale, float* y, float* y2, float act_off, float act_c,
                            void* stream) {
  if (!conv_args_ok(T, B, C, K, chunk)) return -1;
  ConvArgs a{u, ld, gate_off, mask, T, B, C, chunk, wc, bc, wk, bk, scale, 0, 1 << 30, 0, 0};
  hipStream_t st = (hipStream_t)stream;
  const bool gen = chunk < T;
  const ZSplit z(T, K, gen, scale != nullptr);
  const unsigned gx = (C + 63) / 64;
  if (z.n_int() > 0) {
    z.set_int(a);
    S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_fwd_kernel<KK, false, false>), dim3(gx, B, z.n_int()),
                                            dim3(256), conv_smem<KK>(false, 1, false), st, a, y, y2, act_off, act_c));
    S2T_CHECK_LAUNCH();
  }
  if (z.n_edge() > 0) {
    z.set_edge(a);
    const dim3 grid(gx, B, z.n_edge());
    if (!gen) {
      S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_fwd_kernel<KK, false, true>), grid, dim3(256),
                                              conv_smem<KK>(false, 1, true), st, a, y, y2, act_off, act_c));
    } else {
      S2T_CONV_GEN_SB(conv_subblock(chunk),
                      S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_fwd_kernel<KK, true, true, SBV>), grid,
                                                              dim3(256), conv_smem<KK>(false, 1, true), st, a, y,
                                                              y2, act_off, act_c)));
    }
    S2T_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int s2t_zipconv_fwd(const float* u, long ld, int gate_off, const unsigned char* mask,
                               int T, int B, int C, int K, int chunk, const float* wc,
                               const float* bc, const float* wk, const float* bk,
                               const float* scale, float* y, void* stream) {
  return zipconv_fwd_impl(u, ld, gate_off, mask, T, B, C, K, chunk, wc, bc, wk, bk, scale, y, nullptr, 0.f, 0.f,
                          stream);
}

// the same pass with act(y) as a second output: act_kind 1 = SwooshL, 2 = SwooshR (the conv module's
// activation between the depthwise conv and out_proj, model/encoder/zipformer.py:2700-2703)
extern "C" int s2t_zipconv_fwd_act(const float* u, long ld, int gate_off, const unsigned char* mask,
                                   int T, int B, int C, int K, int chunk, const float* wc,
                                   const float* bc, const float* wk, const float* bk,
                                   const float* scale, float* y, float* y_act, int act_kind,
                                   void* stream) {
  if (!y_act || (act_kind != 1 && act_kind != 2)) return -1;
  return zipconv_fwd_impl(u, ld, gate_off, mask, T, B, C, K, chunk, wc, bc, wk, bk, scale, y, y_act,
                          act_kind == 1 ? 4.0f : 1.0f, act_kind == 1 ? 0.035f : 0.313261687f, stream);
}

// dscale (2, C, K) += scratch [2][K][cpad] (see zipconv_bwd_w_kernel)
__global__ __launch_bounds__(256) void zipconv_dscale_commit_kernel(const float* __restrict__ t, int C,
                                                                    int K, int cpad,
                                                                    float* __restrict__ dscale) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * C * K) return;
  const int side = i / (C * K), c = (i / K) % C, p = i % K;
  const float v = t[((long)side * K + p) * cpad + c];
  if (v != 0.f) dscale[i] += v;
}

// floats of scratch s2t_zipconv_bwd needs: the per-block partial sums + the transposed edge-scale
// accumulator of the chunked launches
static long zipconv_part_floats(int T, int B, int C, int K) {
  const long tiles = (long)((T + TT - 1) / TT) * ((C + 63) / 64);
  return tiles * B * 64 * ((K + 1) / 2 + K + 2);
}
extern "C" long s2t_zipconv_bwd_workspace_floats(int T, int B, int C, int K) {
  return zipconv_part_floats(T, B, C, K) + 2L * K * ((C + 63) / 64) * 64;
}

// the data-gradient kernels of a backward call on stream st
static int zipconv_bwd_launch_data(ConvArgs a, int T, int B, int C, int K, bool gen, const ZSplit& z,
                                   const float* dy, float* du, hipStream_t st) {
  const unsigned gx = (C + 63) / 64;
  if (z.n_int() > 0) {
    z.set_int(a);
    S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_bwd_data_kernel<KK, false, false>),
                                            dim3(gx, B, z.n_int()), dim3(256),
                                            conv_smem<KK>(false, 1, false), st, a, dy, du));
    S2T_CHECK_LAUNCH();
  }
  if (z.n_edge() > 0) {
    z.set_edge(a);
    const dim3 grid(gx, B, z.n_edge());
    if (!gen) {
      S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_bwd_data_kernel<KK, false, true>), grid, dim3(256),
                                              conv_smem<KK>(false, 1, true), st, a, dy, du));
    } else {
      S2T_CONV_GEN_SB(conv_subblock(a.chunk),
                      S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_bwd_data_kernel<KK, true, true, SBV>), grid,
                                                              dim3(256), conv_smem<KK>(false, 1, true), st, a, dy,
                                                              du)));
    }
    S2T_CHECK_LAUNCH();
  }
  return 0;
}

// the parameter-gradient kernels (taps, biases, edge scales) + their reduction on stream st
static int zipconv_bwd_launch_params(ConvArgs a, int T, int B, int C, int K, bool gen, const ZSplit& z,
                                     const float* wc, const float* scale, const float* dy, float* dwc,
                                     float* dbc, float* dwk, float* dbk, float* dscale,
                                     float* workspace, hipStream_t st) {
  const unsigned gx = (C + 63) / 64;
  // utterances per block: many -- a workgroup's fixed costs (tap staging, the 49-slot reduction
  // epilogue) are what this kernel's time is made of: at the C3 shapes 768 / 384 / 192 / 128
  // workgroups take 92 / 66 / 56 / 62 us (K = 31) and 46 / 46 / 33 / 27 us (K = 15)
  const long tiles = (long)gx * z.nt;
  const char* env = getenv("S2T_CONV_BLOCKS");         // tuning / tests: workgroup-count target
  int BB = (int)((tiles * B) / (env ? std::max(1, atoi(env)) : (K >= 31 ? 192 : 128)));
  if (BB < 1) BB = 1;
  if (BB > 16) BB = 16;
  const unsigned gy = (B + BB - 1) / BB;             // utterance groups
  float* const dsc = (scale && dscale) ? dscale : nullptr;
  if (z.n_int() > 0) {
    z.set_int(a);
    S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_bwd_w_kernel<KK, false, false>), dim3(gx, gy, z.n_int()),
                                            dim3(256), conv_smem<KK>(true, 1, false), st, a, dy, BB,
                                            workspace, dsc));
    S2T_CHECK_LAUNCH();
  }
  if (z.n_edge() > 0) {
    z.set_edge(a);
    const dim3 gridw(gx, gy, z.n_edge());
    if (!gen) {
      S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_bwd_w_kernel<KK, false, true>), gridw, dim3(256),
                                              conv_smem<KK>(true, 1, true), st, a, dy, BB, workspace, dsc));
    } else {
      const int cpad = (int)gx * 64;
      float* dst = dsc ? workspace + zipconv_part_floats(T, B, C, K) : nullptr;
      if (dst && hipMemsetAsync(dst, 0, sizeof(float) * 2 * K * cpad, st) != hipSuccess) return -3;
      S2T_CONV_GEN_SB(conv_subblock(a.chunk),
                      S2T_CONV_DISPATCH(K, hipLaunchKernelGGL((zipconv_bwd_w_kernel<KK, true, true, SBV>), gridw,
                                                              dim3(256), conv_smem<KK>(true, 1, true), st, a, dy, BB,
                                                              workspace, dsc, dst, cpad)));
      S2T_CHECK_LAUNCH();
      if (dst) hipLaunchKernelGGL(zipconv_dscale_commit_kernel, dim3((2 * C * K + 255) / 256), dim3(256), 0, st,
                                  dst, C, K, cpad, dsc);
    }
    S2T_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(zipconv_reduce_w_kernel, dim3(gx, (K + 1) / 2 + K + 2), dim3(256), 0,
                     st, workspace, (int)(z.nt * gy), C, (K + 1) / 2, K, wc ? dwc : nullptr,
                     wc ? dbc : nullptr, dwk, dbk);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_zipconv_bwd(const float* u, long ld, int gate_off, const unsigned char* mask,
                               int T, int B, int C, int K, int chunk, const float* wc,
                               const float* wk, const float* bk, const float* scale,
                               const float* dy, float* du, float* dwc, float* dbc, float* dwk,
                               float* dbk, float* dscale, float* workspace, void* stream) {
  if (!conv_args_ok(T, B, C, K, chunk)) return -1;
  hipStream_t st = (hipStream_t)stream;
  ConvArgs a{u, ld, gate_off, mask, T, B, C, chunk, wc, nullptr, wk, bk, scale, 0, 1 << 30, 0, 0};
  const bool gen = chunk < T;
  const ZSplit z(T, K, gen, scale != nullptr);
  const unsigned gx = (C + 63) / 64;
  // (a one-kernel form of both gradients was built in round 3 and measured SLOWER -- 137 / 105 / 82 /
  // 49 us against 112 / 89 / 49 / 44 at T = 495 / 248 / 124 / 62: the union of the two bodies needs
  // 358 - 510 registers = one workgroup per CU -- and removed in round 5)
  int rc = zipconv_bwd_launch_data(a, T, B, C, K, gen, z, dy, du, st);
  if (rc != 0) return rc;
  return zipconv_bwd_launch_params(a, T, B, C, K, gen, z, wc, scale, dy, dwc, dbc, dwk, dbk, dscale,
                                   workspace, st);
}

// The two halves as separate calls: the parameter gradients only feed the optimizer, so a caller
// may run them on another stream (ordered after the producers of u / dy; u, dy and workspace
// alive until that stream is joined) while the data gradient stays on the chain.
extern "C" int s2t_zipconv_bwd_data(const float* u, long ld, int gate_off, const unsigned char* mask,
                                    int T, int B, int C, int K, int chunk, const float* wc,
                                    const float* wk, const float* bk, const float* scale,
                                    const float* dy, float* du, void* stream) {
  if (!conv_args_ok(T, B, C, K, chunk)) return -1;
  ConvArgs a{u, ld, gate_off, mask, T, B, C, chunk, wc, nullptr, wk, bk, scale, 0, 1 << 30, 0, 0};
  const bool gen = chunk < T;
  const ZSplit z(T, K, gen, scale != nullptr);
  return zipconv_bwd_launch_data(a, T, B, C, K, gen, z, dy, du, (hipStream_t)stream);
}

extern "C" int s2t_zipconv_bwd_params(const float* u, long ld, int gate_off,
                                      const unsigned char* mask, int T, int B, int C, int K,
                                      int chunk, const float* wc, const float* wk, const float* bk,
                                      const float* scale, const float* dy, float* dwc, float* dbc,
                                      float* dwk, float* dbk, float* dscale, float* workspace,
                                      void* stream) {
  if (!conv_args_ok(T, B, C, K, chunk)) return -1;
  ConvArgs a{u, ld, gate_off, mask, T, B, C, chunk, wc, nullptr, wk, bk, scale, 0, 1 << 30, 0, 0};
  const bool gen = chunk < T;
  const ZSplit z(T, K, gen, scale != nullptr);
  return zipconv_bwd_launch_params(a, T, B, C, K, gen, z, wc, scale, dy, dwc, dbc, dwk, dbk, dscale,
                                   workspace, (hipStream_t)stream);
}
