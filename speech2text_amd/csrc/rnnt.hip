// RNN-T lattice kernels for gfx950: "simple" (am+lm) log-probs, the
// mutual-information (forward/backward) recursion, prune-range selection,
// the fused pruned joiner (never materialises (B,T,R,C)), and the full
// (unpruned) lattice log-probs.
//
// Reference call sites: model/joiner/joiner.py:100-123 (k2.rnnt_loss_smoothed,
// k2.get_rnnt_prune_ranges, k2.do_rnnt_pruning), model/joiner/joiner.py:176-178
// (add + activation), model/loss/pruned_rnnt_loss.py:39-48 (k2.rnnt_loss_pruned),
// model/loss/rnnt_loss.py:42-44 (torchaudio RNNTLoss).  k2 (v1.24.3) and
// torchaudio (0.13.1) are not vendored by the reference: semantics restated in
// oracle/k2_rnnt.py, layout notes in DESIGN.md.
//
// Layouts (all fp32, int64 indices, as k2):
//   am [B][T][C], lm [B][S+1][C], symbols [B][S], boundary [B][4]=(0,0,S_b,T_b)
//   px [B][S][T+1]   log-prob of emitting symbol s at frame t   (col T = -inf)
//   py [B][S+1][T]   log-prob of blank at (s,t)
//   p  [B][S+1][T+1] forward scores of the recursion
// The recursion is a wavefront over anti-diagonals: lane/thread = s, one
// workgroup per utterance, neighbour exchange through LDS (or DPP shuffles
// when S+1 <= 64), the next diagonal's px/py prefetched ahead of the barrier.
#include "common.h"

namespace {

// ---------------------------------------------------------------- row exp
// probs[r][c] = exp(x[r][c] - max_c x[r][:]); one wave per row.
__global__ __launch_bounds__(256) void row_exp_kernel(const float* __restrict__ x, long rows, int C,
                                                      float* __restrict__ probs,
                                                      float* __restrict__ rowmax) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = x + row * C;
  float m = S2T_NEG_INF;
  for (int c = lane; c < C; c += 64) m = fmaxf(m, p[c]);
  m = wave_max(m);
  float* q = probs + row * C;
  for (int c = lane; c < C; c += 64) q[c] = expf(p[c] - m);
  if (lane == 0) rowmax[row] = m;
}

// ------------------------------------------------------- simple px / py
// nrm = lm_probs @ am_probs^T  [B][S1][T]  (linear domain, from rocBLAS)
__global__ __launch_bounds__(256) void simple_pxpy_kernel(
    const float* __restrict__ am, const float* __restrict__ lm, const float* __restrict__ am_max,
    const float* __restrict__ lm_max, const float* __restrict__ nrm,
    const long* __restrict__ symbols, const long* __restrict__ boundary, int S, int T, int C,
    int blank, float* __restrict__ px, float* __restrict__ py) {
  const int b = blockIdx.z, s = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int S1 = S + 1;
  if (t > T) return;
  const long Tb = boundary[b * 4 + 3];
  if (t == T) {
    if (s < S) px[((long)b * S + s) * (T + 1) + T] = S2T_NEG_INF;
    return;
  }
  const float tiny = 1.17549435e-38f;
  const float n = nrm[((long)b * S1 + s) * T + t];
  const float z = logf(n + tiny) + lm_max[b * S1 + s] + am_max[b * T + t];
  const float* amr = am + ((long)b * T + t) * C;
  const float* lmr = lm + ((long)b * S1 + s) * C;
  py[((long)b * S1 + s) * T + t] = amr[blank] + lmr[blank] - z;
  if (s < S) {
    const long y = symbols[(long)b * S + s];
    float v = amr[y] + lmr[y] - z;
    if (t == Tb) v = S2T_NEG_INF;  // fix_for_boundary
    px[((long)b * S + s) * (T + 1) + t] = v;
  }
}

// W[b][s][t] = -(dpx[b][s][t] + dpy[b][s][t]) / (nrm + tiny)
__global__ __launch_bounds__(256) void simple_w_kernel(const float* __restrict__ dpx,
                                                       const float* __restrict__ dpy,
                                                       const float* __restrict__ nrm,
                                                       const float* __restrict__ gscale, int S,
                                                       int T, float* __restrict__ W) {
  const int b = blockIdx.z, s = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int S1 = S + 1;
  const float g = gscale[b];
  float d = dpy[((long)b * S1 + s) * T + t];
  if (s < S) d += dpx[((long)b * S + s) * (T + 1) + t];
  const float tiny = 1.17549435e-38f;
  W[((long)b * S1 + s) * T + t] = -(d * g) / (nrm[((long)b * S1 + s) * T + t] + tiny);
}

// d_am[b][t][c] = am_probs * G_am (+ blank / symbol gather terms); one wave per
// (b,t) row, the row is assembled in LDS so the gather terms can be added there.
__global__ __launch_bounds__(256) void simple_dam_kernel(
    const float* __restrict__ am_probs, const float* __restrict__ G, const float* __restrict__ dpx,
    const float* __restrict__ dpy, const float* __restrict__ gscale,
    const long* __restrict__ symbols, int B, int S, int T, int C, int blank,
    float* __restrict__ d_am, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* rowbuf = reinterpret_cast<float*>(smem_raw) + (threadIdx.x >> 6) * C;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const bool ok = row < (long)B * T;
  const int b = ok ? (int)(row / T) : 0, t = ok ? (int)(row % T) : 0;
  const float g = gscale[b];
  const int S1 = S + 1;
  if (ok)
    for (int c = lane; c < C; c += 64) rowbuf[c] = am_probs[row * C + c] * G[row * C + c];
  __syncthreads();
  if (ok) {
    float sb = 0.f;
    for (int s = lane; s < S1; s += 64) sb += dpy[((long)b * S1 + s) * T + t];
    sb = wave_sum(sb) * g;
    if (lane == 0) atomicAdd(&rowbuf[blank], sb);
    for (int s = lane; s < S; s += 64) {
      const float v = dpx[((long)b * S + s) * (T + 1) + t];
      if (v != 0.f) atomicAdd(&rowbuf[symbols[(long)b * S + s]], v * g);
    }
  }
  __syncthreads();
  if (ok) {
    float* o = d_am + row * C;
    for (int c = lane; c < C; c += 64) o[c] = accumulate ? o[c] + rowbuf[c] : rowbuf[c];
  }
}

// d_lm[b][s][c] = lm_probs * G_lm (+ row sums);  one wave per (b,s) row
__global__ __launch_bounds__(256) void simple_dlm_kernel(
    const float* __restrict__ lm_probs, const float* __restrict__ G, const float* __restrict__ dpx,
    const float* __restrict__ dpy, const float* __restrict__ gscale,
    const long* __restrict__ symbols, int B, int S, int T, int C, int blank,
    float* __restrict__ d_lm, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* rowbuf = reinterpret_cast<float*>(smem_raw) + (threadIdx.x >> 6) * C;
  const int S1 = S + 1;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const bool ok = row < (long)B * S1;
  const int b = ok ? (int)(row / S1) : 0, s = ok ? (int)(row % S1) : 0;
  const float g = gscale[b];
  if (ok)
    for (int c = lane; c < C; c += 64) rowbuf[c] = lm_probs[row * C + c] * G[row * C + c];
  __syncthreads();
  if (ok) {
    float sy = 0.f, sx = 0.f;
    for (int t = lane; t < T; t += 64) sy += dpy[((long)b * S1 + s) * T + t];
    if (s < S)
      for (int t = lane; t <= T; t += 64) sx += dpx[((long)b * S + s) * (T + 1) + t];
    sy = wave_sum(sy) * g;
    sx = wave_sum(sx) * g;
    if (lane == 0) {
      rowbuf[blank] += sy;
      if (s < S) rowbuf[symbols[(long)b * S + s]] += sx;
    }
  }
  __syncthreads();
  if (ok) {
    float* o = d_lm + row * C;
    for (int c = lane; c < C; c += 64) o[c] = accumulate ? o[c] + rowbuf[c] : rowbuf[c];
  }
}

// --------------------------------------------- mutual information recursion
constexpr int MI_PD = 8;   // diagonals of operand look-ahead

// Operands and results move through raw buffer accesses on STRAIGHT-LINE code: a lane outside the
// lattice passes an out-of-range offset (the load returns 0, the store is dropped) instead of
// branching around the access.  With the accesses inside `if (in the lattice)` the compiler closed
// every diagonal with s_waitcnt vmcnt(0) -- 38 of them in the forward kernel -- so a step waited
// for the loads it had just issued for eight diagonals later, and the look-ahead bought nothing
// (134 us for 298 diagonals, an L2 round trip each).
constexpr unsigned MI_OOB = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mi_rsrc(const float* p, long floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}
__device__ __forceinline__ float mi_load(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
__device__ __forceinline__ void mi_store(__amdgpu_buffer_rsrc_t rs, unsigned off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, 0);
}

__global__ void mi_fwd_kernel(const float* __restrict__ px, const float* __restrict__ py,
                              const long* __restrict__ boundary, int S, int T,
                              float* __restrict__ p, float* __restrict__ ans) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* ex = reinterpret_cast<float*>(smem_raw);  // [2][blockDim.x + 1]
  const int b = blockIdx.x, s = threadIdx.x, nt = blockDim.x;
  const int Sb = (int)boundary[b * 4 + 2], Tb = (int)boundary[b * 4 + 3];
  const __amdgpu_buffer_rsrc_t rpx = mi_rsrc(px + (long)b * S * (T + 1), (long)S * (T + 1));
  const __amdgpu_buffer_rsrc_t rpy = mi_rsrc(py + (long)b * (S + 1) * T, (long)(S + 1) * T);
  const __amdgpu_buffer_rsrc_t rp = mi_rsrc(p + (long)b * (S + 1) * (T + 1), (long)(S + 1) * (T + 1));
  float own = S2T_NEG_INF;  // p[s][t-1]
  ex[s] = S2T_NEG_INF;
  ex[nt + 1 + s] = S2T_NEG_INF;
  if (s == 0) {
    ex[nt] = S2T_NEG_INF;
    ex[2 * nt + 1] = S2T_NEG_INF;
  }
  __syncthreads();
  const bool act = s <= Sb;
  // Operands of diagonal d (px[s-1][t], py[s][t-1], t = d - s) are requested MI_PD diagonals
  // ahead: a step of the recursion is a few dozen cycles, a global round trip ~1 us.  The raw
  // values are kept (0 where the cell has no such operand); validity is re-derived at use.
  float rx[MI_PD], ry[MI_PD];
#define MI_FETCH(DD, X, Y)                                                                   \
  {                                                                                          \
    const int t1_ = (DD) - s;                                                                \
    const bool in_ = act && t1_ >= 0 && t1_ <= Tb;                                           \
    X = mi_load(rpx, (in_ && s > 0) ? (unsigned)((s - 1) * (T + 1) + t1_) * 4u : MI_OOB);    \
    Y = mi_load(rpy, (in_ && t1_ > 0) ? (unsigned)(s * T + t1_ - 1) * 4u : MI_OOB);          \
  }
#pragma unroll
  for (int u = 0; u < MI_PD; ++u) MI_FETCH(u, rx[u], ry[u])
  int cur = 0;
  const int D = Sb + Tb;
  // whole groups of MI_PD diagonals: the ones past D have no cell in the lattice (t > Tb for
  // every s <= Sb) and fall through as no-ops -- no exit test inside the unrolled body
  for (int d0 = 0; d0 <= D; d0 += MI_PD) {
#pragma unroll
    for (int u = 0; u < MI_PD; ++u) {
      const int d = d0 + u;
      const int t = d - s;
      const float vx = rx[u], vy = ry[u];
      MI_FETCH(d + MI_PD, rx[u], ry[u])
      const bool in = act && t >= 0 && t <= Tb;
      const float up = (s > 0) ? ex[(cur ^ 1) * (nt + 1) + (s > 0 ? s - 1 : 0)] + vx : S2T_NEG_INF;
      const float left = (t > 0) ? own + vy : S2T_NEG_INF;
      const float step = (d == 0) ? 0.f : log_add_comp(up, left);
      const float val = in ? step : S2T_NEG_INF;
      mi_store(rp, in ? (unsigned)(s * (T + 1) + t) * 4u : MI_OOB, val);
      own = in ? val : own;
      ex[cur * (nt + 1) + s] = val;
      __syncthreads();
      cur ^= 1;
    }
  }
#undef MI_FETCH
  if (s == Sb) ans[b] = own;
}

__global__ void mi_bwd_kernel(const float* __restrict__ px, const float* __restrict__ py,
                              const long* __restrict__ boundary, const float* __restrict__ p,
                              const float* __restrict__ ans_grad, int S, int T,
                              float* __restrict__ px_grad, float* __restrict__ py_grad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* exg = reinterpret_cast<float*>(smem_raw);  // [2][nt+1] p_grad
  const int b = blockIdx.x, s = threadIdx.x, nt = blockDim.x;
  float* exp_ = exg + 2 * (nt + 1);                 // [2][nt+1] p
  const int Sb = (int)boundary[b * 4 + 2], Tb = (int)boundary[b * 4 + 3];
  const __amdgpu_buffer_rsrc_t rpx = mi_rsrc(px + (long)b * S * (T + 1), (long)S * (T + 1));
  const __amdgpu_buffer_rsrc_t rpy = mi_rsrc(py + (long)b * (S + 1) * T, (long)(S + 1) * T);
  const __amdgpu_buffer_rsrc_t rp = mi_rsrc(p + (long)b * (S + 1) * (T + 1), (long)(S + 1) * (T + 1));
  const __amdgpu_buffer_rsrc_t rgx = mi_rsrc(px_grad + (long)b * S * (T + 1), (long)S * (T + 1));
  const __amdgpu_buffer_rsrc_t rgy = mi_rsrc(py_grad + (long)b * (S + 1) * T, (long)(S + 1) * T);
  for (int i = s; i < 2 * (nt + 1); i += nt) {
    exg[i] = 0.f;
    exp_[i] = S2T_NEG_INF;
  }
  __syncthreads();
  const bool act = s <= Sb;
  const float ag = ans_grad ? ans_grad[b] : 1.f;
  float own_g = 0.f, own_p = S2T_NEG_INF;  // p_grad[s][t+1], p[s][t+1]
  int cur = 0;
  const int D = Sb + Tb;
  // operands of diagonal d (p, px, py at (s, t = d - s)) are requested MI_PD diagonals ahead
  // (raw buffer loads, see above); diagonals below 0 are no-ops
  float rp_[MI_PD], rx[MI_PD], ry[MI_PD];
#define MI_FETCH(DD, VP, VX, VY)                                                             \
  {                                                                                          \
    const int t1_ = (DD) - s;                                                                \
    const bool in_ = act && (DD) >= 0 && t1_ >= 0 && t1_ <= Tb;                              \
    VP = mi_load(rp, in_ ? (unsigned)(s * (T + 1) + t1_) * 4u : MI_OOB);                     \
    VX = mi_load(rpx, (in_ && s < Sb) ? (unsigned)(s * (T + 1) + t1_) * 4u : MI_OOB);        \
    VY = mi_load(rpy, (in_ && t1_ < Tb) ? (unsigned)(s * T + t1_) * 4u : MI_OOB);            \
  }
#pragma unroll
  for (int u = 0; u < MI_PD; ++u) MI_FETCH(D - u, rp_[u], rx[u], ry[u])
  for (int d0 = D; d0 >= 0; d0 -= MI_PD) {
#pragma unroll
    for (int u = 0; u < MI_PD; ++u) {
      const int d = d0 - u;
      const int t = d - s;
      const float vp = rp_[u], vx = rx[u], vy = ry[u];
      MI_FETCH(d - MI_PD, rp_[u], rx[u], ry[u])
      const bool in = act && d >= 0 && t >= 0 && t <= Tb;
      const float pd = exp_[(cur ^ 1) * (nt + 1) + s + 1];  // p[s+1][t]
      const float gd = exg[(cur ^ 1) * (nt + 1) + s + 1];   // p_grad[s+1][t]
      float xg = 0.f, yg = 0.f;
      if (in && s < Sb && pd != S2T_NEG_INF && gd != 0.f) {
        const float e = __expf(vp + vx - pd);
        xg = (e == e) ? gd * e : 0.f;
      }
      if (in && t < Tb && own_p != S2T_NEG_INF && own_g != 0.f) {
        const float e = __expf(vp + vy - own_p);
        yg = (e == e) ? own_g * e : 0.f;
      }
      mi_store(rgx, (in && s < Sb && d != D) ? (unsigned)(s * (T + 1) + t) * 4u : MI_OOB, xg);
      mi_store(rgy, (in && t < Tb && d != D) ? (unsigned)(s * T + t) * 4u : MI_OOB, yg);
      const float g = in ? (d == D ? ag : xg + yg) : 0.f;
      own_g = in ? g : own_g;
      own_p = in ? vp : own_p;
      exg[cur * (nt + 1) + s] = g;
      exp_[cur * (nt + 1) + s] = in ? vp : S2T_NEG_INF;
      __syncthreads();
      cur ^= 1;
    }
  }
#undef MI_FETCH
}

// ------------------------------------------------------------ prune ranges
// v[t] <- min(v[t], v[t+1], ..., v[T-1]) for a 256-thread workgroup; v in LDS, cm = 256 longs of
// LDS scratch.  Every thread owns a contiguous chunk; the chunk minima are combined by a
// doubling scan.  All threads must call it; ends with a barrier.
__device__ __forceinline__ void block_suffix_min(long* v, int T, long* cm) {
  const long kMax = 0x7fffffffffffffffL;
  const int tid = threadIdx.x, per = (T + 255) / 256;
  const int lo = tid * per, hi = min(T, lo + per);
  long m = kMax;
  for (int t = hi - 1; t >= lo; --t) {
    m = v[t] < m ? v[t] : m;
    v[t] = m;
  }
  cm[tid] = m;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const long o = tid + off < 256 ? cm[tid + off] : kMax;
    __syncthreads();
    if (o < cm[tid]) cm[tid] = o;
    __syncthreads();
  }
  const long ex = tid + 1 < 256 ? cm[tid + 1] : kMax;
  for (int t = lo; t < hi; ++t)
    if (ex < v[t]) v[t] = ex;
  __syncthreads();
}

// k2.get_rnnt_prune_ranges + _adjust_pruning_lower_bound; one block per b.
__global__ __launch_bounds__(256) void prune_ranges_kernel(
    const float* __restrict__ px_grad, const float* __restrict__ py_grad,
    const long* __restrict__ boundary, int S, int T, int s_range, long* __restrict__ ranges) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ long s_cm[256];
  long* sb = reinterpret_cast<long*>(smem_raw);  // [T]
  const int b = blockIdx.x;
  const int S1 = S + 1;
  const int nblk = S1 - s_range + 1;
  const long Sb = boundary[b * 4 + 2], Tb = boundary[b * 4 + 3];
  const float* gx = px_grad + (long)b * S * (T + 1);
  const float* gy = py_grad + (long)b * S1 * T;
  long pad = Sb - s_range + 1;
  if (pad < 0) pad = 0;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    // sliding window sum over s of py_grad, minus px_grad_pad[s]
    float best = 0.f;  // s'=0: px_grad_pad = 0
    for (int i = 0; i < s_range; ++i) best += gy[(long)i * T + t];
    int arg = 0;
    for (int sp = 1; sp < nblk; ++sp) {
      // recompute the block sum in the reference's order (sum over i ascending)
      float w = 0.f;
      for (int i = 0; i < s_range; ++i) w += gy[(long)(sp + i) * T + t];
      const float v = w - gx[(long)(sp - 1) * (T + 1) + t];
      if (v > best) {
        best = v;
        arg = sp;
      }
    }
    sb[t] = (t < Tb - 1) ? (long)arg : pad;
  }
  __syncthreads();
  {
    // monotonic lower bound: reverse cummin, the same after the change of variable, clamp --
    // as workgroup-wide suffix minima (one thread walking the four passes cost ~60 us of LDS
    // round trips per launch)
    const long k = s_range - 1;
    block_suffix_min(sb, T, s_cm);
    for (int t = threadIdx.x; t < T; t += blockDim.x) sb[t] = -(sb[t] - k * t);
    __syncthreads();
    block_suffix_min(sb, T, s_cm);
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const long v = sb[t] < 0 ? 0 : sb[t];  // clamp(min=0): first frame starts at symbol 0
      sb[t] = -(v - k * t);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * s_range; i += blockDim.x) {
    const int t = i / s_range, r = i % s_range;
    ranges[((long)b * T + t) * s_range + r] = sb[t] + r;
  }
}

// ----------------------------------------------------- fused pruned joiner
__device__ __forceinline__ float act_fwd(float x, int act) {
  return act == 0 ? fmaxf(x, 0.f) : tanhf(x);
}
__device__ __forceinline__ float act_deriv(float pre, float out, int act) {
  return act == 0 ? (pre > 0.f ? 1.f : 0.f) : 1.f - out * out;
}

// One wave per (b,t): for each of the R pruned rows compute
// row = act(am[b,t,:] + lm[b,s,:]), Z = logsumexp(row), px~, py~ -> px,py.
// px/py must be pre-filled with -inf.  lse [B][T][R] saved for backward.
template <int MAXC_PER_LANE>
__global__ __launch_bounds__(256) void pruned_fwd_kernel(
    const float* __restrict__ am, const float* __restrict__ lm, const long* __restrict__ ranges,
    const long* __restrict__ symbols, const long* __restrict__ boundary, int B, int S, int T,
    int C, int R, int blank, int act, float* __restrict__ px, float* __restrict__ py,
    float* __restrict__ lse) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)B * T) return;
  const int b = (int)(row / T), t = (int)(row % T);
  const int S1 = S + 1;
  const long Tb = boundary[b * 4 + 3];
  float a[MAXC_PER_LANE];
  const float* amr = am + row * C;
#pragma unroll
  for (int j = 0; j < MAXC_PER_LANE; ++j) {
    const int c = lane + 64 * j;
    a[j] = c < C ? amr[c] : 0.f;
  }
  const long s0 = ranges[row * R];
  for (int i = 0; i < R; ++i) {
    const long s = s0 + i;
    if (s > S) break;
    const float* lmr = lm + ((long)b * S1 + s) * C;
    float v[MAXC_PER_LANE];
    float m = S2T_NEG_INF;
#pragma unroll
    for (int j = 0; j < MAXC_PER_LANE; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < C ? act_fwd(a[j] + lmr[c], act) : S2T_NEG_INF;
      m = fmaxf(m, v[j]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXC_PER_LANE; ++j) sum += (lane + 64 * j < C) ? expf(v[j] - m) : 0.f;
    sum = wave_sum(sum);
    const float z = m + logf(sum);
    if (lane == 0) lse[row * R + i] = z;
    // gather blank and symbol entries (recompute: cheap, avoids cross-lane index)
    if (lane == 0) {
      const float vb = act_fwd(amr[blank] + lmr[blank], act);
      py[((long)b * S1 + s) * T + t] = vb - z;
      if (s < S) {
        const long y = symbols[(long)b * S + s];
        float vy = act_fwd(amr[y] + lmr[y], act) - z;
        if (t == Tb) vy = S2T_NEG_INF;
        px[((long)b * S + s) * (T + 1) + t] = vy;
      }
    }
  }
}

// d_am[b,t,:] = sum_i act'(.) * ( dpx*(1[y]-P) + dpy*(1[blank]-P) ); wave per (b,t)
template <int MAXC_PER_LANE>
__global__ __launch_bounds__(256) void pruned_dam_kernel(
    const float* __restrict__ am, const float* __restrict__ lm, const long* __restrict__ ranges,
    const long* __restrict__ symbols, const float* __restrict__ lse,
    const float* __restrict__ dpx, const float* __restrict__ dpy, const float* __restrict__ gscale,
    int B, int S, int T, int C, int R, int blank, int act, float* __restrict__ d_am,
    int accumulate) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)B * T) return;
  const int b = (int)(row / T), t = (int)(row % T);
  const int S1 = S + 1;
  const float g = gscale[b];
  float a[MAXC_PER_LANE], acc[MAXC_PER_LANE];
  const float* amr = am + row * C;
#pragma unroll
  for (int j = 0; j < MAXC_PER_LANE; ++j) {
    const int c = lane + 64 * j;
    a[j] = c < C ? amr[c] : 0.f;
    acc[j] = 0.f;
  }
  const long s0 = ranges[row * R];
  for (int i = 0; i < R; ++i) {
    const long s = s0 + i;
    if (s > S) break;
    const float* lmr = lm + ((long)b * S1 + s) * C;
    const float z = lse[row * R + i];
    const float gy = dpy[((long)b * S1 + s) * T + t] * g;
    const float gx = (s < S) ? dpx[((long)b * S + s) * (T + 1) + t] * g : 0.f;
    const long y = (s < S) ? symbols[(long)b * S + s] : -1;
    const float gt = gx + gy;
#pragma unroll
    for (int j = 0; j < MAXC_PER_LANE; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float pre = a[j] + lmr[c];
        const float o = act_fwd(pre, act);
        float d = -gt * expf(o - z);
        if (c == blank) d += gy;
        if (c == y) d += gx;
        acc[j] += d * act_deriv(pre, o, act);
      }
    }
  }
  float* o = d_am + row * C;
#pragma unroll
  for (int j = 0; j < MAXC_PER_LANE; ++j) {
    const int c = lane + 64 * j;
    if (c < C) o[c] = accumulate ? o[c] + acc[j] : acc[j];
  }
}

// d_lm[b,s,:]: one workgroup of DLM_W waves per (b,s); t-range [t_lo,t_hi) with
// s0[t] <= s < s0[t]+R (s0 is non-decreasing in t, so the frames that visit s are contiguous).
// The range is as long as the alignment dwells on s -- a handful of frames on a trained model,
// up to T on a fresh one -- and every frame is a dependent chain of loads + C exponentials: a
// single wave per row left the launch waiting for its longest rows (501 us at C3, 61 us for the
// d_am pass over the same nodes).  The waves of the workgroup stride the range and add up
// through LDS.
constexpr int DLM_W = 8;
template <int MAXC_PER_LANE>
__global__ __launch_bounds__(64 * DLM_W) void pruned_dlm_kernel(
    const float* __restrict__ am, const float* __restrict__ lm, const long* __restrict__ ranges,
    const long* __restrict__ symbols, const float* __restrict__ lse,
    const float* __restrict__ dpx, const float* __restrict__ dpy, const float* __restrict__ gscale,
    int B, int S, int T, int C, int R, int blank, int act, float* __restrict__ d_lm,
    int accumulate) {
  __shared__ float s_acc[DLM_W][64 * MAXC_PER_LANE];
  const int S1 = S + 1;
  const long row = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = (int)(row / S1), s = (int)(row % S1);
  const float g = gscale[b];
  const long* rb = ranges + (long)b * T * R;
  // first t with s0[t] + R > s  (s0 non-decreasing)
  int lo = 0, hi = T;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (rb[(long)mid * R] + R > s) hi = mid; else lo = mid + 1;
  }
  const int t_lo = lo;
  lo = t_lo;
  hi = T;
  while (lo < hi) {  // first t with s0[t] > s
    const int mid = (lo + hi) >> 1;
    if (rb[(long)mid * R] > s) hi = mid; else lo = mid + 1;
  }
  const int t_hi = lo;
  float l[MAXC_PER_LANE], acc[MAXC_PER_LANE];
  const float* lmr = lm + row * C;
#pragma unroll
  for (int j = 0; j < MAXC_PER_LANE; ++j) {
    const int c = lane + 64 * j;
    l[j] = c < C ? lmr[c] : 0.f;
    acc[j] = 0.f;
  }
  const long y = (s < S) ? symbols[(long)b * S + s] : -1;
  for (int t = t_lo + wave; t < t_hi; t += DLM_W) {
    const int i = s - (int)rb[(long)t * R];
    const float* amr = am + ((long)b * T + t) * C;
    const float z = lse[((long)b * T + t) * R + i];
    const float gy = dpy[((long)b * S1 + s) * T + t] * g;
    const float gx = (s < S) ? dpx[((long)b * S + s) * (T + 1) + t] * g : 0.f;
    const float gt = gx + gy;
#pragma unroll
    for (int j = 0; j < MAXC_PER_LANE; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float pre = amr[c] + l[j];
        const float o = act_fwd(pre, act);
        float d = -gt * expf(o - z);
        if (c == blank) d += gy;
        if (c == y) d += gx;
        acc[j] += d * act_deriv(pre, o, act);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXC_PER_LANE; ++j) s_acc[wave][lane + 64 * j] = acc[j];
  __syncthreads();
  float* o = d_lm + row * C;
  for (int c = threadIdx.x; c < C; c += 64 * DLM_W) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < DLM_W; ++w) v += s_acc[w][c];
    o[c] = accumulate ? o[c] + v : v;
  }
}

// ------------------------------------------------- materialised lattice rows
// logits [B][T][U1][V] (full, U1 = S+1) or pruned [B][T][R][V] with ranges.
// One wave per lattice node: Z = logsumexp; writes px/py (pre-filled -inf).
__global__ __launch_bounds__(256) void lattice_fwd_kernel(
    const float* __restrict__ logits, const long* __restrict__ ranges,
    const long* __restrict__ symbols, const long* __restrict__ boundary, int B, int S, int T,
    int V, int R, int blank, float* __restrict__ px, float* __restrict__ py,
    float* __restrict__ lse) {
  const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int S1 = S + 1;
  if (node >= (long)B * T * R) return;
  const int i = (int)(node % R);
  const long bt = node / R;
  const int b = (int)(bt / T), t = (int)(bt % T);
  const long s = ranges ? ranges[bt * R] + i : i;
  if (s > S) return;
  const float* x = logits + node * V;
  float m = S2T_NEG_INF;
  for (int c = lane; c < V; c += 64) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float sum = 0.f;
  for (int c = lane; c < V; c += 64) sum += expf(x[c] - m);
  sum = wave_sum(sum);
  const float z = m + logf(sum);
  if (lane == 0) {
    lse[node] = z;
    const long Tb = boundary[b * 4 + 3];
    py[((long)b * S1 + s) * T + t] = x[blank] - z;
    if (s < S) {
      float v = x[symbols[(long)b * S + s]] - z;
      if (t == Tb) v = S2T_NEG_INF;
      px[((long)b * S + s) * (T + 1) + t] = v;
    }
  }
}

__global__ __launch_bounds__(256) void lattice_bwd_kernel(
    const float* __restrict__ logits, const long* __restrict__ ranges,
    const long* __restrict__ symbols, const float* __restrict__ lse,
    const float* __restrict__ dpx, const float* __restrict__ dpy, const float* __restrict__ gscale,
    int B, int S, int T, int V, int R, int blank, float* __restrict__ d_logits) {
  const long node = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int S1 = S + 1;
  if (node >= (long)B * T * R) return;
  const int i = (int)(node % R);
  const long bt = node / R;
  const int b = (int)(bt / T), t = (int)(bt % T);
  const long s = ranges ? ranges[bt * R] + i : i;
  float* o = d_logits + node * V;
  if (s > S) {
    for (int c = lane; c < V; c += 64) o[c] = 0.f;
    return;
  }
  const float g = gscale[b];
  const float* x = logits + node * V;
  const float z = lse[node];
  const float gy = dpy[((long)b * S1 + s) * T + t] * g;
  const float gx = (s < S) ? dpx[((long)b * S + s) * (T + 1) + t] * g : 0.f;
  const long y = (s < S) ? symbols[(long)b * S + s] : -1;
  const float gt = gx + gy;
  for (int c = lane; c < V; c += 64) {
    float d = -gt * expf(x[c] - z);
    if (c == blank) d += gy;
    if (c == y) d += gx;
    o[c] = d;
  }
}

__global__ void fill_kernel(float* __restrict__ p, long n, float v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

inline int mi_threads(int S) {
  int nt = ((S + 1 + 63) / 64) * 64;
  return nt;
}

}  // namespace

extern "C" int s2t_fill_f32(float* p, long n, float v, void* stream) {
  if (n <= 0) return 0;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, n,
                     v);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_row_exp(const float* x, long rows, int C, float* probs, float* rowmax,
                                void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(row_exp_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, rows, C, probs, rowmax);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_simple_pxpy(const float* am, const float* lm, const float* am_max,
                                    const float* lm_max, const float* nrm, const long* symbols,
                                    const long* boundary, int B, int S, int T, int C, int blank,
                                    float* px, float* py, void* stream) {
  if (B <= 0) return 0;
  if (S < 0 || T <= 0 || C <= 0) return -1;
  dim3 grid((T + 1 + 255) / 256, S + 1, B);
  hipLaunchKernelGGL(simple_pxpy_kernel, grid, dim3(256), 0, (hipStream_t)stream, am, lm, am_max,
                     lm_max, nrm, symbols, boundary, S, T, C, blank, px, py);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_simple_w(const float* dpx, const float* dpy, const float* nrm,
                                 const float* gscale, int B, int S, int T, float* W,
                                 void* stream) {
  if (B <= 0) return 0;
  dim3 grid((T + 255) / 256, S + 1, B);
  hipLaunchKernelGGL(simple_w_kernel, grid, dim3(256), 0, (hipStream_t)stream, dpx, dpy, nrm,
                     gscale, S, T, W);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_simple_bwd(const float* am_probs, const float* lm_probs,
                                   const float* G_am, const float* G_lm, const float* dpx,
                                   const float* dpy, const float* gscale, const long* symbols,
                                   int B, int S, int T, int C, int blank, float* d_am,
                                   float* d_lm, int accumulate, void* stream) {
  if (B <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  long rows = (long)B * T;
  hipLaunchKernelGGL(simple_dam_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256),
                     sizeof(float) * 4 * C, st,
                     am_probs, G_am, dpx, dpy, gscale, symbols, B, S, T, C, blank, d_am,
                     accumulate);
  S2T_CHECK_LAUNCH();
  rows = (long)B * (S + 1);
  hipLaunchKernelGGL(simple_dlm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256),
                     sizeof(float) * 4 * C, st,
                     lm_probs, G_lm, dpx, dpy, gscale, symbols, B, S, T, C, blank, d_lm,
                     accumulate);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_mutual_info_fwd(const float* px, const float* py, const long* boundary, int B,
                                   int S, int T, float* p, float* ans, void* stream) {
  if (B <= 0) return 0;
  if (S < 0 || T < 0 || S + 1 > 1024) return -1;
  const int nt = mi_threads(S);
  hipLaunchKernelGGL(mi_fwd_kernel, dim3(B), dim3(nt), sizeof(float) * 2 * (nt + 1),
                     (hipStream_t)stream, px, py, boundary, S, T, p, ans);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_mutual_info_bwd(const float* px, const float* py, const long* boundary,
                                   const float* p, const float* ans_grad, int B, int S, int T,
                                   float* px_grad, float* py_grad, void* stream) {
  if (B <= 0) return 0;
  if (S < 0 || T < 0 || S + 1 > 1024) return -1;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(px_grad, 0, sizeof(float) * (size_t)B * S * (T + 1), st);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(py_grad, 0, sizeof(float) * (size_t)B * (S + 1) * T, st);
  if (e != hipSuccess) return (int)e;
  const int nt = mi_threads(S);
  hipLaunchKernelGGL(mi_bwd_kernel, dim3(B), dim3(nt), sizeof(float) * 4 * (nt + 1), st, px, py,
                     boundary, p, ans_grad, S, T, px_grad, py_grad);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_prune_ranges(const float* px_grad, const float* py_grad,
                                     const long* boundary, int B, int S, int T, int s_range,
                                     long* ranges, void* stream) {
  if (B <= 0) return 0;
  if (S < 1 || T < 1 || s_range < 1 || s_range > S + 1) return -1;
  hipLaunchKernelGGL(prune_ranges_kernel, dim3(B), dim3(256), sizeof(long) * T,
                     (hipStream_t)stream, px_grad, py_grad, boundary, S, T, s_range, ranges);
  S2T_CHECK_LAUNCH();
  return 0;
}

#define S2T_DISPATCH_C(KERNEL, ...)                                                         \
  do {                                                                                      \
    const unsigned nb__ = (unsigned)((rows + 3) / 4);                                       \
    if (C <= 128)                                                                           \
      hipLaunchKernelGGL(KERNEL<2>, dim3(nb__), dim3(256), 0, st, __VA_ARGS__);             \
    else if (C <= 256)                                                                      \
      hipLaunchKernelGGL(KERNEL<4>, dim3(nb__), dim3(256), 0, st, __VA_ARGS__);             \
    else if (C <= 512)                                                                      \
      hipLaunchKernelGGL(KERNEL<8>, dim3(nb__), dim3(256), 0, st, __VA_ARGS__);             \
    else if (C <= 1024)                                                                     \
      hipLaunchKernelGGL(KERNEL<16>, dim3(nb__), dim3(256), 0, st, __VA_ARGS__);            \
    else                                                                                    \
      return -1;                                                                            \
  } while (0)

extern "C" int s2t_rnnt_pruned_fwd(const float* am, const float* lm, const long* ranges,
                                   const long* symbols, const long* boundary, int B, int S, int T,
                                   int C, int R, int blank, int act, float* px, float* py,
                                   float* lse, void* stream) {
  if (B <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int rc = s2t_fill_f32(px, (long)B * S * (T + 1), S2T_NEG_INF, stream);
  if (rc) return rc;
  rc = s2t_fill_f32(py, (long)B * (S + 1) * T, S2T_NEG_INF, stream);
  if (rc) return rc;
  const long rows = (long)B * T;
  S2T_DISPATCH_C(pruned_fwd_kernel, am, lm, ranges, symbols, boundary, B, S, T, C, R, blank, act,
                 px, py, lse);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_pruned_bwd(const float* am, const float* lm, const long* ranges,
                                   const long* symbols, const float* lse, const float* dpx,
                                   const float* dpy, const float* gscale, int B, int S, int T,
                                   int C, int R, int blank, int act, float* d_am, float* d_lm,
                                   int accumulate, void* stream) {
  if (B <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  {
    const long rows = (long)B * T;
    S2T_DISPATCH_C(pruned_dam_kernel, am, lm, ranges, symbols, lse, dpx, dpy, gscale, B, S, T, C,
                   R, blank, act, d_am, accumulate);
    S2T_CHECK_LAUNCH();
  }
  {
    const unsigned nb = (unsigned)((long)B * (S + 1));      // one DLM_W-wave workgroup per (b, s)
#define S2T_DLM(MC)                                                                              \
  hipLaunchKernelGGL(pruned_dlm_kernel<MC>, dim3(nb), dim3(64 * DLM_W), 0, st, am, lm, ranges, \
                     symbols, lse, dpx, dpy, gscale, B, S, T, C, R, blank, act, d_lm, accumulate)
    if (C <= 128) S2T_DLM(2);
    else if (C <= 256) S2T_DLM(4);
    else if (C <= 512) S2T_DLM(8);
    else if (C <= 1024) S2T_DLM(16);
    else return -1;
#undef S2T_DLM
    S2T_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int s2t_rnnt_lattice_fwd(const float* logits, const long* ranges, const long* symbols,
                                    const long* boundary, int B, int S, int T, int V, int R,
                                    int blank, float* px, float* py, float* lse, void* stream) {
  if (B <= 0) return 0;
  int rc = s2t_fill_f32(px, (long)B * S * (T + 1), S2T_NEG_INF, stream);
  if (rc) return rc;
  rc = s2t_fill_f32(py, (long)B * (S + 1) * T, S2T_NEG_INF, stream);
  if (rc) return rc;
  const long nodes = (long)B * T * R;
  hipLaunchKernelGGL(lattice_fwd_kernel, dim3((unsigned)((nodes + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, logits, ranges, symbols, boundary, B, S, T, V, R, blank,
                     px, py, lse);
  S2T_CHECK_LAUNCH();
  return 0;
}

extern "C" int s2t_rnnt_lattice_bwd(const float* logits, const long* ranges, const long* symbols,
                                    const float* lse, const float* dpx, const float* dpy,
                                    const float* gscale, int B, int S, int T, int V, int R,
                                    int blank, float* d_logits, void* stream) {
  if (B <= 0) return 0;
  const long nodes = (long)B * T * R;
  hipLaunchKernelGGL(lattice_bwd_kernel, dim3((unsigned)((nodes + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, logits, ranges, symbols, lse, dpx, dpy, gscale, B, S, T,
                     V, R, blank, d_logits);
  S2T_CHECK_LAUNCH();
  return 0;
}
