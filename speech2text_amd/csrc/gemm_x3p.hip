// fp32 GEMM on the bf16 matrix cores with the WEIGHT operand split ahead of time -- the forward
// (y = x W^T + b) and data-gradient (dx = g W) products of every Linear on the layer path
// (reference: model/encoder/zipformer.py:1924-1975, 2372-2378, 2643-2695 and
// model/layer/scaling.py:1512-1583 under loss.backward()).
//
//   C[M,N] = A[M,K] . Bm[N,K]^T (+ bias[N]) (* act'(S[M,N])) (+ R[M,N]);   C2 = act(C) (optional)
//
// Arithmetic as gemm.hip's X3 path: every fp32 value is the exact sum of three bf16 pieces, six
// piece products per term on v_mfma_f32_32x32x16_bf16, fp32 accumulation (fp32-level error).
// What is different here:
//   * Bm is a parameter.  Its pieces are written ONCE per optimizer step (s2t_x3p_split: one launch
//     for every weight of the model, both orientations) in fragment-major order
//     [N/32][K/16][3][64 lanes][8 bf16]: the 1 KB one MFMA B operand needs is contiguous, a k-chunk
//     of a column tile is one 6 KB run.  Staging B is a straight 16-byte copy global -> LDS and a
//     fragment is ONE conflict-free ds_read_b128; no VALU work on the weight side at all.
//   * A (activations / gradients, fp32 row-major) is split ONCE per element, when the staged
//     registers are written to LDS, into the same fragment-major image (gemm.hip splits every
//     fragment again in each of the two waves that read it): the split costs a quarter of the VALU
//     cycles and the main loop is {12 ds_read_b128, 24 MFMA} per 16-deep step.
//   * the epilogue carries the layer's elementwise neighbours: bias, residual, the activation's
//     derivative (data gradient through Swoosh) and a second output act(C) (the kept activation).
// Workgroup = 4 waves (2 x 2), block tile (64 TM) x (64 TN), k chunks of 32, register-prefetched.
#include "common.h"
#include "../../include/s2t_mi355.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// staging registers are native vectors (first-class values): HIP's float4 / uint4 are structs, and a
// select between two of them goes through the stack
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// operand pointers are GLOBAL-address-space pointers, said so explicitly: left generic, the
// compiler loses their address space through the tile loop's selects in some instantiations and
// emits flat_load -- which counts in lgkmcnt AS WELL, so every barrier's LDS wait then also waits
// for the stage that was requested a moment ago (the 2x2 tile ran that way until late round 4)
typedef const __attribute__((address_space(1))) float* gf32p;
typedef const __attribute__((address_space(1))) f32x4* gf32x4p;
typedef const __attribute__((address_space(1))) u32x4* gu32x4p;

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
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, u32x4& q0, u32x4& q1,
                                       u32x4& q2) {
  unsigned a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
  split_pair(v0.x, v0.y, a0, a1, a2);
  split_pair(v0.z, v0.w, b0, b1, b2);
  split_pair(v1.x, v1.y, c0, c1, c2);
  split_pair(v1.z, v1.w, d0, d1, d2);
  q0 = u32x4{a0, b0, c0, d0};
  q1 = u32x4{a1, b1, c1, d1};
  q2 = u32x4{a2, b2, c2, d2};
}

// the TWO-piece split of the bf16x2 arithmetic (S2T_GEMM_ARITH=2): x = p0 + p1 + O(2^-18 |x|)
__device__ __forceinline__ void split_pair2(float x0, float x1, unsigned& p0, unsigned& p1) {
  f32x2 x = {x0, x1};
  p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  f32x2 h = {__uint_as_float(p0 << 16), __uint_as_float(p0 & 0xFFFF0000u)};
  x = x - h;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}
// NP pieces of 8 values (NP = 3: exact; NP = 2: the two leading pieces)
template <int NP>
__device__ __forceinline__ void split8n(const f32x4 v0, const f32x4 v1, u32x4 (&q)[3]) {
  if (NP == 3) {
    split8(v0, v1, q[0], q[1], q[2]);
  } else {
    unsigned a0, a1, b0, b1, c0, c1, d0, d1;
    split_pair2(v0.x, v0.y, a0, a1);
    split_pair2(v0.z, v0.w, b0, b1);
    split_pair2(v1.x, v1.y, c0, c1);
    split_pair2(v1.z, v1.w, d0, d1);
    q[0] = u32x4{a0, b0, c0, d0};
    q[1] = u32x4{a1, b1, c1, d1};
  }
}

enum { ACT_NONE = 0, ACT_SWOOSH_L = 1, ACT_SWOOSH_R = 2 };

__device__ __forceinline__ float log1p_fast(float e) {   // as zip_elem.hip
  const float u = 1.f + e;
  return u == 1.f ? e : __logf(u) * __fdividef(e, u - 1.f);
}
__device__ __forceinline__ float swoosh(float x, int kind) {
  // log(1 + exp(x - off)) - 0.08 x - c   (scaling.py:1340-1343, 1418-1423); as zip_elem.hip swoosh_f
  const float off = kind == ACT_SWOOSH_L ? 4.f : 1.f;
  const float c = kind == ACT_SWOOSH_L ? 0.035f : 0.313261687f;
  const float z = x - off;
  return fmaxf(z, 0.f) + log1p_fast(__expf(-fabsf(z))) - 0.08f * x - c;
}
__device__ __forceinline__ float swoosh_deriv(float x, int kind) {
  const float off = kind == ACT_SWOOSH_L ? 4.f : 1.f;
  return __fdividef(1.f, 1.f + __expf(off - x)) - 0.08f;
}

struct X3PMap {
  int on, hw, w;
  long sb, sh, sw, base;
};
__device__ __forceinline__ long x3p_maprow(const X3PMap& m, int r) {
  const int b = r / m.hw, q = r - b * m.hw, i = q / m.w, j = q - i * m.w;
  return m.base + b * m.sb + i * m.sh + j * m.sw;
}

struct X3P {
  const float* A;
  long lda;
  const unsigned short* Bp;   // fragment-major pieces [NT][KB][3][64][8], KB even
  int NT, KB;
  float* C;
  long ldc;
  int M, N, K;
  const float* bias;          // [N] or NULL
  // up to TWO [M][N] operands of the epilogue, in the order they are applied (filled by the entry
  // point from act_src / resid / resid_b); role 1: C *= act'(op) (act_kind), 2: C += op, 3: the
  // second addend -- C += op, or with act2 3 into C2 only
  const float* op[2];
  long ldop[2];
  int role[2];
  int act_kind;
  float* C2;                  // C2 = act2(C) (act2 1 | 2), or C + (role-3 operand) (act2 3), or NULL
  long ldc2;
  int act2;
  int tiles_m, tiles_n;
  int wgs_per_cu;             // persistent form: workgroups per CU (0 = default)
  int prio;                   // 1: wave priority by the workgroup's slot on its CU (see x3p_set_prio)
  // Balancer update in the epilogue (s2t_gemm_x3p_bal): C = act'(S) (A Bm^T) is the gradient w.r.t. S
  // coming through the activation, and the Balancer on S (scaling.py:741-789 in closed form, as
  // zip_elem.hip balancer_apply_fused_kernel) adds |C| (a[c] + b[c] S): bal_stats = [4][1024]: column
  // sums / sums of squares of S over bal_n rows, or NULL
  const float* bal_stats;
  float bal_n, bal_min_mean, bal_max_mean, bal_min_rms, bal_max_rms, bal_gs;
  // sums of squares taken while C leaves (s2t_gemm_x3p_sq: Whiten's backward needs ||g||^2 and ||pg||^2
  // before it can combine them, scaling.py:1024-1027): sq_sums = DEVICE [2][64] partial sums -- slot
  // (workgroup, wave) & 63 of row 0 gets the squares of the role-4 operand (read for this purpose only,
  // not added), of row 1 those of C as stored; or NULL
  float* sq_sums;
  float sq_pad_;
  // implicit operands (s2t_gemm_x3p_map, the 3x3 convolutions of model/encoder/conformer.py:47-57):
  // row r of A / C is not at r * ld but at map(r) = base + b sb + i sh + j sw with r = (b, i, j) over a
  // (hw = rows per image, w = columns per image row) grid, and the K axis of A is nseg segments of seg16
  // stages (16 floats each) that start segoff[s] floats from the row's base
  X3PMap amap, cmap;
  int seg16, nseg;
  long segoff[4];
  long c_elems;               // elements of the mapped C buffer (its buffer resource)
};

// The workgroups of a one-round grid start together and, left alone, run in lockstep: all of a CU's
// workgroups multiply at the same time (sharing the matrix pipe) and then all store at the same
// time (matrix pipe idle, HBM write-bound).  A static wave priority by the workgroup's slot on its
// CU (HW_ID.TG_ID) lets slot 0 win the issue arbitration: it finishes its main loop first and its
// stores drain under the other slots' MFMAs, and so on down the slots.
__device__ __forceinline__ void x3p_set_prio(int on) {
  if (!on) return;
  if (on == 2) { __builtin_amdgcn_s_setprio(3); return; }   // every wave above the other kernels on the CU
  const unsigned tg = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (16 << 6) | 4) & 15u;   // HW_ID.TG_ID
  if (tg == 0) __builtin_amdgcn_s_setprio(3);
  else if (tg == 1) __builtin_amdgcn_s_setprio(2);
  else if (tg == 2) __builtin_amdgcn_s_setprio(1);
}

// ---- epilogue shared by the kernels below: bias, act' of a saved tensor, residual, second output.
// One SLICE = rows 16 h .. 16 h + 15 of the wave's 32 x 32 sub-tile (i, j): lane holds column
// (lane & 31), rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5); the 16 rows go through a per-wave LDS
// scratch (16 x 36 floats) and leave as 16-byte row pieces.  Wave-local: no workgroup barrier.
//
// ORDER OF THE MEMORY OPERATIONS.  vmcnt counts loads and stores in issue order, so a wait for a
// load that was issued AFTER a store also waits for that store's acknowledgement -- a bias or
// residual load placed next to its use makes every slice wait out the previous slice's stores
// (one write round trip per slice: the epilogue then takes 4 TM TN of them).  Hence, per row block
// i of the wave's sub-tiles (2 TN slices), TWO PHASES: (1) ALL operand loads of the row block are
// issued together; every slice is brought into its final form -- LDS exchange, bias / act' /
// residuals -- and written BACK into the accumulator registers (a lane owns 16 values of a
// 32 x 32 sub-tile in either layout); nothing is stored; (2) the row block's stores, back to back,
// nothing waits for them (the next row block's operand wait does: TM - 1 such waits per tile, none
// without operands).  The bias quad is loaded once per column block, before phase 1.
// Loads and stores are BUFFER operations on straight-line code: a lane outside the matrix (or an
// absent matrix: a resource of zero bytes) gets an out-of-range offset, which loads 0 / drops the
// store without a branch.
struct EpiOps {               // one slice's operands: [row of the pair][slot]
  f32x4 v[2][2];
};
struct EpiRs {                // buffer resources of the epilogue's matrices (absent: zero bytes)
  __amdgpu_buffer_rsrc_t c, c2, op[2];
};
constexpr unsigned kOob = 0xFFFFFFF0u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3p_rsrc(const float* p, long ld, int rows) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0,
                                           p ? (int)((long)rows * ld * 4) : 0, 0x00020000);
}
__device__ __forceinline__ EpiRs x3p_epi_rsrc(const X3P& g) {
  EpiRs r;
  r.c = x3p_rsrc(g.C, g.ldc, g.M);
  r.c2 = x3p_rsrc(g.C2, g.ldc2, g.M);
  r.op[0] = x3p_rsrc(g.op[0], g.ldop[0], g.M);
  r.op[1] = x3p_rsrc(g.op[1], g.ldop[1], g.M);
  return r;
}
__device__ __forceinline__ f32x4 x3p_bload(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
__device__ __forceinline__ void x3p_bstore(__amdgpu_buffer_rsrc_t rs, unsigned off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off, 0, 0);
}

__device__ __forceinline__ f32x4 x3p_epi_bias(const X3P& g, int j, int n0, int wcb, int lane) {
  const int col = n0 + 32 * (wcb + j) + (lane & 7) * 4;
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (g.bias && col < g.N) b = *reinterpret_cast<const f32x4*>(g.bias + col);
  return b;
}

__device__ __forceinline__ void x3p_epi_load(const X3P& g, const EpiRs& rs, EpiOps& o, int i, int j,
                                             int h, int m0, int n0, int wrb, int wcb, int lane) {
  const int er = lane >> 3, ec = (lane & 7) * 4;     // this lane's row (of 8) and column quad
  const int col = n0 + 32 * (wcb + j) + ec;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = m0 + 32 * (wrb + i) + 16 * h + er + 8 * q;
    const bool ok = row < g.M && col < g.N;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      o.v[q][k] = z;
      if (g.role[k])                                 // (uniform)
        o.v[q][k] = x3p_bload(rs.op[k], ok ? (unsigned)(row * (int)g.ldop[k] + col) * 4u : kOob);
    }
  }
}

// per-channel coefficients of the Balancer update for this lane's column quad (formulas and clamps
// of zip_elem.hip balancer_apply_fused_kernel)
struct BalQ {
  f32x4 a, b;
};
__device__ __forceinline__ BalQ x3p_epi_balcoef(const X3P& g, int j, int n0, int wcb, int lane) {
  // the per-column (a, b) of the update, computed once per launch by bal_coef_kernel into the
  // second half of the caller's buffer (a tile computing them itself spent more cycles on the
  // sqrt / log / divisions of its 64 columns than on a K = 128 main loop: the ConvNeXt data
  // gradient took 1130 us with the Balancer against 570 without)
  BalQ r;
  r.a = r.b = f32x4{0.f, 0.f, 0.f, 0.f};
  const int col = n0 + 32 * (wcb + j) + (lane & 7) * 4;
  if (col >= g.N) return r;
  r.a = *reinterpret_cast<const f32x4*>(g.bal_stats + 2048 + col);
  r.b = *reinterpret_cast<const f32x4*>(g.bal_stats + 3072 + col);
  return r;
}

// stats [2][1024] (column sums | sums of squares over n rows) -> coefficients [2][1024] (a | b) of
// model/layer/scaling.py:741-789 in closed form (zip_elem.hip balancer_apply_fused_kernel)
__global__ __launch_bounds__(256) void bal_coef_kernel(const float* __restrict__ stats, float* __restrict__ coef,
                                                        int N, float n, float min_mean, float max_mean,
                                                        float min_rms, float max_rms, float gs) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const float inv_n = 1.f / n;
  const float mean = stats[c] * inv_n, uvar = stats[1024 + c] * inv_n;
  const float raw_var = uvar - mean * mean;
  const bool live_v = raw_var > 1.0e-20f, live_r = uvar > 1.0e-20f;
  const float var = fmaxf(raw_var, 1.0e-20f);
  const float sd = sqrtf(var);
  const float rms = sqrtf(fmaxf(uvar, 1.0e-20f));
  const float m = mean / sd;
  const float mc = fminf(fmaxf(m, min_mean), max_mean);
  const float s_m = (m > mc) ? 1.f : ((m < mc) ? -1.f : 0.f);
  const float rc = fminf(fmaxf(rms, min_rms), max_rms);
  const float lq = logf(rc / rms);
  const float s_r = (lq > 0.f) ? -1.f : ((lq < 0.f) ? 1.f : 0.f);
  const float a = s_m * inv_n * (live_v ? (1.f / sd + mean * mean / (sd * var)) : 1.f / sd);
  const float b = (live_v ? -s_m * inv_n * mean / (sd * var) : 0.f) + (live_r ? s_r * inv_n / (rms * rms) : 0.f);
  const float lg_rms = fmaxf(sqrtf(fmaxf(a * a + 2.f * a * b * mean + b * b * uvar, 0.f)), 1.0e-20f);
  const float k = gs / lg_rms;
  coef[c] = a * k;
  coef[1024 + c] = b * k;
}

// phase 1 of one slice: a[8 h .. 8 h + 7] (MFMA layout) -> the final values of this lane's two row
// pieces (q = 0, 1: a[8 h + 4 q .. + 3])
template <bool BAL = false>
__device__ __forceinline__ void x3p_epi_xform(const X3P& g, f32x16& a, float* scr, const EpiOps& o,
                                              const f32x4 bq, int h, int lane, const BalQ& bal = BalQ{}) {
  const int hi = lane >> 5, lo = lane & 31;
  const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
  for (int r = 0; r < 8; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * hi) * 36 + lo] = a[8 * h + r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    f32x4 v = *reinterpret_cast<const f32x4*>(scr + (er + 8 * q) * 36 + ec);
    v += bq;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const f32x4 x = o.v[q][k];
      if (g.role[k] == 1) {
        v.x *= swoosh_deriv(x.x, g.act_kind);
        v.y *= swoosh_deriv(x.y, g.act_kind);
        v.z *= swoosh_deriv(x.z, g.act_kind);
        v.w *= swoosh_deriv(x.w, g.act_kind);
        if (BAL) {                                   // the Balancer on the activation's input
          v.x += fabsf(v.x) * fmaf(bal.b.x, x.x, bal.a.x);
          v.y += fabsf(v.y) * fmaf(bal.b.y, x.y, bal.a.y);
          v.z += fabsf(v.z) * fmaf(bal.b.z, x.z, bal.a.z);
          v.w += fabsf(v.w) * fmaf(bal.b.w, x.w, bal.a.w);
        }
      } else if (g.role[k] == 2 || (g.role[k] == 3 && g.act2 != 3)) {
        v += x;
      }
    }
    a[8 * h + 4 * q + 0] = v.x;
    a[8 * h + 4 * q + 1] = v.y;
    a[8 * h + 4 * q + 2] = v.z;
    a[8 * h + 4 * q + 3] = v.w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_wave_barrier();
}

// phase 2 of one slice: the two row pieces leave (out-of-matrix lanes: dropped by the buffer check)
struct EpiSq {                // this lane's running sums of squares (s2t_gemm_x3p_sq): operand, C
  float o = 0.f, c = 0.f;
};
template <bool CMAP = false>
__device__ __forceinline__ void x3p_epi_store(const X3P& g, const EpiRs& rs, const f32x16& a,
                                              const EpiOps& o, int i, int j, int h, int m0, int n0,
                                              int wrb, int wcb, int lane, EpiSq& sq) {
  const int er = lane >> 3, ec = (lane & 7) * 4;
  const int col = n0 + 32 * (wcb + j) + ec;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = m0 + 32 * (wrb + i) + 16 * h + er + 8 * q;
    const bool ok = row < g.M && col < g.N;
    const f32x4 v = {a[8 * h + 4 * q], a[8 * h + 4 * q + 1], a[8 * h + 4 * q + 2], a[8 * h + 4 * q + 3]};
    if (CMAP) {                                      // (mapped output rows; no second output)
      x3p_bstore(rs.c, ok ? (unsigned)(x3p_maprow(g.cmap, min(row, g.M - 1)) + col) * 4u : kOob, v);
      continue;
    }
    x3p_bstore(rs.c, ok ? (unsigned)(row * (int)g.ldc + col) * 4u : kOob, v);
    if (g.sq_sums && ok) {                           // (operands of lanes outside the matrix were read as 0)
      sq.c += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      const f32x4 t = g.role[0] == 4 ? o.v[q][0] : o.v[q][1];
      sq.o += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
    if (g.act2 == 3) {                               // C2 = C + the role-3 operand (still in its slot)
      const f32x4 u = v + (g.role[0] == 3 ? o.v[q][0] : o.v[q][1]);
      x3p_bstore(rs.c2, ok ? (unsigned)(row * (int)g.ldc2 + col) * 4u : kOob, u);
    } else if (g.act2) {
      const f32x4 u = {swoosh(v.x, g.act2), swoosh(v.y, g.act2), swoosh(v.z, g.act2), swoosh(v.w, g.act2)};
      x3p_bstore(rs.c2, ok ? (unsigned)(row * (int)g.ldc2 + col) * 4u : kOob, u);
    }
  }
}

// LEAN: one slice at a time (operands of ONE slice live: 16 registers instead of 32 TN) -- for the
// kernels that run three or more workgroups per CU on a tight register budget, where the other
// workgroups cover a slice's load -> store round trip
// BAL: the Balancer update of s2t_gemm_x3p_bal is compiled in (its own instantiations: the others pay
// no registers for it)
template <int TM, int TN, bool LEAN = false, bool BAL = false, bool CMAP = false>
__device__ __forceinline__ void x3p_epilogue(const X3P& g, f32x16 (&acc)[TM][TN], unsigned char* smem,
                                             int m0, int n0, int wrb, int wcb, int wave, int lane,
                                             bool sync = true) {
  if (sync) __syncthreads();                         // all waves finished reading the stage buffers
  float* scr = reinterpret_cast<float*>(smem) + wave * (16 * 36);
  EpiRs rs = x3p_epi_rsrc(g);
  if (CMAP) rs.c = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, (int)(g.c_elems * 4), 0x00020000);
  EpiSq sq;
  // this tile's sums of squares: one atomic pair per wave into one of 64 slots (the same two words for
  // every wave of the chip would serialise)
  auto sq_commit = [&]() {
    if (!g.sq_sums) return;                          // (uniform)
    const float so = wave_sum(sq.o), sc = wave_sum(sq.c);
    if (lane == 0) {
      const int slot = (blockIdx.x * 4 + wave) & 63;
      atomicAdd(g.sq_sums + slot, so);
      atomicAdd(g.sq_sums + 64 + slot, sc);
    }
  };
  if (LEAN) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const f32x4 bqj = x3p_epi_bias(g, j, n0, wcb, lane);
      const BalQ balj = BAL ? x3p_epi_balcoef(g, j, n0, wcb, lane) : BalQ{};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          EpiOps o;
          x3p_epi_load(g, rs, o, i, j, h, m0, n0, wrb, wcb, lane);
          x3p_epi_xform<BAL>(g, acc[i][j], scr, o, bqj, h, lane, balj);
          x3p_epi_store<CMAP>(g, rs, acc[i][j], o, i, j, h, m0, n0, wrb, wcb, lane, sq);
        }
    }
    sq_commit();
    return;
  }
  f32x4 bq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) bq[j] = x3p_epi_bias(g, j, n0, wcb, lane);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    EpiOps ops[2 * TN];                              // slice t of the row block = (j, h) = (t / 2, t & 1)
#pragma unroll
    for (int t = 0; t < 2 * TN; ++t) x3p_epi_load(g, rs, ops[t], i, t / 2, t & 1, m0, n0, wrb, wcb, lane);
#pragma unroll
    for (int t = 0; t < 2 * TN; ++t) x3p_epi_xform(g, acc[i][t / 2], scr, ops[t], bq[t / 2], t & 1, lane);
#pragma unroll
    for (int t = 0; t < 2 * TN; ++t)
      x3p_epi_store(g, rs, acc[i][t / 2], ops[t], i, t / 2, t & 1, m0, n0, wrb, wcb, lane, sq);
    __builtin_amdgcn_sched_barrier(0);               // (the next row block's loads stay behind these stores)
  }
  sq_commit();
}

// ---- register-staged, software-pipelined form: k stages of 16, TWO LDS stages, one barrier per stage.
// Iteration kb {barrier; split + store the staged registers (stage kb+1) into the other buffer; request
// stage kb+2 from global memory; multiply stage kb} -- a wave's staging VALU / LDS stores and its MFMAs
// belong to the same barrier interval, so the matrix pipe of a SIMD is fed by every resident wave all
// the time instead of by whichever workgroup happens to be in its "multiply" phase.
// (Round 6 removed this kernel's diagnostic and experimental instantiations -- per-iteration clock
// stamps, ablation builds, the start stagger, the sliced epilogue under the next tile: their
// measurements are DESIGN 3f; `git log -- speech2text_amd/csrc/gemm_x3p.hip` has the code.)
// (waves per SIMD the register allocation must leave room for: 2 / 3 / 4 workgroups per CU for the
// 2x2 / 1x2, 2x1 / 1x1 tiles -- the epilogue's operand loads are hoisted as far as this allows)
// NP: pieces per operand (3: the exact bf16x3 split, six products per term; 2: bf16x2, the three
// leading products -- the LDS images, the weight-piece copies and the fragment reads shrink to the
// first NP pieces of every 3-piece group of the plane image, which is the same for both)
template <int TM, int TN, bool BAL = false, int NP = 3>
__global__ __launch_bounds__(256, TM * TN == 4 ? 2 : (TM * TN == 2 ? 3 : 4))
void x3p_db_kernel(X3P g) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_ST = 2 * TM * NP * 1024, B_ST = 2 * TN * NP * 1024, ST = A_ST + B_ST;
  constexpr int NAU = (128 * TM + 255) / 256;       // A units (8 k of one row) per thread and stage
  constexpr int NBP = 2 * TN * NP * 64;             // B 16-byte pieces per stage
  constexpr int NBU = (NBP + 255) / 256;            // ... per thread
  constexpr int SCR = 4 * 16 * 36 * 4;              // epilogue scratch
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * ST > SCR ? 2 * ST : SCR];

  // PERSISTENT workgroups: the grid is (a multiple of 8, at most) what the chip holds at once; a
  // workgroup walks tiles loc, loc + stride, ... of its XCD's contiguous range (n fastest: the
  // n-tiles of an m-panel share that XCD's L2 copy of the A panel).  The output stores of tile t
  // drain while tile t+1 is multiplied: stage 0 of the next tile is requested BEFORE the stores are
  // issued, so its wait (vmcnt counts loads and stores in issue order) does not include them.
  const int total = g.tiles_m * g.tiles_n;
  const int per_xcd = (total + 7) / 8;
  const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
  int loc = blockIdx.x >> 3;
  if (loc >= per_xcd || xcd * per_xcd + loc >= total) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrb = (wave >> 1) * TM, wcb = (wave & 1) * TN;
  x3p_set_prio(g.prio);

  // A unit u: row r = (u & 15) + 16 (u >> 5), k half kq = (u >> 4) & 1: 16 consecutive lanes store 16
  // consecutive fragment slots (conflict-free), both halves of a row's 64 bytes in one instruction
  int a_r[NAU], a_k[NAU];
  unsigned a_dst[NAU];
  bool a_on[NAU];
#pragma unroll
  for (int i = 0; i < NAU; ++i) {
    const int u = tid + 256 * i;
    a_on[i] = u < 128 * TM;
    const int uu = a_on[i] ? u : 0;
    const int r = (uu & 15) + 16 * (uu >> 5), kq = (uu >> 4) & 1;
    a_r[i] = r;
    a_k[i] = 8 * kq;
    a_dst[i] = (unsigned)((((r >> 5) * NP) * 64 + kq * 32 + (r & 31)) * 16);
  }
  int b_seg[NBU], b_off[NBU];
  bool b_on[NBU];
#pragma unroll
  for (int i = 0; i < NBU; ++i) {
    const int idx = tid + 256 * i;
    b_on[i] = idx < NBP;
    const int ii = b_on[i] ? idx : 0;
    b_seg[i] = ii / (64 * NP);                      // 32-column block; the first NP KB of its 3 KB chunk
    b_off[i] = ii - b_seg[i] * (64 * NP);
  }
  const float* asrc[NAU];
  const u32x4* bsrc[NBU];
  int m0, n0;
#define X3P_TILE(LOC)                                                                        \
  {                                                                                          \
    const int lin_ = xcd * per_xcd + (LOC);                                                  \
    const int tm_ = lin_ / g.tiles_n, tn_ = lin_ - tm_ * g.tiles_n;                          \
    m0 = tm_ * BM;                                                                           \
    n0 = tn_ * BN;                                                                           \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i)                                          \
      asrc[i] = g.A + (long)min(m0 + a_r[i], g.M - 1) * g.lda + a_k[i];                      \
    _Pragma("unroll") for (int i = 0; i < NBU; ++i) {                                        \
      const int nt_ = min((n0 >> 5) + b_seg[i], g.NT - 1);                                   \
      bsrc[i] = reinterpret_cast<const u32x4*>(g.Bp + (long)nt_ * g.KB * 1536) + b_off[i];   \
    }                                                                                        \
  }

  const int nst = (g.K + 15) >> 4;                  // stages (the pieces are zero beyond K)
  // Two sets of staging registers: stage kb+2 is requested at the TOP of iteration kb into the set
  // iteration kb-1 emptied, a whole iteration before its data is touched (same-box A/B: 3 % faster
  // than one set).
  f32x4 ra[2][NAU][2];
  u32x4 rb[2][NBU];
  u32x4 qa[NAU][3];                                 // (NP of them used)
  // global -> registers, unconditional (past the end the last stage again; a k tail is read from
  // the row's start and zeroed when it is split): nothing here waits for the data
#define X3P_LOAD(SET, S)                                                                     \
  {                                                                                          \
    const int ss_ = min((S), nst - 1);                                                       \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i) {                                        \
      gf32p p_ = (gf32p)(asrc[i] + ((16 * ss_ + a_k[i] < g.K) ? 16 * ss_ : 0));              \
      ra[SET][i][0] = *reinterpret_cast<gf32x4p>(p_);                                        \
      ra[SET][i][1] = *reinterpret_cast<gf32x4p>(p_ + 4);                                    \
    }                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NBU; ++i) rb[SET][i] = ((gu32x4p)bsrc[i])[(long)ss_ * 192]; \
  }
#define X3P_SPLIT(SET, S)                                                                    \
  {                                                                                          \
    const int ss_ = min((S), nst - 1);                                                       \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i) {                                        \
      const bool v_ = 16 * ss_ + a_k[i] < g.K;                                               \
      const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                 \
      split8n<NP>(v_ ? ra[SET][i][0] : z_, v_ ? ra[SET][i][1] : z_, qa[i]);                  \
    }                                                                                        \
  }
#define X3P_STORE(SET, BUF)                                                                  \
  {                                                                                          \
    unsigned char* const sa_ = smem + (BUF) * ST;                                            \
    unsigned char* const sb_ = sa_ + A_ST;                                                   \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i)                                          \
      if (NAU * 256 == 128 * TM || a_on[i]) {                                                \
        _Pragma("unroll") for (int p = 0; p < NP; ++p)                                       \
          *reinterpret_cast<u32x4*>(sa_ + a_dst[i] + 1024 * p) = qa[i][p];                   \
      }                                                                                      \
    _Pragma("unroll") for (int i = 0; i < NBU; ++i)                                          \
      if (NBU * 256 == NBP || b_on[i])                                                       \
        *reinterpret_cast<u32x4*>(sb_ + (tid + 256 * i) * 16) = rb[SET][i];                  \
  }
#define X3P_TERM(PA, PB)                                                                        \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0);
  // one stage: CUR = the register set that holds stage KB+1, NXT = the set stage KB+2 goes into
#define X3P_ITER(KB, CUR, NXT)                                                               \
  {                                                                                          \
    __syncthreads(); /* stage KB is in LDS for everyone; the other buffer's readers are done */ \
    X3P_LOAD(NXT, (KB) + 2)                                                                  \
    const unsigned char* const sa = smem + ((KB) & 1) * ST;                                  \
    const unsigned char* const sb = sa + A_ST;                                               \
    bf16x8 fa[TM][NP], fb[TN][NP];                                                           \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int p = 0; p < NP; ++p) \
      fa[i][p] = *reinterpret_cast<const bf16x8*>(sa + (((wrb + i) * NP + p) * 64 + lane) * 16); \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int p = 0; p < NP; ++p) \
      fb[j][p] = *reinterpret_cast<const bf16x8*>(sb + (((wcb + j) * NP + p) * 64 + lane) * 16); \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    /* (1) the smallest product groups with the split of stage KB+1's A registers in the      \
       MFMAs' shadow (an MFMA holds the SIMD's issue for 8 of its 32 cycles) */              \
    X3P_SPLIT(CUR, (KB) + 1)                                                                 \
    if (NP == 3) { X3P_TERM(NP - 1, 0) X3P_TERM(1, 1) } else { X3P_TERM(1, 0) }              \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    /* (2) stage KB+1 into the other LDS buffer */                                           \
    X3P_STORE(CUR, ((KB) + 1) & 1)                                                           \
    if (NP == 3) { X3P_TERM(0, NP - 1) } else { X3P_TERM(0, 1) }                             \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    /* (3) the rest of the products */                                                       \
    if (NP == 3) { X3P_TERM(1, 0) X3P_TERM(0, 1) }                                           \
    X3P_TERM(0, 0)                                                                           \
  }
  X3P_TILE(loc)
  X3P_LOAD(0, 0)
  for (;;) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    X3P_SPLIT(0, 0)
    X3P_STORE(0, 0)
    X3P_LOAD(0, 1)
    for (int kb = 0; kb < nst; kb += 2) {
      X3P_ITER(kb, 0, 1)
      if (kb + 1 < nst) X3P_ITER(kb + 1, 1, 0)
    }
    // next tile: its stage 0 is requested before this tile's stores go out
    const int em0 = m0, en0 = n0;
    loc += stride;
    const bool more = loc < per_xcd && xcd * per_xcd + loc < total;
    if (more) {
      X3P_TILE(loc)
      X3P_LOAD(0, 0)
    }
    x3p_epilogue<TM, TN, BAL || (TM == 1 && TN == 2), BAL>(g, acc, smem, em0, en0, wrb, wcb, wave, lane, true);
    if (!more) break;
    __syncthreads();     // the stage buffers (and the epilogue scratch in them) are free
  }
#undef X3P_ITER
#undef X3P_TERM
#undef X3P_LOAD
#undef X3P_SPLIT
#undef X3P_STORE
#undef X3P_TILE
}

// ---- LDS-DMA form (round 5): the weight pieces of a stage go global -> LDS directly
// (global_load_lds_dwordx4: the fragment-major image makes a stage of B one straight run), no
// staging registers and no ds_write for them; A as before (registers, split once, stored as
// pieces), ONE set of staging registers.  What that buys is registers and LDS-store cycles: the
// 2 x 2 tile fits THREE workgroups per CU (<= 168 registers, 48 KB of LDS), 2 x 1 / 1 x 2 four --
// one more wave per SIMD to cover a workgroup's dependent chain barrier -> fragment reads -> MFMAs,
// and a grid of 768 slots, which holds the 744 tiles of the 15 872-row products in ONE round where
// 512 slots ran 1.45 rounds (the second less than half full).
// The DMA is inline asm (as a builtin the compiler would drain vmcnt(0) before every LDS read that
// follows it in program order) and is waited for by hand; the A loads stay ordinary loads, and the
// schedule keeps the two kinds of wait from seeing each other's operations: a DMA is issued AFTER
// the point where the compiler waits for the A registers and has landed (hand-counted wait at the
// top of the next iteration) BEFORE the compiler waits again.  (A loads as inline asm do not work:
// the compiler may copy an asm output to other registers, or reuse it as a temporary, before the
// hand-written wait -- the pending load then lands on top of an address computation.)
//   iteration kb:  wait B(kb) landed (vmcnt = the A loads in flight);  barrier;
//                  fragments of stage kb;  split A(kb+1) (the compiler's wait: only A is in flight);
//                  DMA B(kb+1) -> buffer (kb+1) & 1;  request A(kb+2);  first products;
//                  store A(kb+1)'s pieces -> buffer (kb+1) & 1;  the other products.
// (per-lane 64-bit address, no SGPR base: the form the round-3 LDS-DMA kernel ran with)
__device__ __forceinline__ void x3p_glds16(const void* g, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_addr)
               : "memory");     // (m0 is reserved: the compiler keeps nothing in it across statements)
}
__device__ __forceinline__ unsigned x3p_lds_addr(const void* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}

// NP: pieces per operand (3 | 2, as x3p_db_kernel).  KS: 16-deep sub-stages per barrier interval
// (1 | 2): with two pieces a 16-deep stage holds only 12 MFMAs per wave (2 x 2 tile) between two
// barriers; KS = 2 stages 32 k at a time -- the same 24 MFMAs per barrier as the three-piece form,
// half the barriers, waits and address arithmetic per product.
// WM: waves along M (2: the 4-wave workgroup, block tile 64 TM x 64 TN; 4: an 8-wave workgroup, block
// tile 128 TM x 64 TN -- every wave keeps its 32 TM x 32 TN, a weight stage is shared by twice the
// rows: 0.75 of the global -> LDS bytes per product at TM = TN = 2)
template <int TM, int TN, int WPC, bool MAP = false, int NP = 3, int KS = 1, bool BAL = false, int WM = 2>
__global__ __launch_bounds__(128 * WM, WPC * WM / 2) void x3p_dma_kernel(X3P g) {   // (HIP: waves per SIMD)
  constexpr int NT = 128 * WM, NWV = 2 * WM;        // threads, waves
  constexpr int BM = 32 * WM * TM, BN = 64 * TN;
  constexpr int A_SUB = WM * TM * NP * 1024, B_SUB = 2 * TN * NP * 1024;   // one 16-deep sub-stage
  constexpr int A_ST = KS * A_SUB, B_ST = KS * B_SUB, ST = A_ST + B_ST;
  constexpr int NAU = (2 * BM * KS + NT - 1) / NT;  // A units (8 k of one row) per thread and stage
  constexpr int NRUN = 2 * TN * KS * NP;            // B: 1 KB runs (one wave-instruction each) per stage
  constexpr int NBW = (NRUN + NWV - 1) / NWV;       // ... per wave
  constexpr int SCR = NWV * 16 * 36 * 4;
  static_assert(2 * BM * KS % NT == 0 || (TM == 1 && KS == 1), "A units");
  // (implicit operands with KS = 2: a segment is a whole number of 32-deep intervals, checked by the entry point)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * ST > SCR ? 2 * ST : SCR];
  const unsigned lds0 = x3p_lds_addr(smem);

  const int total = g.tiles_m * g.tiles_n;
  const int per_xcd = (total + 7) / 8;
  const int xcd = blockIdx.x & 7, stride = gridDim.x >> 3;
  int loc = blockIdx.x >> 3;
  if (loc >= per_xcd || xcd * per_xcd + loc >= total) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wrb = (wave >> 1) * TM, wcb = (wave & 1) * TN;
  if (g.prio >= 2) x3p_set_prio(g.prio);

  // A unit u = (sub-stage, row r, k half kq): 16 consecutive lanes store 16 consecutive fragment slots
  int a_r[NAU], a_k[NAU];
  unsigned a_dst[NAU];
  bool a_on[NAU];
#pragma unroll
  for (int i = 0; i < NAU; ++i) {
    const int u = tid + NT * i;
    a_on[i] = u < 2 * BM * KS;
    const int uu = a_on[i] ? u : 0;
    const int sub = uu / (2 * BM), v = uu - sub * (2 * BM);
    const int r = (v & 15) + 16 * (v >> 5), kq = (v >> 4) & 1;
    a_r[i] = r;
    a_k[i] = 16 * sub + 8 * kq;
    a_dst[i] = (unsigned)(sub * A_SUB + (((r >> 5) * NP) * 64 + kq * 32 + (r & 31)) * 16);
  }
  // B: wave-instruction q of this wave moves the 1 KB run pw = wave + NWV q = ((32-column block) KS +
  // sub-stage) NP + piece of the stage (a wave past the end repeats the last run: same bytes to the
  // same place, no branch); a block's 16-deep chunk is 3 KB = 1536 bf16 of the plane image, of which
  // the first NP KB are read
  const int nst = (g.K + 15) >> 4;                  // 16-deep stages (the pieces are zero beyond K)
  const int nss = (nst + KS - 1) / KS;              // barrier intervals
  const int lane8 = lane * 8;                       // (bf16 elements: 16 bytes per lane)
  const float* asrc[NAU];
  const unsigned short* bsrc[NBW];
  unsigned b_dst[NBW];
#pragma unroll
  for (int q = 0; q < NBW; ++q) {
    const int pw = min(wave + NWV * q, NRUN - 1);
    const int blk = pw / (KS * NP), rem = pw - blk * (KS * NP), sub = rem / NP, pc = rem - sub * NP;
    b_dst[q] = (unsigned)(A_ST + sub * B_SUB + (blk * NP + pc) * 1024);
  }
  int m0, n0;
#define XD_TILE(LOC)                                                                         \
  {                                                                                          \
    const int lin_ = xcd * per_xcd + (LOC);                                                  \
    const int tm_ = lin_ / g.tiles_n, tn_ = lin_ - tm_ * g.tiles_n;                          \
    m0 = tm_ * BM;                                                                           \
    n0 = tn_ * BN;                                                                           \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i)                                          \
      asrc[i] = g.A + (MAP ? x3p_maprow(g.amap, min(m0 + a_r[i], g.M - 1))                   \
                           : (long)min(m0 + a_r[i], g.M - 1) * g.lda);                       \
    _Pragma("unroll") for (int q = 0; q < NBW; ++q) {                                        \
      const int pw_ = min(wave + NWV * q, NRUN - 1);                                         \
      const int blk_ = pw_ / (KS * NP), rem_ = pw_ - blk_ * (KS * NP);                       \
      const int sub_ = rem_ / NP, pc_ = rem_ - sub_ * NP;                                    \
      const int nt_ = min((n0 >> 5) + blk_, g.NT - 1);        /* 32-column block */          \
      bsrc[q] = g.Bp + (long)nt_ * g.KB * 1536 + sub_ * 1536 + pc_ * 512;                    \
    }                                                                                        \
  }
  f32x4 ra[NAU][2];
  u32x4 qa[NAU][3];
#define XD_LOAD_A(S)                                                                         \
  {                                                                                          \
    const int ss_ = min((S), nss - 1);                                                       \
    long ko_ = 16L * KS * ss_;                                                               \
    if (MAP) {                         /* stage -> (segment, offset inside it): uniform */    \
      const int spi_ = g.seg16 / KS;                       /* intervals per segment */        \
      const int sg_ = ss_ / spi_;                                                            \
      /* (selects, not g.segoff[sg_]: a dynamic index would put the array in scratch) */      \
      const long so_ = sg_ == 0 ? g.segoff[0] : sg_ == 1 ? g.segoff[1] : sg_ == 2 ? g.segoff[2] : g.segoff[3]; \
      ko_ = so_ + 16L * KS * (ss_ - sg_ * spi_);                                             \
    }                                                                                        \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i) {                                        \
      /* (a k tail is read from the row's start and zeroed when it is split) */              \
      gf32p p_ = (gf32p)(asrc[i] + ((16 * KS * ss_ + a_k[i] < g.K) ? ko_ + a_k[i] : 0L));    \
      ra[i][0] = *reinterpret_cast<gf32x4p>(p_);                                             \
      ra[i][1] = *reinterpret_cast<gf32x4p>(p_ + 4);                                         \
    }                                                                                        \
  }
#define XD_DMA_B(S, BUF)                                                                     \
  {                                                                                          \
    const int ss_ = min((S), nss - 1);                                                       \
    _Pragma("unroll") for (int q = 0; q < NBW; ++q)                                          \
      x3p_glds16(bsrc[q] + (long)ss_ * (KS * 1536) + lane8, lds0 + (BUF) * ST + b_dst[q]);   \
  }
#define XD_SPLIT(S)                                                                          \
  {                                                                                          \
    const int ss_ = min((S), nss - 1);                                                       \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i) {                                        \
      const bool v_ = 16 * KS * ss_ + a_k[i] < g.K;                                          \
      const f32x4 z_ = {0.f, 0.f, 0.f, 0.f};                                                 \
      split8n<NP>(v_ ? ra[i][0] : z_, v_ ? ra[i][1] : z_, qa[i]);                            \
    }                                                                                        \
  }
#define XD_STORE_A(BUF)                                                                      \
  {                                                                                          \
    unsigned char* const sa_ = smem + (BUF) * ST;                                            \
    _Pragma("unroll") for (int i = 0; i < NAU; ++i)                                          \
      if (NAU * NT == 2 * BM * KS || a_on[i]) {                                              \
        _Pragma("unroll") for (int p = 0; p < NP; ++p)                                       \
          *reinterpret_cast<u32x4*>(sa_ + a_dst[i] + 1024 * p) = qa[i][p];                   \
      }                                                                                      \
  }
#define XD_TERM(SUB, PA, PB)                                                                    \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SUB][i][PA], fb[SUB][j][PB], acc[i][j], 0, 0, 0);
  // the products of sub-stage SUB that follow its first group (smallest pieces first)
#define XD_REST(SUB)                                                                         \
  if (NP == 3) { XD_TERM(SUB, 0, NP - 1) XD_TERM(SUB, 1, 0) XD_TERM(SUB, 0, 1) XD_TERM(SUB, 0, 0) } \
  else { XD_TERM(SUB, 0, 1) XD_TERM(SUB, 0, 0) }

  for (;;) {
    XD_TILE(loc)
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // prologue: B(0) by DMA, A(0) through the registers into buffer 0, A(1) requested
    XD_DMA_B(0, 0)
    XD_LOAD_A(0)
    XD_SPLIT(0)
    XD_STORE_A(0)
    XD_LOAD_A(1)
    for (int kb = 0; kb < nss; ++kb) {
      // B(kb) landed (its DMAs are older than the 2 NAU loads of A(kb+1), which stay in flight);
      // stage kb's A pieces were stored before this barrier by every wave
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * NAU) : "memory");
      __syncthreads();
      const unsigned char* const sa = smem + (kb & 1) * ST;
      const unsigned char* const sb = sa + A_ST;
      bf16x8 fa[KS][TM][NP], fb[KS][TN][NP];
#pragma unroll
      for (int u = 0; u < KS; ++u) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int p = 0; p < NP; ++p)
            fa[u][i][p] = *reinterpret_cast<const bf16x8*>(sa + u * A_SUB + (((wrb + i) * NP + p) * 64 + lane) * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int p = 0; p < NP; ++p)
            fb[u][j][p] = *reinterpret_cast<const bf16x8*>(sb + u * B_SUB + (((wcb + j) * NP + p) * 64 + lane) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
      XD_SPLIT(kb + 1)                 // (the compiler's wait for A(kb+1): nothing else is in flight)
      __builtin_amdgcn_sched_barrier(0);
      XD_DMA_B(kb + 1, (kb + 1) & 1)   // every wave is past this iteration's barrier: the buffer's readers are done
      XD_LOAD_A(kb + 2)                // (the registers are free again: their values sit in qa)
      if (NP == 3) { XD_TERM(0, NP - 1, 0) XD_TERM(0, 1, 1) } else { XD_TERM(0, 1, 0) }
      __builtin_amdgcn_sched_barrier(0);
      XD_STORE_A((kb + 1) & 1)
      XD_REST(0)
      if (KS == 2) {
        __builtin_amdgcn_sched_barrier(0);
        if (NP == 3) { XD_TERM(KS - 1, NP - 1, 0) XD_TERM(KS - 1, 1, 1) } else { XD_TERM(KS - 1, 1, 0) }
        XD_REST(KS - 1)
      }
    }
    // the trailing (dummy) DMA and loads must have landed before the stage buffers become the
    // epilogue's scratch / the next tile's stages
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MAP && g.cmap.on) x3p_epilogue<TM, TN, true, false, true>(g, acc, smem, m0, n0, wrb, wcb, wave, lane, true);
    else x3p_epilogue<TM, TN, true, BAL>(g, acc, smem, m0, n0, wrb, wcb, wave, lane, true);
    loc += stride;
    if (!(loc < per_xcd && xcd * per_xcd + loc < total)) break;
    __syncthreads();
  }
#undef XD_TILE
#undef XD_LOAD_A
#undef XD_DMA_B
#undef XD_SPLIT
#undef XD_STORE_A
#undef XD_TERM
#undef XD_REST
}

// ---- the weights' pieces, all matrices of a model in one launch.  Descriptor d covers blocks
// [blk_begin[d], blk_begin[d+1]); a wave = one (column tile nt, k block kb) = three 1 KB fragments.
__global__ __launch_bounds__(256) void x3p_split_kernel(const float* __restrict__ base,
                                                        const S2tPlaneDesc* __restrict__ tab, int n,
                                                        unsigned short* __restrict__ dst) {
  int lo_d = 0, hi_d = n - 1;
  while (lo_d < hi_d) {                       // last descriptor whose blk_begin <= blockIdx.x
    const int mid = (lo_d + hi_d + 1) >> 1;
    if (tab[mid].blk_begin <= blockIdx.x) lo_d = mid; else hi_d = mid - 1;
  }
  const S2tPlaneDesc d = tab[lo_d];
  const int NT = (d.N + 31) >> 5, KB = 2 * ((d.K + 31) >> 5);
  const long q = (long)(blockIdx.x - d.blk_begin) * 4 + (threadIdx.x >> 6);
  if (q >= (long)NT * KB) return;
  const int nt = (int)(q / KB), kb = (int)(q - (long)nt * KB);
  const int lane = threadIdx.x & 63, lo = lane & 31, hi = lane >> 5;
  const int nn = 32 * nt + lo, k0 = 16 * kb + 8 * hi;
  const float* src = base + d.src_off;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    const bool ok = nn < d.N && k < d.K;
    const long a = d.transposed ? (long)(ok ? k : 0) * d.ld + (ok ? nn : 0)
                                : (long)(ok ? nn : 0) * d.ld + (ok ? k : 0);
    const float x = src[a];
    v[e] = ok ? x : 0.f;
  }
  uint4 q0, q1, q2;
  split_pair(v[0], v[1], q0.x, q1.x, q2.x);
  split_pair(v[2], v[3], q0.y, q1.y, q2.y);
  split_pair(v[4], v[5], q0.z, q1.z, q2.z);
  split_pair(v[6], v[7], q0.w, q1.w, q2.w);
  unsigned short* o = dst + d.dst_off + (((long)nt * KB + kb) * 3 * 64 + lane) * 8;
  *reinterpret_cast<uint4*>(o) = q0;
  *reinterpret_cast<uint4*>(o + 512) = q1;
  *reinterpret_cast<uint4*>(o + 1024) = q2;
}

// a launch that hands an armed (start, stop) event pair (csrc/streams.hip, declared in common.h)
// to the kernel itself
#define X3P_LAUNCH(KERNEL, GRID, BLOCK, SMEM)                                                     \
  do {                                                                                            \
    if (s2t_prof_start) {                                                                         \
      hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), SMEM, st, s2t_prof_start,            \
                            s2t_prof_stop, 0, g);                                                 \
      s2t_prof_start = s2t_prof_stop = nullptr;                                                   \
    } else {                                                                                      \
      hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), SMEM, st, g);                           \
    }                                                                                             \
  } while (0)

// LDS-DMA form: WPC workgroups per CU (the register allocation is bounded accordingly)
template <int TM, int TN, int WPC, int NP = 3, int KS = 1>
void launch_x3p_dma(X3P& g, hipStream_t st) {
  g.tiles_m = (g.M + 64 * TM - 1) / (64 * TM);
  g.tiles_n = (g.N + 64 * TN - 1) / (64 * TN);
  const int total = g.tiles_m * g.tiles_n;
  const int grid = std::min(((total + 7) / 8) * 8, 256 * WPC);
  if (NP == 2 && g.bal_stats)
    X3P_LAUNCH((x3p_dma_kernel<TM, TN, WPC, false, NP, KS, (NP == 2)>), grid, 256, 0);
  else
    X3P_LAUNCH((x3p_dma_kernel<TM, TN, WPC, false, NP, KS>), grid, 256, 0);
}

// the 8-wave workgroup (WM = 4): block tile 128 TM x 64 TN, 512 threads
template <int TM, int TN, int WPC, int NP, int KS>
void launch_x3p_dma8(X3P& g, hipStream_t st) {
  g.tiles_m = (g.M + 128 * TM - 1) / (128 * TM);
  g.tiles_n = (g.N + 64 * TN - 1) / (64 * TN);
  const int total = g.tiles_m * g.tiles_n;
  const int grid = std::min(((total + 7) / 8) * 8, 256 * WPC);
  X3P_LAUNCH((x3p_dma_kernel<TM, TN, WPC, false, NP, KS, false, 4>), grid, 512, 0);
}

template <int TM, int TN, int WPC, int NP = 3, int KS = 1>
void launch_x3p_map(X3P& g, hipStream_t st) {
  g.tiles_m = (g.M + 64 * TM - 1) / (64 * TM);
  g.tiles_n = (g.N + 64 * TN - 1) / (64 * TN);
  const int total = g.tiles_m * g.tiles_n;
  const int grid = std::min(((total + 7) / 8) * 8, 256 * WPC);
  X3P_LAUNCH((x3p_dma_kernel<TM, TN, WPC, true, NP, KS>), grid, 256, 0);
}

template <int TM, int TN, int NP = 3>
void launch_x3p(X3P& g, hipStream_t st) {
  g.tiles_m = (g.M + 64 * TM - 1) / (64 * TM);
  g.tiles_n = (g.N + 64 * TN - 1) / (64 * TN);
  const int total = g.tiles_m * g.tiles_n;
  {
    // persistent grid: S2T_X3P_WGS workgroups per CU (default: what registers / LDS admit, <= 3)
    static int wgs = -1;
    if (wgs < 0) { const char* e = getenv("S2T_X3P_WGS"); wgs = e ? atoi(e) : 0; }
    const int per_cu = wgs > 0 ? wgs : g.wgs_per_cu > 0 ? g.wgs_per_cu : (TM * TN >= 4 ? 2 : TM * TN >= 2 ? 3 : 4);
    const int cap = 256 * per_cu;
    const int grid = std::min(((total + 7) / 8) * 8, cap);
    if (g.bal_stats)
      X3P_LAUNCH((x3p_db_kernel<TM, TN, true, NP>), grid, 256, 0);
    else
      X3P_LAUNCH((x3p_db_kernel<TM, TN, false, NP>), grid, 256, 0);
  }
}

// block tile from the shape: the widest tile that still gives the chip >= ~2 rounds of workgroups
int pick_tile(int M, int N) {
  static int force = -1;      // S2T_X3P_TILE = 11 | 12 | 21 | 22: tuning
  if (force < 0) { const char* e = getenv("S2T_X3P_TILE"); force = e ? atoi(e) : 0; }
  if (force > 0) return force;
  const long t22 = (long)((M + 127) / 128) * ((N + 127) / 128);
  if (t22 >= 768) return 22;
  const long t21 = (long)((M + 127) / 128) * ((N + 63) / 64);
  if (N <= 64 || t21 >= 640) return 21;
  const long t12 = (long)((M + 63) / 64) * ((N + 127) / 128);
  if (t12 >= 640 && N > 64) return 12;
  return 11;
}

}  // namespace

static int g_arith_forced = 0;

extern "C" {

// ---- the arithmetic of the bf16 GEMMs (this file, the weight-gradient and NT / NN kernels of
// gemm.hip): 3 = every fp32 operand as the EXACT sum of three bf16 pieces, six piece products per
// term (fp32-level error, <= 2e-6 of max); 2 = two pieces per operand, the three leading products
// a_hi b_hi + a_hi b_lo + a_lo b_hi (error ~ 2^-17 per term: torch's float32 matmul precision
// "high"; the reference trains with the looser "medium", build_task.py:79).  Read PER CALL, per
// CLASS of product (include/s2t_mi355.h): S2T_GEMM_ARITH=3 | 2 (also "bf16x3" / "bf16x2") for every
// class, S2T_GEMM_ARITH_F / _D / _W / _S for one (forward, data gradient, weight gradient,
// statistics), unless s2t_gemm_arith_set pinned one value for all (0 = follow the environment again).
static int arith_parse(const char* e, int dflt) {
  if (!e || !*e) return dflt;
  const char* x = strchr(e, 'x');        // "2", "3", "bf16x2", "bf16x3", "bf16x2/3", "bf16x3/6"
  const char c = x ? x[1] : e[0];
  return c == '2' ? 2 : c == '3' ? 3 : dflt;
}
static thread_local int g_cls = -1;
int s2t_gemm_class_set(int cls) {
  const int prev = g_cls;
  g_cls = (cls >= 0 && cls <= 3) ? cls : -1;
  return prev;
}
int s2t_gemm_arith_of(int cls) {
  if (g_arith_forced) return g_arith_forced;
  static const char* const kEnv[4] = {"S2T_GEMM_ARITH_F", "S2T_GEMM_ARITH_D", "S2T_GEMM_ARITH_W", "S2T_GEMM_ARITH_S"};
  static const int kDefault[4] = {S2T_GEMM_ARITH_DEFAULT_F, S2T_GEMM_ARITH_DEFAULT_D, S2T_GEMM_ARITH_DEFAULT_W,
                                  S2T_GEMM_ARITH_DEFAULT_S};
  // S2T_GEMM_ARITH sets EVERY class (also the statistics: "3" is the all-six-product step of rounds
  // 3-5, "2" the all-three-product one); the per-class variables override it
  const char* all = getenv("S2T_GEMM_ARITH");
  const bool has_all = all && *all;
  if (cls < 0 || cls > 3) return arith_parse(all, S2T_GEMM_ARITH_DEFAULT);
  return arith_parse(getenv(kEnv[cls]), has_all ? arith_parse(all, kDefault[cls]) : kDefault[cls]);
}
int s2t_gemm_arith(void) { return s2t_gemm_arith_of(g_cls); }
int s2t_gemm_arith_set(int arith) {
  if (arith != 0 && arith != 2 && arith != 3) return -1;
  g_arith_forced = arith;
  return 0;
}

long s2t_x3p_plane_elems(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return 3L * 512 * ((N + 31) / 32) * (2L * ((K + 31) / 32));
}

long s2t_x3p_split_blocks(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (((long)((N + 31) / 32) * (2L * ((K + 31) / 32))) + 3) / 4;
}

int s2t_x3p_split(const float* base, const void* tab, int n, int total_blocks, unsigned short* dst,
                  void* stream) {
  if (n <= 0 || total_blocks <= 0) return 0;
  if (!base || !tab || !dst || (reinterpret_cast<uintptr_t>(dst) & 15)) return -1;
  hipLaunchKernelGGL(x3p_split_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     base, reinterpret_cast<const S2tPlaneDesc*>(tab), n, dst);
  S2T_CHECK_LAUNCH();
  return 0;
}

struct BalArm {
  const float* stats = nullptr;
  float n = 0.f, min_mean = 0.f, max_mean = 0.f, min_rms = 0.f, max_rms = 0.f, gs = 0.f;
};
static thread_local BalArm g_bal;        // armed by s2t_gemm_x3p_bal for the one launch it makes
struct SqArm {
  float* sums = nullptr;
  const float* other = nullptr;
  long ld = 0;
};
static thread_local SqArm g_sq;          // armed by s2t_gemm_x3p_sq for the one launch it makes

// ---- sampled kernel-attached timing of this entry point, kept HERE so that launches issued by the
// native layer executor (csrc/zip_layer.hip) and by the Python call sites are sampled alike: while a
// sample is open every `every`-th launch is issued with its own (start, stop) event pair
// (hipExtLaunchKernelGGL: the kernel's begin / end times) and its algorithmic bytes / flops are added up.
namespace {
struct Sampler {
  int every = 0;
  long count = 0, launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool, used;
  double bytes = 0.0, flops = 0.0, bytes_min = 0.0;
} g_samp;
}  // namespace
static long g_x3p_calls = 0;

int s2t_gemm_x3p(const float* A, long lda, const unsigned short* Bp, int N, int K, float* C, long ldc,
                 int M, const float* bias, const float* resid, long ldr, const float* act_src,
                 long ld_act, int act_kind, float* C2, long ldc2, int act2, const float* resid_b,
                 long ldrb, int tile, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !Bp || !C) return -1;
  if (act_kind < 0 || act_kind > 2 || act2 < 0 || act2 > 3 || (act_src && act_kind == 0) ||
      (C2 && act2 == 0) || (act2 == 3 && (!C2 || !resid_b)))
    return -1;
  // tile = 100 wgs + (10 tm + tn): the register-staged form, wgs = persistent workgroups per CU (0 =
  // default); 2000 + 100 ks + tm tn: the LDS-DMA form at its own occupancy, ks = 2: 32-deep barrier
  // intervals (two-piece arithmetic only)
  // 3000 + 100 ks + tm tn: the LDS-DMA form with 8-wave workgroups (block tile 128 tm x 64 tn; two-piece
  // arithmetic, no Balancer epilogue)
  const int arith = s2t_gemm_arith();
  if (tile < 0 || (tile >= 1000 && tile < 2000) || tile >= 4000) return -1;
  const bool w8 = tile / 1000 == 3;
  int dma = tile / 1000 == 2 || w8;
  const int wgs = (tile / 100) % 10;
  tile %= 100;
  if ((tile != 0 && tile != 11 && tile != 12 && tile != 21 && tile != 22) || wgs > 8 ||
      (dma && (tile == 0 || wgs > 2)) || (w8 && (tile == 11 || (wgs != 0 && wgs != 2) || (wgs == 2 && tile == 21))))
    return -1;
  if (dma && wgs == 2 && arith != 2) return -2;      // (the plan's candidate list follows the arithmetic)
  if (w8 && (arith != 2 || g_bal.stats)) return -2;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if ((K & 7) || (N & 3) || (lda & 3) || (ldc & 3) || !al16(A) || !al16(Bp) || !al16(C) ||
      (bias && !al16(bias)) || (resid && (!al16(resid) || (ldr & 3))) ||
      (act_src && (!al16(act_src) || (ld_act & 3))) || (C2 && (!al16(C2) || (ldc2 & 3))) ||
      (resid_b && (!al16(resid_b) || (ldrb & 3))))
    return -2;
  // the epilogue addresses its matrices through 32-bit buffer offsets
  auto fits = [M](const void* p, long ld) { return !p || (long)M * ld * 4 < 0x7FFFFF00L; };
  if (!fits(C, ldc) || !fits(C2, ldc2) || !fits(resid, ldr) || !fits(act_src, ld_act) ||
      !fits(resid_b, ldrb))
    return -2;
  if (act_src && resid && resid_b) return -2;        // two operand slots
  X3P g{A, lda, Bp, (N + 31) / 32, 2 * ((K + 31) / 32), C, ldc, M, N, K, bias, {nullptr, nullptr},
        {0, 0}, {0, 0}, act_kind, C2, ldc2, act2, 0, 0, wgs, 0,
        nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, g_sq.sums, 0.f, X3PMap{0, 1, 1, 0, 0, 0, 0},
        X3PMap{0, 1, 1, 0, 0, 0, 0}, 1, 1, {0, 0, 0, 0}, 0};
  if (g_bal.stats) {
    if (!act_src || N > 1024) return -2;
    g.bal_stats = g_bal.stats;
    g.bal_n = g_bal.n;
    g.bal_min_mean = g_bal.min_mean;
    g.bal_max_mean = g_bal.max_mean;
    g.bal_min_rms = g_bal.min_rms;
    g.bal_max_rms = g_bal.max_rms;
    g.bal_gs = g_bal.gs;
  }
  {
    int k = 0;
    if (act_src) { g.op[k] = act_src; g.ldop[k] = ld_act; g.role[k++] = 1; }
    if (resid) { g.op[k] = resid; g.ldop[k] = ldr; g.role[k++] = 2; }
    if (resid_b) { g.op[k] = resid_b; g.ldop[k] = ldrb; g.role[k++] = 3; }
    if (g_sq.sums && g_sq.other) {       // the companion matrix: read for its squares only (role 4)
      if (k > 1) return -2;
      g.op[k] = g_sq.other; g.ldop[k] = g_sq.ld; g.role[k++] = 4;
    }
  }
  {
    // S2T_X3P_PRIO: 0 = none; 1 = by the workgroup's slot on its CU (register-staged form; round 4);
    // 2 (default, round 6) = every wave of these main-stream kernels above the side stream's kernels
    // on the same CU (33.29 / 33.16 / 33.09 -> 33.07 / 33.17 / 32.92 ms per step, one box)
    static int prio = -1;
    if (prio < 0) { const char* e = getenv("S2T_X3P_PRIO"); prio = e ? atoi(e) : 2; }
    g.prio = prio;
  }
  hipStream_t st = (hipStream_t)stream;
  ++g_x3p_calls;
  if (g_samp.every > 0 && !s2t_prof_start && (g_samp.count++ % g_samp.every) == 0) {
    std::pair<hipEvent_t, hipEvent_t> pr;
    bool ok = true;
    if (!g_samp.pool.empty()) {
      pr = g_samp.pool.back();
      g_samp.pool.pop_back();
    } else {
      ok = hipEventCreate(&pr.first) == hipSuccess && hipEventCreate(&pr.second) == hipSuccess;
    }
    if (ok) {
      g_samp.used.push_back(pr);
      s2t_prof_start = pr.first;
      s2t_prof_stop = pr.second;
      const int extra = (resid != nullptr) + (act_src != nullptr) + (C2 != nullptr) + (resid_b != nullptr);
      g_samp.bytes += 4.0 * M * ((double)N + K + (double)extra * N) + 2.0 * arith * (double)N * K;
      g_samp.bytes_min += 4.0 * M * ((double)N + K) + 2.0 * arith * (double)N * K;
      g_samp.flops += 2.0 * (double)M * N * K;
      ++g_samp.launches;
    }
  }
  if (w8) {
    switch (100 * wgs + tile) {
      case 22: launch_x3p_dma8<2, 2, 2, 2, 1>(g, st); break;
      case 21: launch_x3p_dma8<2, 1, 2, 2, 1>(g, st); break;
      case 12: launch_x3p_dma8<1, 2, 2, 2, 1>(g, st); break;
      case 222: launch_x3p_dma8<2, 2, 1, 2, 2>(g, st); break;
      default: launch_x3p_dma8<1, 2, 2, 2, 2>(g, st); break;     // 212
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  if (dma && g.bal_stats && arith != 2) dma = 0;    // (three pieces: the Balancer epilogue lives in the register-staged form)
  if (dma && arith == 2 && wgs == 2) {              // 32-deep intervals: 64 / 48 / 48 / 32 KB of LDS
    switch (tile) {
      case 22: launch_x3p_dma<2, 2, 2, 2, 2>(g, st); break;
      case 21: launch_x3p_dma<2, 1, 3, 2, 2>(g, st); break;
      case 12: launch_x3p_dma<1, 2, 3, 2, 2>(g, st); break;
      default: launch_x3p_dma<1, 1, 4, 2, 2>(g, st); break;
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  if (dma && arith == 2) {
    switch (tile) {
      case 22: launch_x3p_dma<2, 2, 3, 2>(g, st); break;
      case 21: launch_x3p_dma<2, 1, 4, 2>(g, st); break;
      case 12: launch_x3p_dma<1, 2, 4, 2>(g, st); break;
      default: launch_x3p_dma<1, 1, 5, 2>(g, st); break;
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  if (dma) {
    switch (tile) {
      case 22: launch_x3p_dma<2, 2, 3>(g, st); break;
      case 21: launch_x3p_dma<2, 1, 4>(g, st); break;
      case 12: launch_x3p_dma<1, 2, 4>(g, st); break;
      default: launch_x3p_dma<1, 1, 5>(g, st); break;
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  if (tile == 0) tile = pick_tile(M, N);
  if (arith == 2) {
    switch (tile) {
      case 22: launch_x3p<2, 2, 2>(g, st); break;
      case 21: launch_x3p<2, 1, 2>(g, st); break;
      case 12: launch_x3p<1, 2, 2>(g, st); break;
      default: launch_x3p<1, 1, 2>(g, st); break;
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  switch (tile) {
    case 22: launch_x3p<2, 2>(g, st); break;
    case 21: launch_x3p<2, 1>(g, st); break;
    case 12: launch_x3p<1, 2>(g, st); break;
    default: launch_x3p<1, 1>(g, st); break;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}

long s2t_gemm_x3p_calls(void) { return g_x3p_calls; }

// every > 0: open a sample (every n-th launch from now on); s2t_x3p_sample_end closes it, waits for
// the sampled kernels and reports {launches, their total ms, algorithmic bytes, flops}
int s2t_x3p_sample_begin(int every) {
  for (auto& pr : g_samp.used) g_samp.pool.push_back(pr);
  g_samp.used.clear();
  g_samp.every = every > 0 ? every : 0;
  g_samp.count = g_samp.launches = 0;
  g_samp.bytes = g_samp.flops = g_samp.bytes_min = 0.0;
  return 0;
}
// bytes the sampled products must move at the least (A, C and the weight pieces read: no epilogue
// operands, no second output) -- of the sample s2t_x3p_sample_end closed last
int s2t_x3p_sample_min_bytes(double* bytes_min) {
  if (!bytes_min) return -1;
  *bytes_min = g_samp.bytes_min;
  return 0;
}
int s2t_x3p_sample_end(long* launches, double* total_ms, double* bytes, double* flops) {
  g_samp.every = 0;
  double ms = 0.0;
  long n = 0;
  for (auto& pr : g_samp.used) {
    float t = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
      ms += t;
      ++n;
    }
    g_samp.pool.push_back(pr);
  }
  g_samp.used.clear();
  // (every sampled launch is counted; a pair the profiler left without timestamps -- PMC passes --
  //  drops out of the average only: the total is scaled back to the sampled count)
  if (launches) *launches = g_samp.launches;
  if (total_ms) *total_ms = n > 0 ? ms * (double)g_samp.launches / (double)n : 0.0;
  if (bytes) *bytes = g_samp.bytes;
  if (flops) *flops = g_samp.flops;
  return 0;
}

// C = A' . Bm^T (+ bias) with IMPLICIT operands: row r of A' is nseg segments of `seg` contiguous floats
// of the buffer A, at amap(r) + segoff[s]; row r of C is at cmap(r) (cmap NULL: plain rows of ldc).
// map(r) = base + b sb + i sh + j sw for r = (b, i, j) over (hw rows per image, w per image row).
// The 3x3 / stride-2 convolution of the conformer's Subsampling (model/encoder/conformer.py:47-57)
// without a patch matrix: forward = 3 segments of 3 C floats per output position; data gradient =
// one launch per input-pixel parity class, whose rows gather 1 / 2 / 2 / 4 taps of the zero-bordered
// output gradient and scatter to every second pixel (speech2text_amd/conf_kernels.py).
int s2t_gemm_x3p_map(const float* A, const S2tRowMap* amap, int seg, int nseg, const long* segoff,
                     const unsigned short* Bp, int N, float* C, long ldc, const S2tRowMap* cmap,
                     long c_elems, int M, const float* bias, int tile, void* stream) {
  if (!A || !amap || !segoff || !Bp || !C || M <= 0 || N <= 0 || seg <= 0 || nseg < 1 || nseg > 4) return -1;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  auto m4 = [](const S2tRowMap* m) { return !((m->sb | m->sh | m->sw | m->base) & 3); };
  if ((seg & 15) || (N & 3) || !al16(A) || !al16(Bp) || !al16(C) || (bias && !al16(bias)) || !m4(amap) ||
      (cmap && (!m4(cmap) || c_elems * 4 >= 0x7FFFFF00L)) || (!cmap && ((ldc & 3) || (long)M * ldc * 4 >= 0x7FFFFF00L)))
    return -2;
  for (int i = 0; i < nseg; ++i)
    if (segoff[i] & 3) return -2;
  const int K = seg * nseg;
  X3P g{A, 0, Bp, (N + 31) / 32, 2 * ((K + 31) / 32), C, ldc, M, N, K, bias, {nullptr, nullptr},
        {0, 0}, {0, 0}, 0, nullptr, 0, 0, 0, 0, 0, 0,
        nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, nullptr, 0.f,
        X3PMap{1, amap->hw, amap->w, amap->sb, amap->sh, amap->sw, amap->base},
        cmap ? X3PMap{1, cmap->hw, cmap->w, cmap->sb, cmap->sh, cmap->sw, cmap->base} : X3PMap{0, 1, 1, 0, 0, 0, 0},
        seg / 16, nseg, {0, 0, 0, 0}, c_elems};
  for (int i = 0; i < nseg; ++i) g.segoff[i] = segoff[i];
  hipStream_t st = (hipStream_t)stream;
  ++g_x3p_calls;
  // tile + 200: 32-deep barrier intervals (two-piece arithmetic; segments of whole 32-deep intervals) -- a
  // row's 128 bytes per interval are one cache line, fetched once (16-deep stages fetch every line twice)
  if (tile >= 200) {
    if (s2t_gemm_arith() != 2 || (seg & 31)) return -2;
    switch (tile - 200) {
      case 21: launch_x3p_map<2, 1, 3, 2, 2>(g, st); break;
      case 12: launch_x3p_map<1, 2, 3, 2, 2>(g, st); break;
      case 11: launch_x3p_map<1, 1, 4, 2, 2>(g, st); break;
      default: launch_x3p_map<2, 2, 2, 2, 2>(g, st); break;
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  if (s2t_gemm_arith() == 2) {
    switch (tile) {
      case 21: launch_x3p_map<2, 1, 4, 2>(g, st); break;
      case 12: launch_x3p_map<1, 2, 4, 2>(g, st); break;
      case 11: launch_x3p_map<1, 1, 5, 2>(g, st); break;
      default: launch_x3p_map<2, 2, 3, 2>(g, st); break;
    }
    S2T_CHECK_LAUNCH();
    return 0;
  }
  switch (tile) {
    case 21: launch_x3p_map<2, 1, 4>(g, st); break;
    case 12: launch_x3p_map<1, 2, 4>(g, st); break;
    case 11: launch_x3p_map<1, 1, 5>(g, st); break;
    default: launch_x3p_map<2, 2, 3>(g, st); break;
  }
  S2T_CHECK_LAUNCH();
  return 0;
}

// s2t_gemm_x3p for the data gradient through an activation WITH the Balancer that sits on the
// activation's input folded into the epilogue: C = act'(S) (A Bm^T), then C += |C| (a[c] + b[c] S)
// with the per-channel a, b of model/layer/scaling.py:741-789 (closed form, zip_elem.hip) derived in
// the epilogue from bal_stats = [4][1024] floats: rows 0-1 column sums / sums of squares of S over its M
// rows (s2t_balancer_stats), rows 2-3 written HERE (the per-column a, b: one small launch before the product).  Replaces s2t_balancer_apply's pass over the (M, N) gradient.  act_src (= S)
// is required; N <= 1024.
int s2t_gemm_x3p_bal(const float* A, long lda, const unsigned short* Bp, int N, int K, float* C, long ldc,
                     int M, const float* resid, long ldr, const float* act_src, long ld_act, int act_kind,
                     int tile, float* bal_stats, float min_mean, float max_mean, float min_rms,
                     float max_rms, float grad_scale, void* stream) {
  if (!bal_stats || !act_src || N > 1024) return -1;
  const bool have_coef = (tile & S2T_X3P_BAL_COEF_READY) != 0;    // (s2t_balancer_coef ran where the statistics were taken)
  tile &= ~S2T_X3P_BAL_COEF_READY;
  if (!have_coef)
    hipLaunchKernelGGL(bal_coef_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, bal_stats,
                       bal_stats + 2048, N, (float)M, min_mean, max_mean, min_rms, max_rms, grad_scale);
  g_bal.stats = bal_stats;
  g_bal.n = (float)M;
  g_bal.min_mean = min_mean;
  g_bal.max_mean = max_mean;
  g_bal.min_rms = min_rms;
  g_bal.max_rms = max_rms;
  g_bal.gs = grad_scale;
  const int rc = s2t_gemm_x3p(A, lda, Bp, N, K, C, ldc, M, nullptr, resid, ldr, act_src, ld_act, act_kind, nullptr,
                              0, 0, nullptr, 0, tile, stream);
  g_bal.stats = nullptr;
  return rc;
}

// The coefficient pass of s2t_gemm_x3p_bal as its own call: bal_stats [0..2048) (s2t_balancer_stats over
// `rows` rows) -> [2048..4096).  A caller that takes the statistics early (forward pass, side stream) runs
// this there too and passes tile | S2T_X3P_BAL_COEF_READY to s2t_gemm_x3p_bal.
int s2t_balancer_coef(float* bal_stats, int N, long rows, float min_mean, float max_mean, float min_rms,
                      float max_rms, float grad_scale, void* stream) {
  if (!bal_stats || N <= 0 || N > 1024 || rows <= 0) return -1;
  hipLaunchKernelGGL(bal_coef_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, bal_stats,
                     bal_stats + 2048, N, (float)rows, min_mean, max_mean, min_rms, max_rms, grad_scale);
  S2T_CHECK_LAUNCH();
  return 0;
}

// s2t_gemm_x3p (+ bias) that ALSO adds, while C leaves the accumulators, the squares of C and of a
// companion matrix `other` (M, N; rows ld_other apart; read for this purpose only) into sums =
// DEVICE [2][64] partial sums (row 0: other, row 1: C; zeroed by the caller; the totals are the
// sums over a row's 64 slots).  Whiten's backward (reference model/layer/scaling.py:994-1028): A = x,
// Bp = pieces of dcov, C = pg = x dcov + bias, other = g; s2t_whiten_combine64 then forms
// g + pg * grad_scale ||g|| / (||pg|| + 1e-20) from the two norms.
int s2t_gemm_x3p_sq(const float* A, long lda, const unsigned short* Bp, int N, int K, float* C, long ldc,
                    int M, const float* bias, const float* other, long ld_other, float* sums, int tile,
                    void* stream) {
  if (!sums) return -1;
  // other == NULL: only C's squares are taken (sums row 1) -- the form Whiten's forward uses: x dcov does not
  // depend on the gradient, so it runs where the statistics run; backward adds ||g||^2 with s2t_sumsq64
  if (other && ((reinterpret_cast<uintptr_t>(other) & 15) || (ld_other & 3) || (long)M * ld_other * 4 >= 0x7FFFFF00L))
    return -2;
  g_sq.sums = sums;
  g_sq.other = other;
  g_sq.ld = ld_other;
  const int rc = s2t_gemm_x3p(A, lda, Bp, N, K, C, ldc, M, bias, nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr,
                              0, tile, stream);
  g_sq.sums = nullptr;
  return rc;
}

}  // extern "C"
