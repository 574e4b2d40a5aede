// Native per-layer executor of the Zipformer2 encoder layer (training hot path).
//
// One call issues every launch of a layer's forward (s2t_zip_layer_fwd) or hand-scheduled backward
// (s2t_zip_layer_bwd): the host-side mirror of speech2text_amd/zip_layer.py, which remains the
// reference implementation of the same launch sequence (reference model:
// model/encoder/zipformer.py:909-1338 Zipformer2EncoderLayer.forward under loss.backward(), with
// the gradient-shaping ops of model/layer/scaling.py:741-789, 994-1028, 1153-1190).
//
//   * the caller (Python, at the reference's call sites and in the reference's order) draws the
//     layer's random decisions -- which Balancer / Whiten / limit_param_value / score penalty fire
//     this call -- and hands them over as a decision vector: they are DATA here;
//   * a descriptor (S2tZipLayerDesc) carries the layer's shapes, parameter / gradient addresses,
//     the pre-split bf16 pieces of its weights and the constants of its gradient-shaping modules;
//   * every intermediate lives in a caller-provided workspace (bump-allocated, sized by a dry run
//     of the same code: s2t_zip_layer_ws_floats); what backward needs is remembered in an opaque
//     host-side state block;
//   * which kernel serves a product (our pre-split bf16x3 GEMM or the library plan cache) is read
//     from a table of the timings zip_kernels.lt_matmul took (s2t_zl_plan_put): a layer whose
//     shapes have not been timed yet is refused (-5) before anything is launched and the caller
//     runs the Python executor, which times them.
// Every kernel is an extern "C" entry point of this library, called in exactly the order
// zip_layer.py calls it: results are bit-identical wherever the kernels are (fp32 atomics of the
// weight-gradient GEMM and the Whiten / Balancer statistics excepted).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/s2t_mi355.h"
#include "common.h"

namespace {

constexpr int kMagic = 0x5a4c3532;
constexpr float kPenLimit = 25.0f;                 // zipformer.py:2024-2040 penalize_abs_values_gt
constexpr int kMaxProb = 48;

thread_local char g_err[256] = "";

// ---- pinned host words + events shared by the calls of a process (a slot is in use from a forward
// to the backward of the same step: rings far larger than a step's needs)
std::mutex g_mu;
float* g_pin = nullptr;
unsigned g_pin_next = 0;
constexpr unsigned kPin = 8192;
std::vector<hipEvent_t> g_ev;
unsigned g_ev_next = 0;
constexpr unsigned kEv = 8192;

float* pinned_slot() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_pin) {
    if (hipHostMalloc((void**)&g_pin, kPin * sizeof(float), hipHostMallocDefault) != hipSuccess) return nullptr;
    memset(g_pin, 0, kPin * sizeof(float));
  }
  return g_pin + (g_pin_next++ % kPin);
}
hipEvent_t ring_event() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_ev.size() < kEv) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_ev.push_back(e);
    return e;
  }
  return g_ev[g_ev_next++ % kEv];
}

// ---- plan table: {mode, half-octave of rows, N, K} -> (library ms, own ms | < 0, tile)
struct Base { double t_lib, t_own; int tile; };
std::unordered_map<uint64_t, Base> g_plans;
// (mode: bit 0 = forward / data gradient, bits 4-5 = the arithmetic the bucket was timed under,
//  s2t_gemm_arith(): the candidates and their times differ between three and two pieces)
uint64_t plan_key(int mode, int ho, int N, int K) {
  return ((uint64_t)(mode & 1) << 63) | ((uint64_t)((mode >> 4) & 3) << 56) | ((uint64_t)(ho & 0xFF) << 48) |
         ((uint64_t)(N & 0xFFFFFF) << 24) | (uint64_t)(K & 0xFFFFFF);
}
int plan_mode(int mode) { return (mode & 1) | (s2t_gemm_arith_of(mode & 1) << 4); }
int half_octave(long m) {            // floor(2 log2 m), as zip_kernels._half_octave
  if (m < 1) m = 1;
  int b = 63 - __builtin_clzll((unsigned long long)m);
  return 2 * b + (((double)m * (double)m >= std::ldexp(1.0, 2 * b + 1)) ? 1 : 0);
}

std::atomic<uint64_t> g_bal_seq{0};
uint64_t g_bal_waited = 0;            // (backward runs on one thread at a time: the autograd engine's)
void* g_bal_waited_on = nullptr;

constexpr int kBalSites = 15;
struct WStat {
  int on;
  float *cov, *mean, *scal, *host;
  hipEvent_t ev;
  int G, cg;
  // round-6 form (whiten_x3p == 2): written in forward on the statistics' stream; sums = [2][64] slots
  float *dcov, *bias, *sums;
  unsigned short* pieces;
  float* pg;                 // x dcov + bias, taken in forward as well (whiten_fwd_pg), or NULL
};
struct FfS { float *h, *a, *y; WStat st; };
struct SaS { float *v, *o, *y; WStat st; int dv; };
struct CvS { float *u, *y, *a; WStat st; int chunk; };
struct NaS { float *u, *xs, *z, *o, *y; WStat st1, st2; int C; };

struct State {
  int magic, T, B;
  const float* x[12];
  float *qkp, *posp, *W, *nscales;
  const float* pos2;
  WStat kst, wst;
  float* pen_slot;
  hipEvent_t pen_ev;
  FfS ff[3];
  NaS na;
  SaS sa[2];
  CvS cv[2];
  int fm_fused;
  // column statistics of the firing Balancers' inputs, taken in FORWARD on the side stream (round 6):
  // sites 0-2 ff.post, 3-5 ff.hidden, 6-7 conv.bal1, 8-9 conv.bal2, 10 na.post, 11 na.bal, 12 bal1,
  // 13 bal2, 14 bal_keys; NULL = backward takes them itself.  bal_ev: recorded on the side stream after
  // the layer's last statistics launch; backward's stream waits for it once.
  float* bst[kBalSites];
  hipEvent_t bal_ev;
  int bal_fwd;
  uint64_t bal_seq;          // order of bal_ev among the process's records on the side stream
  // backward
  int wh_active[S2T_ZL_NWHITEN];
  int pen_active;
  float *dO[2], *dW0, *delta, *dqkp, *dpos, *g0, *d0m;
  S2tTnProblem probs[kMaxProb];
  int nprob;
  long bwd_off;             // arena offset where phase 1 of the backward stopped
};

struct Arena {
  float* base;
  long cap, off;
  float* alloc(long n) {
    const long a = (n + 63) & ~63L;              // 256-byte granules
    float* p = base ? base + off : reinterpret_cast<float*>(uintptr_t(0x10000) + (uintptr_t)off * 4);
    off += a;
    return p;
  }
};

struct Ctx {
  const S2tZipLayerDesc& d;
  const S2tZipLayerCall& c;
  State& s;
  Arena ar;
  hipStream_t st, side;
  bool dry;
  long R;
  // side-stream launches of this pass that wait for its ONE fork (side_run / flush_side below)
  std::vector<std::function<int()>> later;
};

int fail(int rc, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s failed with code %d", what, rc);
  return rc == 0 ? -1 : rc;
}
#define RUN(expr)                                 \
  do {                                            \
    if (!c.dry) {                                 \
      const int rc_ = (expr);                     \
      if (rc_ != 0) return fail(rc_, #expr);      \
    }                                             \
  } while (0)
#define TRY(expr)                 \
  do {                            \
    const int rc_ = (expr);       \
    if (rc_ != 0) return rc_;     \
  } while (0)
#define HIPRUN(expr)                                              \
  do {                                                            \
    if (!c.dry) {                                                 \
      const hipError_t e_ = (expr);                               \
      if (e_ != hipSuccess) return fail((int)e_, #expr);          \
    }                                                             \
  } while (0)

inline bool dec(const Ctx& c, int i) { return c.c.dec[i] != 0; }

// ---- streams
int fork_side(Ctx& c) {              // side stream ordered after the work enqueued so far on the main one
  RUN(s2t_stream_order((void*)c.st, (void*)c.side));
  return 0;
}

// A fork is an event record on the MAIN stream, i.e. a barrier packet of its own between two kernels: the
// queue runs nothing for 6-7 us around it (kernel trace of the C3 step: ~210 such gaps, 1.8 ms per step,
// every one of them at a fork).  Nothing on the main stream waits for the side stream's work inside a
// pass -- forward's statistics are read by backward, backward's parameter gradients by the optimizer -- so
// the side-stream launches of a pass are collected and leave behind ONE fork at its end (their operands
// live in the pass's workspace, which the caller keeps until the join).  S2T_SIDE_DEFER=0: a fork per site.
// (bits: 1 forward's statistics, 2 backward's parameter gradients, 4 backward's implied event waits)
int defer_bits() {
  static const int bits = [] { const char* e = getenv("S2T_SIDE_DEFER"); return e ? atoi(e) : 7; }();
  return bits;
}
template <class F>
int side_run(Ctx& c, int bit, F&& f) {   // f(): launches on c.side, ordered after the main stream's work so far
  if (defer_bits() & bit) {
    c.later.emplace_back(std::forward<F>(f));
    return 0;
  }
  TRY(fork_side(c));
  return f();
}
int flush_side(Ctx& c) {
  if (c.later.empty()) return 0;
  TRY(fork_side(c));
  for (auto& f : c.later) TRY(f());
  c.later.clear();
  return 0;
}

// ---- forward / data-gradient product of a Linear with its elementwise neighbours
// (zip_kernels.lt_matmul): mode 0: x (R,K) W (N,K)^T (+bias) -> (R,N); mode 1: x (R,N) W (N,K) -> (R,K);
// then (* act'(act_src)) (+ resid2) (+ resid_b); act2 1 | 2: out2 = SwooshL | R (out); act2 3: out2 =
// out + resid_b (out itself then excludes resid_b).  All (R, cols) operands dense.
struct Epi {
  const float* bias = nullptr;
  const float* resid2 = nullptr;
  const float* act_src = nullptr;
  int act_kind = 0;
  int act2 = 0;
  float* out2 = nullptr;
  const float* resid_b = nullptr;
  const S2tZlBal* bal = nullptr;     // Balancer on act_src, folded into the epilogue (s2t_gemm_x3p_bal)
  float* bal_stats = nullptr;        // its column statistics if forward took them (4096 floats), else NULL
};
constexpr float kSwOff[3] = {0.f, 4.0f, 1.0f};
constexpr float kSwC[3] = {0.f, 0.035f, 0.313261687f};

int plan_missing(const Ctx& c, int mode, long R, const S2tZlLin& L) {
  if (!c.c.x3p_on || R == 0) return 0;
  const unsigned short* pp = mode == 0 ? L.pf : L.pb;
  if (!pp) return 0;
  return g_plans.find(plan_key(plan_mode(mode), half_octave(R), L.N, L.K)) == g_plans.end();
}

int balancer_bwd(Ctx& c, const S2tZlBal& b, const float* x, long ldx, const float* g, long ldg, long R,
                 int C, float* out, long ldo, float act_off, const float* stats = nullptr);

int lt_matmul(Ctx& c, int mode, const float* x, long ldx, long R, const S2tZlLin& L, const Epi& e,
              float* out) {
  const S2tGemmClass cls(mode);        // the arithmetic of forward (0) / data-gradient (1) products
  const int cols = mode == 0 ? L.N : L.K, inner = mode == 0 ? L.K : L.N;
  const long n = R * cols;
  const unsigned short* pp = mode == 0 ? L.pf : L.pb;
  const bool fused = e.act_src || e.act2 || e.resid_b;
  bool own = false;
  int tile = 0;
  if (c.c.x3p_on && R > 0 && pp) {
    auto it = g_plans.find(plan_key(plan_mode(mode), half_octave(R), L.N, L.K));
    if (it == g_plans.end()) {
      if (!c.dry) return fail(-5, "lt_matmul: shape bucket not timed yet");
      own = true;
    } else if (it->second.t_own >= 0.0) {
      const Base& b = it->second;
      const double rc = (double)R * cols;
      const double pass_ms = 4.0e-3 + 12.0 * rc / 3.0e9;
      const int n_pass = (e.act_src != nullptr) + (e.act_src && e.resid2) + (e.resid_b != nullptr) +
                         (e.act2 == 1 || e.act2 == 2) + (e.bal != nullptr);
      const int n_ops = (e.act_src != nullptr) + (e.resid_b != nullptr) + (e.act2 != 0) + (e.bal != nullptr);
      const double cost_lt = b.t_lib * (fused ? 1.0 : (double)c.c.x3p_margin) + n_pass * pass_ms;
      const double cost_own = b.t_own + n_ops * 4.0 * rc / 3.0e9;
      own = cost_own < cost_lt;
      tile = b.tile;
    }
  }
  if (e.bal && !c.c.bal_epi) own = false;          // (A/B switch: plain product + the two-pass update)
  float* bstats = e.bal_stats ? e.bal_stats : ((e.bal && own) ? c.ar.alloc(4096) : nullptr);   // sums | squares | a | b
  if (own && !c.dry && e.bal) {
    if (cols <= 1024) {
      if (!e.bal_stats) {
        HIPRUN(hipMemsetAsync(bstats, 0, 2048 * sizeof(float), c.st));
        RUN(s2t_balancer_stats(e.act_src, cols, R, cols, bstats, (void*)c.st));
      }
      const int rc = s2t_gemm_x3p_bal(x, ldx, pp, cols, inner, out, cols, (int)R, e.resid2, cols, e.act_src, cols,
                                      e.act_kind, tile | (e.bal_stats ? S2T_X3P_BAL_COEF_READY : 0), bstats,
                                      e.bal->min_mean, e.bal->max_mean, e.bal->min_rms,
                                      e.bal->max_rms, e.bal->grad_scale, (void*)c.st);
      if (rc == 0) return 0;
      if (rc != -2) return fail(rc, "s2t_gemm_x3p_bal");
    }
  } else if (own && !c.dry) {
    const int rc = s2t_gemm_x3p(x, ldx, pp, cols, inner, out, cols, (int)R, e.bias, e.resid2, cols, e.act_src,
                                cols, e.act_kind, e.out2, cols, e.act2, e.resid_b, cols, tile, (void*)c.st);
    if (rc == 0) return 0;
    if (rc != -2) return fail(rc, "s2t_gemm_x3p");
  }
  // library path: bias + one residual in the GEMM, the rest as separate passes
  if (e.bal) {                        // plain product, then the Balancer's two-pass update through act'
    float* t = c.ar.alloc(n);
    Epi plain;
    TRY(lt_matmul(c, mode, x, ldx, R, L, plain, t));
    return balancer_bwd(c, *e.bal, e.act_src, cols, t, cols, R, cols, out, cols, kSwOff[e.act_kind], e.bal_stats);
  }
  float* y = out;
  float* tmp = nullptr;
  if (e.act_src) y = tmp = c.ar.alloc(n);
  if (!c.dry) {
    const float* res = e.act_src ? nullptr : e.resid2;
    int rc = s2t_linear_lt(mode, x, ldx, L.w, L.K, e.bias, res, cols, res ? 1.0f : 0.0f, y, cols, (int)R, L.N,
                           L.K, c.c.lt_ws, c.c.lt_ws_bytes, (void*)c.st);
    if (rc == -2)     // no library algorithm for this shape: our NT / NN kernel (the Python path takes ATen's)
      rc = s2t_gemm_f32(mode, x, ldx, L.w, L.K, y, cols, (int)R, cols, inner, e.bias, res, cols, nullptr, 0, 0, 0,
                        0, nullptr, 0, (void*)c.st);
    if (rc != 0) return fail(rc, "s2t_linear_lt");
  }
  if (e.act_src) {
    RUN(s2t_swoosh_bwd(e.act_src, tmp, out, n, kSwOff[e.act_kind], (void*)c.st));
    if (e.resid2) RUN(s2t_add_f32(out, e.resid2, out, n, (void*)c.st));
  }
  if (e.act2 == 3) {
    RUN(s2t_add_f32(out, e.resid_b, e.out2, n, (void*)c.st));
  } else {
    if (e.resid_b) RUN(s2t_add_f32(out, e.resid_b, out, n, (void*)c.st));
    if (e.act2) RUN(s2t_swoosh_fwd(out, e.out2, n, kSwOff[e.act2], kSwC[e.act2], (void*)c.st));
  }
  return 0;
}

// ---- batch of products of the nonlinear attention (zip_kernels.batched_matmul)
int bmm(Ctx& c, int mode, const float* a, const float* b, float* out, int n, int M, int N, int K) {
  if (c.dry) return 0;
  const int mn = M < N ? (M < K ? M : K) : (N < K ? N : K);
  if (c.c.bmm_own && mode != 2 && K % 4 == 0 && (mode == 0 || N % 4 == 0) && mn >= 4) {
    const long lda = K, sA = (long)M * K;
    const long ldb = mode == 0 ? K : N, sB = (long)N * K;
    const int rc = s2t_gemm_f32_batched(mode, a, lda, sA, b, ldb, sB, out, N, (long)M * N, M, N, K, n, (void*)c.st);
    if (rc == 0) return 0;
    if (rc != -2) return fail(rc, "s2t_gemm_f32_batched");
  }
  RUN(s2t_bmm_lt(mode, a, b, out, n, M, N, K, c.c.lt_ws, c.c.lt_ws_bytes, (void*)c.st));
  return 0;
}

// ---- Whiten statistics in forward (zip_kernels.WhitenStats): x^T x + column sums on the side
// stream, covariance / metric by one small kernel, the metric to a pinned host word
const S2tZlWhScratch* wh_scratch(const Ctx& c, int C) {
  for (int i = 0; i < c.c.nwh; ++i)
    if (c.c.wh[i].C == C) return &c.c.wh[i];
  return nullptr;
}

int whiten_stats(Ctx& c, WStat& s, const float* x, long ldx, long R, int C, int groups) {
  s.on = 1;
  s.G = groups;
  s.cg = C / groups;
  s.cov = c.ar.alloc((long)groups * s.cg * s.cg);
  s.mean = c.ar.alloc(C);
  s.scal = c.ar.alloc(4);
  // round-6 form: d metric / d cov, its bias row and dcov's bf16 pieces are taken NOW, on the statistics'
  // stream (they depend on x only): backward's chain is the penalty product on the pre-split-weight kernel
  // with the two norms in its epilogue, and the combining pass
  const S2tZlWhScratch* sc0 = wh_scratch(c, C);
  const bool fused = c.c.whiten_x3p == 2 && c.c.x3p_on && C >= 16 && (C & 7) == 0 && s.cg <= 1024 &&
                     R >= 4 && R * C * 4 < 0x7FFFFF00L && R * ldx * 4 < 0x7FFFFF00L && sc0 && sc0->tab;
  s.pieces = nullptr;
  if (fused) {
    s.dcov = c.ar.alloc((long)C * C);
    s.bias = c.ar.alloc(C);
    s.sums = c.ar.alloc(128);
    s.pieces = reinterpret_cast<unsigned short*>(c.ar.alloc((s2t_x3p_plane_elems(C, C) + 1) / 2));
  }
  // the penalty product itself depends on x only: on the statistics' stream too (backward then adds
  // ||g||^2 and combines: the product was 35 us per firing Whiten on the data-gradient chain)
  s.pg = (fused && c.c.whiten_fwd_pg) ? c.ar.alloc(R * C) : nullptr;
  if (c.dry) return 0;
  const S2tZlWhScratch* sc = wh_scratch(c, C);
  if (!sc) return fail(-1, "whiten_stats: no scratch for this channel count");
  if ((ldx & 3) || (C & 3) || R < 4 || (reinterpret_cast<uintptr_t>(x) & 15))
    return fail(-1, "whiten_stats: layout outside the TN kernel's rules");
  s.host = pinned_slot();
  s.ev = ring_event();
  if (!s.host || !s.ev) return fail(-1, "whiten_stats: pinned slot / event");
  const bool on_side = c.c.stats_side && c.side;
  auto launch = [&c, &s, sc, x, ldx, R, C, groups, on_side]() -> int {
    hipStream_t q = on_side ? c.side : c.st;
    float* xtx = sc->acc;
    float* colsum = sc->acc + (long)C * C;
    RUN(s2t_gemm_xtx(x, ldx, (int)R, C, s.cg, xtx, C, colsum, (void*)q));
    RUN(s2t_whiten_metric(xtx, colsum, R, groups, s.cg, s.cov, s.mean, s.scal, s.host, sc->ws, (void*)q));
    if (s.pieces) {
      RUN(s2t_whiten_prep(s.cov, s.mean, s.scal, s.G, s.cg, s.dcov, s.bias, s.sums, (void*)q));
      RUN(s2t_x3p_split(s.dcov, sc->tab, 1, sc->blocks, s.pieces, (void*)q));
      if (s.pg) {
        static const int pg_cls = [] { const char* e = getenv("S2T_WHITEN_PG_CLS"); return e ? atoi(e) : 1; }();
        const S2tGemmClass cls(pg_cls);
        const bool two = s2t_gemm_arith_of(pg_cls) == 2;
        const int tile = c.c.x3p_tile ? c.c.x3p_tile : ((C & 127) == 0 ? (two ? 2212 : 312) : (two ? 2221 : 321));
        const int rc = s2t_gemm_x3p_sq(x, ldx, s.pieces, C, C, s.pg, C, (int)R, s.bias, nullptr, 0, s.sums, tile, (void*)q);
        if (rc != 0) return fail(rc, "s2t_gemm_x3p_sq(whiten, forward)");
      }
    }
    // (backward waits for this event on the host before it reads the metric: everything above is then
    //  complete, whichever stream it ran on)
    HIPRUN(hipEventRecord(s.ev, q));
    return 0;
  };
  return on_side ? side_run(c, 1, launch) : launch();
}

// zip_kernels.whiten_backward: g (R,C) dense -> *out (g itself when the penalty is inactive)
int whiten_bwd(Ctx& c, int site, const S2tZlWh& w, WStat& s, const float* x, long ldx, long R, int C,
               const float* g, const float** out) {
  bool active = true;
  if (!c.dry) {
    if (hipEventSynchronize(s.ev) != hipSuccess) return fail(-1, "whiten_bwd: event");
    const float metric = *reinterpret_cast<volatile float*>(s.host);
    active = metric >= w.limit;
  }
  c.s.wh_active[site] = active ? 1 : 0;
  *out = g;
  if (!active) return 0;
  if (s.pieces && s.pg && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // pg and ||pg||^2 were taken in forward (the host wait on s.ev above covers them): ||g||^2, then combine
    float* o = c.ar.alloc(R * C);
    if (c.dry) return 0;
    RUN(s2t_sumsq64(g, R * C, s.sums, (void*)c.st));
    RUN(s2t_whiten_combine64(g, s.pg, R * C, w.grad_scale, s.sums, o, (void*)c.st));
    *out = o;
    return 0;
  }
  if (s.pieces && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    // (the prep launches were issued on the side stream in forward: the main stream joined it at the
    //  end of that pass, long before this call)
    float* pg = c.ar.alloc(R * C);
    float* o = c.ar.alloc(R * C);
    if (c.dry) return 0;
    // the penalty product is a data-gradient-like product (class D: two pieces by default -- measured:
    // every C3 gradient stays where the six-product step has it; it is the covariance x^T x
    // that needs the six products); S2T_WHITEN_PG_CLS=3 files it under the statistics
    static const int pg_cls = [] { const char* e = getenv("S2T_WHITEN_PG_CLS"); return e ? atoi(e) : 1; }();
    const S2tGemmClass cls(pg_cls);
    // block tile by the output width (the plan table has no bucket for these C x C products): 128-wide
    // column tiles where C is a multiple of 128, 64-wide otherwise; the 32-deep LDS-DMA form with two pieces
    const bool two = s2t_gemm_arith_of(pg_cls) == 2;
    const int tile = c.c.x3p_tile ? c.c.x3p_tile : ((C & 127) == 0 ? (two ? 2212 : 312) : (two ? 2221 : 321));
    const int rc = s2t_gemm_x3p_sq(x, ldx, s.pieces, C, C, pg, C, (int)R, s.bias, g, C, s.sums, tile,
                                   (void*)c.st);
    if (rc != 0) return fail(rc, "s2t_gemm_x3p_sq(whiten)");      // (the shape rules were checked in forward)
    RUN(s2t_whiten_combine64(g, pg, R * C, w.grad_scale, s.sums, o, (void*)c.st));
    *out = o;
    return 0;
  }
  float* dcov = c.ar.alloc((long)C * C);
  float* bias = c.ar.alloc(C);
  float* sums = c.ar.alloc(2);
  float* pg = c.ar.alloc(R * C);
  float* o = c.ar.alloc(R * C);
  if (c.dry) return 0;
  RUN(s2t_whiten_dcov(s.cov, s.mean, s.scal, s.G, s.cg, dcov, bias, sums, (void*)c.st));
  // the penalty's product x dcov: class S (statistics) -- S2T_WHITEN_PG_CLS=1 files it under the data
  // gradients instead (experiment: which of the two statistics products needs the six-product form)
  static const int pg_cls = [] { const char* e = getenv("S2T_WHITEN_PG_CLS"); return e ? atoi(e) : 3; }();
  const S2tGemmClass cls(pg_cls);
  if (c.c.whiten_sq) {               // the two norms of (g, pg) from the product's own epilogue
    const int rc = s2t_gemm_f32_sq(1, x, ldx, dcov, C, pg, C, (int)R, C, C, bias, g, C, sums, (void*)c.st);
    if (rc == 0) {
      RUN(s2t_whiten_combine(g, pg, R * C, w.grad_scale, sums, o, (void*)c.st));
      *out = o;
      return 0;
    }
    if (rc != -2) return fail(rc, "s2t_gemm_f32_sq(whiten)");
  }
  RUN(s2t_gemm_f32(1, x, ldx, dcov, C, pg, C, (int)R, C, C, bias, nullptr, 0, nullptr, 0, 0, 0, 0, nullptr, 0,
                   (void*)c.st));
  RUN(s2t_whiten_apply(g, pg, R * C, w.grad_scale, sums, o, (void*)c.st));
  *out = o;
  return 0;
}

int balancer_bwd(Ctx& c, const S2tZlBal& b, const float* x, long ldx, const float* g, long ldg, long R,
                 int C, float* out, long ldo, float act_off, const float* stats) {
  if (c.dry) return 0;
  if (stats) {                       // forward took the column statistics (side stream): the update alone
    RUN(s2t_balancer_apply(x, ldx, g, ldg, R, C, b.min_mean, b.max_mean, b.min_rms, b.max_rms, b.grad_scale, out,
                           ldo, stats, act_off, (void*)c.st));
    return 0;
  }
  const int parity = s2t_balancer_next_parity();
  RUN(s2t_balancer_bwd(x, ldx, g, ldg, R, C, b.min_mean, b.max_mean, b.min_rms, b.max_rms, b.grad_scale,
                       out, ldo, c.c.bal_ws, parity, act_off, (void*)c.st));
  return 0;
}

int wgrad(Ctx& c, const S2tZlLin& L, const float* g2, long ldg, const float* a2, long lda, long R) {
  if (c.s.nprob >= kMaxProb) return fail(-1, "wgrad: too many problems");
  S2tTnProblem& q = c.s.probs[c.s.nprob++];
  q.A = g2;
  q.lda = ldg;
  q.B = a2;
  q.ldb = lda;
  q.C = L.gw;
  q.ldc = L.K;
  q.M = L.N;
  q.N = L.K;
  q.K = (int)R;
  q.colsum = L.gb;
  q.alpha = 1.0f;
  return 0;
}

int copy2d(Ctx& c, float* dst, long ldd, const float* src, long lds, int cols, long rows) {
  HIPRUN(hipMemcpy2DAsync(dst, ldd * 4, src, lds * 4, (size_t)cols * 4, (size_t)rows, hipMemcpyDeviceToDevice, c.st));
  return 0;
}

// Column statistics of a firing Balancer's input where the input is produced: the side stream, ordered
// after the producer; backward then runs only the update on the data-gradient chain (the statistics pass
// was 54 launches per step on that chain: 0.8 ms).  n = 4096 for a Balancer whose update may ride in a
// data-gradient GEMM's epilogue (s2t_gemm_x3p_bal writes its coefficients into the second half).
int bal_stats_fwd(Ctx& c, int site, const float* x, long ldx, long R, int C, int n = 2048,
                  const S2tZlBal* coef = nullptr) {
  c.s.bst[site] = nullptr;
  if (!c.c.bal_fwd_side || (!c.dry && !c.side) || C > 1024) return 0;   // (the dry run has no streams: it sizes for the side form)
  float* st = c.ar.alloc(n);
  c.s.bst[site] = st;
  if (c.dry) return 0;
  c.s.bal_fwd = 1;
  return side_run(c, 1, [&c, st, x, ldx, R, C, coef]() -> int {
    HIPRUN(hipMemsetAsync(st, 0, 2048 * sizeof(float), c.side));
    RUN(s2t_balancer_stats(x, ldx, R, C, st, (void*)c.side));
    if (coef)                        // (the epilogue form's per-column coefficients, off the data-gradient chain too)
      RUN(s2t_balancer_coef(st, C, R, coef->min_mean, coef->max_mean, coef->min_rms, coef->max_rms, coef->grad_scale,
                            (void*)c.side));
    return 0;
  });
}

// the module's last projection: plain (+ residual) or the module's own output as well
int out_proj(Ctx& c, const S2tZlLin& L, const float* a, long R, const float* x_in, bool need_y, float** y,
             float** out) {
  const int D = c.d.D;
  *out = c.ar.alloc(R * D);
  Epi e;
  e.bias = L.b;
  if (!need_y) {
    *y = nullptr;
    e.resid2 = x_in;
    return lt_matmul(c, 0, a, L.K, R, L, e, *out);
  }
  *y = c.ar.alloc(R * D);
  e.act2 = 3;
  e.out2 = *out;
  e.resid_b = x_in;
  return lt_matmul(c, 0, a, L.K, R, L, e, *y);
}

// ------------------------------------------------------------------------------------- forward
int ff_fwd(Ctx& c, int i, int d0, const float* x_in, const float** x_out) {
  const S2tZlFf& m = c.d.ff[i];
  FfS& sv = c.s.ff[i];
  const bool fw = dec(c, d0 + 1), fp = dec(c, d0 + 2);
  const long R = c.R;
  const int F = m.in.N;
  sv.h = c.ar.alloc(R * F);
  sv.a = c.ar.alloc(R * F);
  Epi e;
  e.bias = m.in.b;
  e.act2 = 1;
  e.out2 = sv.a;
  TRY(lt_matmul(c, 0, x_in, c.d.D, R, m.in, e, sv.h));
  c.s.bst[3 + i] = nullptr;
  if (dec(c, d0)) TRY(bal_stats_fwd(c, 3 + i, sv.h, F, R, F, 4096, &m.hidden));
  float* out;
  TRY(out_proj(c, m.out, sv.a, R, x_in, fw || fp, &sv.y, &out));
  c.s.bst[i] = nullptr;
  if (fp) TRY(bal_stats_fwd(c, i, sv.y, c.d.D, R, c.d.D));
  sv.st.on = 0;
  if (fw) TRY(whiten_stats(c, sv.st, sv.y, c.d.D, R, c.d.D, m.out_wh.groups));
  *x_out = out;
  return 0;
}

int sa_fwd(Ctx& c, int i, int dfw, const float* x_in, const float** x_out) {
  const S2tZlSa& m = c.d.sa[i];
  SaS& sv = c.s.sa[i];
  const bool fw = dec(c, dfw);
  const long R = c.R;
  const int HD = m.in.N, H = c.d.H;
  sv.dv = HD / H;
  sv.v = c.ar.alloc(R * HD);
  Epi e;
  e.bias = m.in.b;
  TRY(lt_matmul(c, 0, x_in, c.d.D, R, m.in, e, sv.v));
  sv.o = c.ar.alloc(R * HD);
  RUN(s2t_attn_apply(c.s.W, sv.v, c.c.T, c.c.B, H, sv.dv, 0, sv.o, (void*)c.st));
  float* out;
  TRY(out_proj(c, m.out, sv.o, R, x_in, fw, &sv.y, &out));
  sv.st.on = 0;
  if (fw) TRY(whiten_stats(c, sv.st, sv.y, c.d.D, R, c.d.D, m.wh.groups));
  *x_out = out;
  return 0;
}

int conv_fwd(Ctx& c, int i, int d0, const float* x_in, const float** x_out) {
  const S2tZlConv& m = c.d.cv[i];
  CvS& sv = c.s.cv[i];
  const bool fw = dec(c, d0 + 2);
  const long R = c.R;
  const int D = c.d.D, T = c.c.T, B = c.c.B;
  if (c.c.chunk_size >= 0 && !m.causal) return fail(-1, "conv module: chunk_size needs causal=True");
  sv.u = c.ar.alloc(R * 2 * D);
  Epi e;
  e.bias = m.in.b;
  TRY(lt_matmul(c, 0, x_in, D, R, m.in, e, sv.u));
  c.s.bst[6 + i] = c.s.bst[8 + i] = nullptr;
  if (dec(c, d0)) TRY(bal_stats_fwd(c, 6 + i, sv.u + D, 2 * D, R, D));
  sv.chunk = (c.c.chunk_size < 0 || c.c.chunk_size > T) ? T : c.c.chunk_size;
  sv.y = c.ar.alloc(R * D);
  sv.a = c.ar.alloc(R * D);
  // (SwooshR(y) leaves with the conv's output tile: no activation pass)
  RUN(s2t_zipconv_fwd_act(sv.u, 2 * D, D, c.c.k8, T, B, D, m.K, sv.chunk, m.wc, m.bc, m.wk, m.bk, m.scale, sv.y,
                          sv.a, 2, (void*)c.st));
  if (dec(c, d0 + 1)) TRY(bal_stats_fwd(c, 8 + i, sv.y, D, R, D, 4096, &m.bal2));
  sv.st.on = 0;
  if (fw) TRY(whiten_stats(c, sv.st, sv.y, D, R, D, m.wh.groups));
  float* out = c.ar.alloc(R * D);
  Epi e2;
  e2.bias = m.out.b;
  e2.resid2 = x_in;
  TRY(lt_matmul(c, 0, sv.a, D, R, m.out, e2, out));
  *x_out = out;
  return 0;
}

int na_fwd(Ctx& c, const float* x_in, const float** x_out) {
  const S2tZlNa& m = c.d.na;
  NaS& sv = c.s.na;
  const bool fw1 = dec(c, 8), fw2 = dec(c, 9), fp = dec(c, 10);
  const long R = c.R;
  const int D = c.d.D, T = c.c.T, B = c.c.B;
  const int C = m.in.N / 3;
  sv.C = C;
  sv.u = c.ar.alloc(R * 3 * C);
  Epi e;
  e.bias = m.in.b;
  TRY(lt_matmul(c, 0, x_in, D, R, m.in, e, sv.u));
  c.s.bst[10] = c.s.bst[11] = nullptr;
  if (dec(c, 7)) TRY(bal_stats_fwd(c, 11, sv.u, 3 * C, R, C));
  sv.xs = c.ar.alloc(R * C);
  RUN(s2t_nonlin_gate_fwd(sv.u, T, B, C, sv.xs, (void*)c.st));
  sv.z = c.ar.alloc(R * C);
  { const S2tGemmClass cls(0); TRY(bmm(c, 1, c.s.W, sv.xs, sv.z, B, T, C, T)); }   // W0 @ x
  sv.o = c.ar.alloc(R * C);
  RUN(s2t_nonlin_out_fwd(sv.z, sv.u, T, B, C, sv.o, (void*)c.st));
  sv.st1.on = 0;
  if (fw1) TRY(whiten_stats(c, sv.st1, sv.u + C, 3 * C, R, C, m.wh1.groups));
  float* out;
  TRY(out_proj(c, m.out, sv.o, R, x_in, fw2 || fp, &sv.y, &out));
  if (fp) TRY(bal_stats_fwd(c, 10, sv.y, D, R, D));
  sv.st2.on = 0;
  if (fw2) TRY(whiten_stats(c, sv.st2, sv.y, D, R, D, m.wh2.groups));
  *x_out = out;
  return 0;
}

int layer_fwd(Ctx& c) {
  const S2tZipLayerDesc& d = c.d;
  State& s = c.s;
  const int T = c.c.T, B = c.c.B, D = d.D, H = d.H, qd = d.qd, pd = d.pd;
  const long R = c.R;
  const int Dp = d.attn_in.N;
  s.magic = kMagic;
  s.T = T;
  s.B = B;
  s.nprob = 0;
  s.x[0] = c.c.x0;
  s.bal_fwd = 0;
  for (int i = 0; i < kBalSites; ++i) s.bst[i] = nullptr;
  // attention weights (zipformer.py:1966-2066)
  s.qkp = c.ar.alloc(R * Dp);
  {
    Epi e;
    e.bias = d.attn_in.b;
    TRY(lt_matmul(c, 0, s.x[0], D, R, d.attn_in, e, s.qkp));
  }
  if (dec(c, 0)) TRY(bal_stats_fwd(c, 14, s.qkp + H * qd, Dp, R, H * qd));
  s.kst.on = 0;
  if (dec(c, 1)) TRY(whiten_stats(c, s.kst, s.qkp + H * qd, Dp, R, H * qd, d.wh_keys.groups));
  s.posp = nullptr;
  s.pos2 = c.c.pos;
  if (dec(c, 2)) {
    s.posp = c.ar.alloc((long)(2 * T - 1) * H * pd);
    Epi e;
    TRY(lt_matmul(c, 0, c.c.pos, d.pos_dim, 2 * T - 1, d.attn_pos, e, s.posp));
  }
  s.W = c.ar.alloc((long)H * B * T * T);
  s.pen_slot = nullptr;
  if (dec(c, 3)) {
    if (!c.dry) {
      s.pen_slot = pinned_slot();
      s.pen_ev = ring_event();
      if (!s.pen_slot || !s.pen_ev) return fail(-1, "penalty flag slot");
      *s.pen_slot = 0.f;
    }
    RUN(s2t_relpos_attn_fwd_flag(s.qkp, s.posp, c.c.k8, c.c.a8, T, B, H, qd, pd, s.W, kPenLimit, s.pen_slot,
                                 (void*)c.st));
    HIPRUN(hipEventRecord(s.pen_ev, c.st));
  } else {
    RUN(s2t_relpos_attn_fwd(s.qkp, s.posp, c.c.k8, c.c.a8, T, B, H, qd, pd, s.W, (void*)c.st));
  }
  TRY(ff_fwd(c, 0, 4, s.x[0], &s.x[1]));
  TRY(na_fwd(c, s.x[1], &s.x[2]));
  TRY(sa_fwd(c, 0, 11, s.x[2], &s.x[3]));
  TRY(conv_fwd(c, 0, 12, s.x[3], &s.x[4]));
  TRY(ff_fwd(c, 1, 15, s.x[4], &s.x[5]));
  float* x6 = c.ar.alloc(R * D);
  RUN(s2t_bypass_fwd(s.x[0], s.x[5], d.byp_mid.x, R, D, x6, (void*)c.st));
  s.x[6] = x6;
  TRY(sa_fwd(c, 1, 19, s.x[6], &s.x[7]));
  TRY(conv_fwd(c, 1, 20, s.x[7], &s.x[8]));
  TRY(ff_fwd(c, 2, 23, s.x[8], &s.x[9]));
  if (dec(c, 26)) TRY(bal_stats_fwd(c, 12, s.x[9], D, R, D));
  s.nscales = c.ar.alloc(R);
  s.x[10] = nullptr;                     // norm(x9) is never stored: backward recomputes x9 * nscales
  s.x[11] = c.c.out;
  // BiasNorm + the layer's bypass in one pass; the stack's feature mask rides in it unless a
  // gradient-shaping op of this call needs the unmasked output
  s.fm_fused = c.c.fm != nullptr && !(dec(c, 30) || dec(c, 29));
  RUN(s2t_norm_bypass_fwd(s.x[9], d.norm_bias.x, d.norm_ls.x, s.x[0], d.byp.x, s.fm_fused ? c.c.fm : nullptr, B, R,
                          D, c.c.out, s.nscales, (void*)c.st));
  if (dec(c, 29)) TRY(bal_stats_fwd(c, 13, c.c.out, D, R, D));
  s.wst.on = 0;
  if (dec(c, 30)) TRY(whiten_stats(c, s.wst, c.c.out, D, R, D, d.wh_out.groups));
  TRY(flush_side(c));                    // the pass's side-stream launches, behind one fork
  if (s.bal_fwd && !c.dry) {             // backward's stream waits for this (layer_bwd)
    s.bal_ev = ring_event();
    if (!s.bal_ev) return fail(-1, "balancer statistics event");
    HIPRUN(hipEventRecord(s.bal_ev, c.side));
    s.bal_seq = ++g_bal_seq;
  }
  return 0;
}

// ------------------------------------------------------------------------------------ backward
int ff_bwd(Ctx& c, int i, int d0, const float* x_in, const float* g, const float** g_out) {
  const S2tZlFf& m = c.d.ff[i];
  FfS& sv = c.s.ff[i];
  const bool fb = dec(c, d0), fw = dec(c, d0 + 1), fp = dec(c, d0 + 2);
  const long R = c.R;
  const int D = c.d.D, F = m.in.N;
  const int site = i == 0 ? 1 : (i == 1 ? 6 : 9);
  const float* gy = g;
  if (fp) {
    float* o = c.ar.alloc(R * D);
    TRY(balancer_bwd(c, m.post, sv.y, D, gy, D, R, D, o, D, -1.0f, c.s.bst[i]));
    gy = o;
  }
  if (fw) TRY(whiten_bwd(c, site, m.out_wh, sv.st, sv.y, D, R, D, gy, &gy));
  TRY(wgrad(c, m.out, gy, D, sv.a, F, R));
  float* dh = c.ar.alloc(R * F);
  if (fb) {                                  // Swoosh backward AND the Balancer's update in the dgrad epilogue
    Epi e;
    e.act_src = sv.h;
    e.act_kind = 1;
    e.bal = &m.hidden;
    e.bal_stats = c.s.bst[3 + i];
    TRY(lt_matmul(c, 1, gy, D, R, m.out, e, dh));
  } else {                                   // ... or in the data-gradient GEMM's epilogue
    Epi e;
    e.act_src = sv.h;
    e.act_kind = 1;
    TRY(lt_matmul(c, 1, gy, D, R, m.out, e, dh));
  }
  TRY(wgrad(c, m.in, dh, F, x_in, D, R));
  float* gx = c.ar.alloc(R * D);
  Epi e;
  e.resid2 = g;
  TRY(lt_matmul(c, 1, dh, F, R, m.in, e, gx));
  *g_out = gx;
  return 0;
}

int sa_bwd(Ctx& c, int i, int dfw, const float* x_in, const float* g, const float** g_out) {
  const S2tZlSa& m = c.d.sa[i];
  SaS& sv = c.s.sa[i];
  const bool fw = dec(c, dfw);
  const long R = c.R;
  const int D = c.d.D, HD = m.in.N, H = c.d.H;
  const float* gy = g;
  if (fw) TRY(whiten_bwd(c, i == 0 ? 4 : 7, m.wh, sv.st, sv.y, D, R, D, g, &gy));
  TRY(wgrad(c, m.out, gy, D, sv.o, HD, R));
  float* dO = c.ar.alloc(R * HD);
  {
    Epi e;
    TRY(lt_matmul(c, 1, gy, D, R, m.out, e, dO));
  }
  float* dV = c.ar.alloc(R * HD);
  RUN(s2t_attn_apply(c.s.W, dO, c.c.T, c.c.B, H, sv.dv, 1, dV, (void*)c.st));
  // pairs are collected in backward order: self_attn2 first (zip_layer.py `pairs`)
  c.s.dO[i == 1 ? 0 : 1] = dO;
  TRY(wgrad(c, m.in, dV, HD, x_in, D, R));
  float* gx = c.ar.alloc(R * D);
  Epi e;
  e.resid2 = g;
  TRY(lt_matmul(c, 1, dV, HD, R, m.in, e, gx));
  *g_out = gx;
  return 0;
}

int conv_bwd(Ctx& c, int i, int d0, const float* x_in, const float* g, const float** g_out) {
  const S2tZlConv& m = c.d.cv[i];
  CvS& sv = c.s.cv[i];
  const bool fb1 = dec(c, d0), fb2 = dec(c, d0 + 1), fw = dec(c, d0 + 2);
  const long R = c.R;
  const int D = c.d.D, T = c.c.T, B = c.c.B;
  TRY(wgrad(c, m.out, g, D, sv.a, D, R));
  const float* dy;
  if (fb2 && !fw) {                          // Swoosh backward AND the Balancer's update in the dgrad epilogue
    float* o = c.ar.alloc(R * D);
    Epi e;
    e.act_src = sv.y;
    e.act_kind = 2;
    e.bal = &m.bal2;
    e.bal_stats = c.s.bst[8 + i];
    TRY(lt_matmul(c, 1, g, D, R, m.out, e, o));
    dy = o;
  } else {                                   // ... or in the data-gradient GEMM's epilogue
    float* t = c.ar.alloc(R * D);
    Epi e;
    e.act_src = sv.y;
    e.act_kind = 2;
    TRY(lt_matmul(c, 1, g, D, R, m.out, e, t));
    dy = t;
    if (fw) TRY(whiten_bwd(c, i == 0 ? 5 : 8, m.wh, sv.st, sv.y, D, R, D, dy, &dy));
    if (fb2) {
      float* o = c.ar.alloc(R * D);
      TRY(balancer_bwd(c, m.bal2, sv.y, D, dy, D, R, D, o, D, -1.0f, c.s.bst[8 + i]));
      dy = o;
    }
  }
  float* du = c.ar.alloc(R * 2 * D);
  const long wsn = s2t_zipconv_bwd_workspace_floats(T, B, D, m.K);
  float* ws = c.ar.alloc(wsn);
  if (c.c.conv_w_side && c.side && !c.c.conv_fused) {
    RUN(s2t_zipconv_bwd_data(sv.u, 2 * D, D, c.c.k8, T, B, D, m.K, sv.chunk, m.wc, m.wk, m.bk, m.scale, dy, du,
                             (void*)c.st));
    if (!c.dry)
      TRY(side_run(c, 2, [&c, &m, &sv, dy, ws, T, B, D]() -> int {
        RUN(s2t_zipconv_bwd_params(sv.u, 2 * D, D, c.c.k8, T, B, D, m.K, sv.chunk, m.wc, m.wk, m.bk, m.scale, dy,
                                   m.gwc, m.gbc, m.gwk, m.gbk, m.gscale, ws, (void*)c.side));
        return 0;
      }));
  } else {
    RUN(s2t_zipconv_bwd(sv.u, 2 * D, D, c.c.k8, T, B, D, m.K, sv.chunk, m.wc, m.wk, m.bk, m.scale, dy, du, m.gwc,
                        m.gbc, m.gwk, m.gbk, m.gscale, ws, (void*)c.st));
  }
  if (fb1) TRY(balancer_bwd(c, m.bal1, sv.u + D, 2 * D, du + D, 2 * D, R, D, du + D, 2 * D, -1.0f, c.s.bst[6 + i]));
  TRY(wgrad(c, m.in, du, 2 * D, x_in, D, R));
  float* gx = c.ar.alloc(R * D);
  Epi e;
  e.resid2 = g;
  TRY(lt_matmul(c, 1, du, 2 * D, R, m.in, e, gx));
  *g_out = gx;
  return 0;
}

int na_bwd(Ctx& c, const float* x_in, const float* g, const float** g_out) {
  const S2tZlNa& m = c.d.na;
  NaS& sv = c.s.na;
  const bool fb = dec(c, 7), fw1 = dec(c, 8), fw2 = dec(c, 9), fp = dec(c, 10);
  const long R = c.R;
  const int D = c.d.D, T = c.c.T, B = c.c.B, C = sv.C;
  const float* gy = g;
  if (fp) {
    float* o = c.ar.alloc(R * D);
    TRY(balancer_bwd(c, m.post, sv.y, D, gy, D, R, D, o, D, -1.0f, c.s.bst[10]));
    gy = o;
  }
  if (fw2) TRY(whiten_bwd(c, 3, m.wh2, sv.st2, sv.y, D, R, D, gy, &gy));
  TRY(wgrad(c, m.out, gy, D, sv.o, C, R));
  float* dout = c.ar.alloc(R * C);
  {
    Epi e;
    TRY(lt_matmul(c, 1, gy, D, R, m.out, e, dout));
  }
  float* dz = c.ar.alloc(R * C);
  float* du = c.ar.alloc(R * 3 * C);
  RUN(s2t_nonlin_out_bwd(dout, sv.z, sv.u, T, B, C, dz, du, (void*)c.st));
  float* dxs = c.ar.alloc(R * C);
  const S2tGemmClass cls_d(1);
  TRY(bmm(c, 2, c.s.W, dz, dxs, B, T, C, T));                     // W0^T @ dz
  c.s.dW0 = c.ar.alloc((long)B * T * T);
  TRY(bmm(c, 0, dz, sv.xs, c.s.dW0, B, T, T, C));                 // dz @ x^T
  RUN(s2t_nonlin_gate_bwd(dxs, sv.u, T, B, C, du, (void*)c.st));
  if (fb) TRY(balancer_bwd(c, m.bal, sv.u, 3 * C, du, 3 * C, R, C, du, 3 * C, -1.0f, c.s.bst[11]));
  if (fw1) {
    float* gc = c.ar.alloc(R * C);
    TRY(copy2d(c, gc, C, du + C, 3 * C, C, R));
    const float* o;
    TRY(whiten_bwd(c, 2, m.wh1, sv.st1, sv.u + C, 3 * C, R, C, gc, &o));
    if (o != gc) TRY(copy2d(c, du + C, 3 * C, o, C, C, R));
  }
  TRY(wgrad(c, m.in, du, 3 * C, x_in, D, R));
  float* gx = c.ar.alloc(R * D);
  Epi e;
  e.resid2 = g;
  TRY(lt_matmul(c, 1, du, 3 * C, R, m.in, e, gx));
  *g_out = gx;
  return 0;
}

// phase 0: everything; 1: up to (excluding) the attention-weights backward; 2: the rest, with dqkp /
// dpos supplied by the caller in the buffers phase 1 allocated
int layer_bwd(Ctx& c, int phase) {
  const S2tZipLayerDesc& d = c.d;
  State& s = c.s;
  const int T = c.c.T, B = c.c.B, D = d.D, H = d.H, qd = d.qd, pd = d.pd;
  const long R = c.R;
  const int Dp = d.attn_in.N;
  if (phase != 2) {
    s.nprob = 0;
    for (int i = 0; i < S2T_ZL_NWHITEN; ++i) s.wh_active[i] = -1;
    s.pen_active = 0;
    // forward's Balancer statistics (side stream).  The side stream runs in order: once this stream has
    // waited for a LATER layer's event -- backward visits the layers in reverse -- the wait (a barrier packet
    // of its own) is implied
    if (s.bal_fwd && !c.dry && !((defer_bits() & 4) && g_bal_waited_on == (void*)c.st && s.bal_seq <= g_bal_waited)) {
      HIPRUN(hipStreamWaitEvent(c.st, s.bal_ev, 0));
      g_bal_waited_on = (void*)c.st;
      g_bal_waited = s.bal_seq;
    }
    if (dec(c, 3) && !c.dry) {
      if (hipEventSynchronize(s.pen_ev) != hipSuccess) return fail(-1, "penalty flag event");
      s.pen_active = *reinterpret_cast<volatile float*>(s.pen_slot) != 0.f;
    }
    const float* g = c.c.g;
    const float* x11 = s.x[11];
    if (dec(c, 30)) TRY(whiten_bwd(c, 10, d.wh_out, s.wst, x11, D, R, D, g, &g));
    if (dec(c, 29)) {
      float* o = c.ar.alloc(R * D);
      TRY(balancer_bwd(c, d.bal2, x11, D, g, D, R, D, o, D, -1.0f, s.bst[13]));
      g = o;
    }
    // per-channel parameter gradients: [bypass scale | bypass_mid scale | norm bias | log_scale]
    float* acc = c.c.layer_acc;
    float* d0 = c.ar.alloc(R * D);
    float* g9w = c.ar.alloc(R * D);
    RUN(s2t_norm_bypass_bwd(s.x[9], d.norm_bias.x, s.nscales, s.x[0], d.byp.x, g, s.fm_fused ? c.c.fm : nullptr, B,
                            R, D, g9w, d0, acc, acc + 2 * D, acc + 3 * D, (void*)c.st));
    const float* g9 = g9w;
    if (dec(c, 26)) {
      float* o = c.ar.alloc(R * D);
      TRY(balancer_bwd(c, d.bal1, s.x[9], D, g9, D, R, D, o, D, -1.0f, s.bst[12]));
      g9 = o;
    }
    const float *g8, *g7, *g6, *g4, *g3, *g2, *g1, *g0;
    TRY(ff_bwd(c, 2, 23, s.x[8], g9, &g8));
    TRY(conv_bwd(c, 1, 20, s.x[7], g8, &g7));
    TRY(sa_bwd(c, 1, 19, s.x[6], g7, &g6));
    float* d0m = c.ar.alloc(R * D);
    float* g5 = c.ar.alloc(R * D);
    RUN(s2t_bypass_bwd_acc(s.x[0], s.x[5], d.byp_mid.x, g6, d0, R, D, d0m, g5, acc + D, (void*)c.st));
    {
      S2tCommit it[4] = {
          {d.byp.x, acc, d.byp.grad, d.byp.lo, d.byp.hi, dec(c, 28) ? 1 : 0, (long)D},
          {d.byp_mid.x, acc + D, d.byp_mid.grad, d.byp_mid.lo, d.byp_mid.hi, dec(c, 18) ? 1 : 0, (long)D},
          {d.norm_bias.x, acc + 2 * D, d.norm_bias.grad, 0.f, 0.f, 0, (long)D},
          {d.norm_ls.x, acc + 3 * D, d.norm_ls.grad, d.norm_ls.lo, d.norm_ls.hi, dec(c, 27) ? 1 : 0, 1L}};
      RUN(s2t_param_grad_commit_n(4, it, (void*)c.st));
    }
    TRY(ff_bwd(c, 1, 15, s.x[4], g5, &g4));
    TRY(conv_bwd(c, 0, 12, s.x[3], g4, &g3));
    TRY(sa_bwd(c, 0, 11, s.x[2], g3, &g2));
    TRY(na_bwd(c, s.x[1], g2, &g1));
    TRY(ff_bwd(c, 0, 4, s.x[0], g1, &g0));
    // attention weights: delta from the consumers, then dS -> dq, dk, dp, dpos
    s.delta = c.ar.alloc((long)H * B * T);
    RUN(s2t_attn_delta_pairs(s.W, s.dW0, s.dO[0], s.sa[1].o, s.sa[1].dv, s.dO[1], s.sa[0].o, s.sa[0].dv, T, B, H,
                             s.delta, (void*)c.st));
    s.dqkp = c.ar.alloc(R * Dp);
    s.dpos = s.posp ? c.ar.alloc((long)(2 * T - 1) * H * pd) : nullptr;
    s.g0 = const_cast<float*>(g0);
    s.d0m = d0m;
    if (phase == 1 || (phase == 0 && s.pen_active)) {
      s.bwd_off = c.ar.off;
      TRY(flush_side(c));
      return s.pen_active ? 1 : 0;
    }
    float* aws = nullptr;
    if (s.posp) aws = c.ar.alloc(s2t_relpos_attn_bwd_workspace_floats(T, B, H, pd));
    RUN(s2t_relpos_attn_bwd(s.qkp, s.posp, c.c.k8, c.c.a8, T, B, H, qd, pd, s.W, nullptr, s.dW0, s.dO[0],
                            s.sa[1].v, s.sa[1].dv, s.dO[1], s.sa[0].v, s.sa[0].dv, 1, s.delta, s.dqkp, s.dpos, aws,
                            (void*)c.st));
  }
  float* dqkp = s.dqkp;
  if (dec(c, 1) || dec(c, 0)) {
    const int Ck = H * qd;
    float* gk = c.ar.alloc(R * Ck);
    TRY(copy2d(c, gk, Ck, dqkp + Ck, Dp, Ck, R));
    const float* cur = gk;
    if (dec(c, 1)) TRY(whiten_bwd(c, 0, d.wh_keys, s.kst, s.qkp + Ck, Dp, R, Ck, cur, &cur));
    if (dec(c, 0)) {
      float* o = c.ar.alloc(R * Ck);
      TRY(balancer_bwd(c, d.bal_keys, s.qkp + Ck, Dp, cur, Ck, R, Ck, o, Ck, -1.0f, s.bst[14]));
      cur = o;
    }
    TRY(copy2d(c, dqkp + Ck, Dp, cur, Ck, Ck, R));
  }
  if (s.dpos) TRY(wgrad(c, d.attn_pos, s.dpos, H * pd, s.pos2, d.pos_dim, 2 * T - 1));
  TRY(wgrad(c, d.attn_in, dqkp, Dp, s.x[0], D, R));
  {
    Epi e;
    e.resid2 = s.g0;
    e.resid_b = s.d0m;
    TRY(lt_matmul(c, 1, dqkp, Dp, R, d.attn_in, e, c.c.gx));
  }
  if (!c.dry && s.nprob > 0) {
    if (c.c.wgrad_side && c.side) {
      TRY(side_run(c, 2, [&c, &s]() -> int {
        RUN(s2t_gemm_tn_grouped(s.nprob, s.probs, (void*)c.side));
        return 0;
      }));
    } else {
      RUN(s2t_gemm_tn_grouped(s.nprob, s.probs, (void*)c.st));
    }
  }
  return flush_side(c);
}

}  // namespace

extern "C" {

long s2t_zip_layer_state_bytes(void) { return (long)sizeof(State); }

void* s2t_zip_layer_error(void) { return (void*)g_err; }

int s2t_zl_plan_put(int mode, int half_oct, int N, int K, double t_lib_ms, double t_own_ms, int tile) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_plans[plan_key(mode, half_oct, N, K)] = Base{t_lib_ms, t_own_ms, tile};
  return 0;
}
int s2t_zl_plan_clear(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  g_plans.clear();
  return 0;
}
long s2t_zl_plan_count(void) { return (long)g_plans.size(); }

// floats of workspace the forward (backward = 0) or backward (1) pass of this call may need: a dry
// run of the same code with every decision of `call` as given (pass all ones for the bound) and
// every Whiten penalty active
long s2t_zip_layer_ws_floats(const S2tZipLayerDesc* desc, const S2tZipLayerCall* call, int backward) {
  if (!desc || !call) return -1;
  State tmp;
  memset(&tmp, 0, sizeof(tmp));
  Ctx c{*desc, *call, tmp, Arena{nullptr, 0, 0}, nullptr, nullptr, true, (long)call->T * call->B};
  if (layer_fwd(c) != 0) return -1;
  const long fwd = c.ar.off;
  if (!backward) return fwd;
  c.ar.off = 0;
  if (layer_bwd(c, 0) != 0) return -1;
  return c.ar.off + 64;
}

// 0 = shapes timed (or not needed), 1 = some product of this call has no plan entry yet
int s2t_zip_layer_plans_missing(const S2tZipLayerDesc* desc, int T, int B) {
  if (!desc) return -1;
  S2tZipLayerCall call;
  memset(&call, 0, sizeof(call));
  call.x3p_on = 1;
  State tmp;
  Ctx c{*desc, call, tmp, Arena{nullptr, 0, 0}, nullptr, nullptr, true, (long)T * B};
  const long R = (long)T * B;
  const S2tZlLin* both[] = {&desc->attn_in,   &desc->ff[0].in, &desc->ff[0].out, &desc->ff[1].in, &desc->ff[1].out,
                            &desc->ff[2].in,  &desc->ff[2].out, &desc->na.in,    &desc->na.out,   &desc->sa[0].in,
                            &desc->sa[0].out, &desc->sa[1].in, &desc->sa[1].out, &desc->cv[0].in, &desc->cv[0].out,
                            &desc->cv[1].in,  &desc->cv[1].out};
  std::lock_guard<std::mutex> lock(g_mu);
  for (const S2tZlLin* L : both)
    if (plan_missing(c, 0, R, *L) || plan_missing(c, 1, R, *L)) return 1;
  if (plan_missing(c, 0, 2 * T - 1, desc->attn_pos)) return 1;
  return 0;
}

int s2t_zip_layer_fwd(const S2tZipLayerDesc* desc, const S2tZipLayerCall* call, void* state, float* ws,
                      long ws_floats, void* stream, void* side) {
  if (!desc || !call || !state || !ws || call->T < 4 || call->B < 1 || !call->x0 || !call->out) return -1;
  g_err[0] = 0;
  State& s = *reinterpret_cast<State*>(state);
  memset(&s, 0, sizeof(State));
  Ctx c{*desc, *call, s, Arena{ws, ws_floats, 0}, (hipStream_t)stream, (hipStream_t)side, false,
        (long)call->T * call->B};
  {
    // sized by the caller from the dry run; checked here before anything is launched
    State tmp;
    memset(&tmp, 0, sizeof(tmp));
    Ctx dr{*desc, *call, tmp, Arena{nullptr, 0, 0}, nullptr, nullptr, true, c.R};
    if (layer_fwd(dr) != 0 || dr.ar.off > ws_floats) return fail(-4, "s2t_zip_layer_fwd: workspace too small");
  }
  if (s2t_zip_layer_plans_missing(desc, call->T, call->B) == 1 && call->x3p_on)
    return fail(-5, "s2t_zip_layer_fwd: a product of this shape has not been timed yet");
  return layer_fwd(c);
}

// returns 0, or 1 when the score penalty of this call is active (phase 0 then stopped before the
// attention-weights backward: the caller writes dqkp / dpos and calls again with phase 2)
int s2t_zip_layer_bwd(const S2tZipLayerDesc* desc, const S2tZipLayerCall* call, void* state, float* ws,
                      long ws_floats, int phase, void* stream, void* side) {
  if (!desc || !call || !state || !ws || !call->g || !call->gx || phase < 0 || phase > 2) return -1;
  g_err[0] = 0;
  State& s = *reinterpret_cast<State*>(state);
  if (s.magic != kMagic || s.T != call->T || s.B != call->B) return fail(-1, "s2t_zip_layer_bwd: stale state");
  Ctx c{*desc, *call, s, Arena{ws, ws_floats, phase == 2 ? s.bwd_off : 0}, (hipStream_t)stream,
        (hipStream_t)side, false, (long)call->T * call->B};
  if (phase != 2) {
    State tmp = s;
    Ctx dr{*desc, *call, tmp, Arena{nullptr, 0, 0}, nullptr, nullptr, true, c.R};
    if (layer_bwd(dr, 0) != 0 || dr.ar.off + 64 > ws_floats)
      return fail(-4, "s2t_zip_layer_bwd: workspace too small");
  }
  return layer_bwd(c, phase);
}

// what: 0 whiten site `idx` active (-1 not fired / not reached), 1 penalty active, 2.. device addresses of
// buffers the penalised-score fallback needs
long s2t_zip_layer_info(const void* state, int what, int idx) {
  const State& s = *reinterpret_cast<const State*>(state);
  if (s.magic != kMagic) return -1;
  switch (what) {
    case 0: return (idx >= 0 && idx < S2T_ZL_NWHITEN) ? s.wh_active[idx] : -1;
    case 1: return s.pen_active;
    case 2: return (long)(uintptr_t)s.qkp;
    case 3: return (long)(uintptr_t)s.posp;
    case 4: return (long)(uintptr_t)s.W;
    case 5: return (long)(uintptr_t)s.dO[idx & 1];
    case 6: return (long)(uintptr_t)s.sa[idx == 0 ? 1 : 0].v;      // pairs are in backward order: [self_attn2, self_attn1]
    case 7: return (long)(uintptr_t)s.dW0;
    case 8: return (long)(uintptr_t)s.dqkp;
    case 9: return (long)(uintptr_t)s.dpos;
    case 10: return s.sa[idx == 0 ? 1 : 0].dv;
    default: return -1;
  }
}

}  // extern "C"
