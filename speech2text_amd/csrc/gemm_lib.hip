// Thin host-side caller of hipBLASLt for the PLAIN dense GEMMs of the training path (forward
// y = x W^T and data gradient dx = g W of every Linear).  "hipBLASLt only for plain library
// GEMMs": no kernel here, just the C API driven directly (one heuristic query per distinct
// shape, then cached) so that
//   * bias AND residual ride in the GEMM epilogue (D = A B + bias + beta C): the reference's
//     `src = src + linear(h)` (model/encoder/zipformer.py:1095-1221) is one launch, and a data
//     gradient can be accumulated onto an existing gradient (beta = 1);
//   * the host pays a ~µs table lookup per call instead of a framework dispatch.
// Shapes change almost every step under the reference's duration-bucketed batching
// (dataset/sampler.py:71-96: M = T*B).  A plan (descriptors + the heuristic's candidates) is made
// per exact shape (cheap), but candidates are TIMED only once per bucket {mode, half-octave
// of M, N, K, bias}: later shapes of a bucket take the bucket's winning kernel if the heuristic
// offers it for them too.  The number of timed buckets per process and the plan table are capped.
// hipBLASLt is column-major; a row-major (rows, cols, ld) matrix is passed as the column-major
// (cols, rows, ld) one, i.e. the row-major product Y = X W^T is computed as Y^T = W X^T.
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

extern "C" int s2t_gemm_f32(int mode, const float* A, long lda, const float* B, long ldb, float* C,
                            long ldc, int M, int N, int K, const float* bias, const float* resid,
                            long ldr, const float* act_src, long lds, int act_kind, int pro_a,
                            int pro_b, float* colsum, int accumulate, void* stream);
extern "C" int s2t_nn_x3(int set);
extern "C" int s2t_gemm_f32_tiled(int mode, const float* A, long lda, const float* B, long ldb,
                                  float* C, long ldc, int M, int N, int K, const float* bias,
                                  const float* resid, long ldr, int tile, void* stream);

namespace {

constexpr int MAX_CAND = 16;

// Our own NT / NN MFMA kernel (gemm.hip, bf16x3 form) as one more candidate of the plan: same
// product, bias and residual in its epilogue.  Usable when the residual weight is 0 or 1.
bool own_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("S2T_LT_OWN"); v = e ? atoi(e) : 1; }
  return v == 1;
}
float own_margin() {
  static float v = -1.f;
  if (v < 0.f) { const char* e = getenv("S2T_LT_OWN_MARGIN"); v = e ? (float)atof(e) : 0.97f; }
  return v;
}
int run_own(int mode, const float* X, long ldx, const float* W, long ldw, const float* bias,
            const float* C, long ldc, float beta, float* D, long ldd, int M, int N, int K, int tile,
            hipStream_t st) {
  const float* resid = (beta == 1.f && C && C != D) ? C : nullptr;
  if (beta != 0.f && !resid) return -2;
  if (mode == 0)
    return s2t_gemm_f32_tiled(0, X, ldx, W, ldw, D, ldd, M, N, K, bias, resid, ldc, tile, st);
  return s2t_gemm_f32_tiled(1, X, ldx, W, ldw, D, ldd, M, K, N, nullptr, resid, ldc, tile, st);
}

struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
  // the heuristic's candidates, timed on the real operands at the first call (tune())
  hipblasLtMatmulHeuristicResult_t cand[MAX_CAND];
  int ncand = 0;
  bool tuned = false;
  bool own = false;      // the timed choice is OUR MFMA kernel (s2t_gemm_f32, bf16x3 form), not a library one
  int own_tile = 0;      // ... with this block tile
};

using Key = std::tuple<int, int, int, int, long, long, long, long, int>;
std::map<Key, Plan> g_plans;
using BKey = std::tuple<int, int, int, int, int>;          // mode, half-octave of M, N, K, bias
struct Winner {
  hipblasLtMatmulAlgo_t algo;
  bool own;
  int own_tile;
};
std::map<BKey, Winner> g_winner;                           // the timed choice of a bucket
int g_tunings = 0;
long g_own_calls = 0;
constexpr size_t MAX_PLANS = 8192;

int tune_budget() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("S2T_LT_TUNE_MAX");
    v = e ? atoi(e) : 192;
  }
  return v;
}

int half_octave(int m) { return (int)std::floor(2.0 * std::log2((double)(m < 1 ? 1 : m))); }

void destroy_plan(Plan& p) {
  if (p.a) hipblasLtMatrixLayoutDestroy(p.a);
  if (p.b) hipblasLtMatrixLayoutDestroy(p.b);
  if (p.c) hipblasLtMatrixLayoutDestroy(p.c);
  if (p.d) hipblasLtMatrixLayoutDestroy(p.d);
  if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
}
std::mutex g_mu;
hipblasLtHandle_t g_handle = nullptr;
long g_lt_calls = 0;

#define LT_CHECK(x)                                   \
  do {                                                \
    hipblasStatus_t s__ = (x);                        \
    if (s__ != HIPBLAS_STATUS_SUCCESS) return -100 - (int)s__; \
  } while (0)

int make_plan(Plan& p, int mode, int M, int N, int K, long ldx, long ldw, long ldc, long ldd,
              bool has_bias, size_t ws_bytes) {
  if (!g_handle) LT_CHECK(hipblasLtCreate(&g_handle));
  LT_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  // mode 0: Yc (N x M) = op_T(Wc: K x N, ld ldw) . Xc (K x M, ld ldx)
  // mode 1: dXc (K x M) = Wc (K x N, ld ldw) . Gc (N x M, ld ldx)      (x := g, output cols := K)
  hipblasOperation_t opa = mode == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = HIPBLAS_OP_N;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
  if (has_bias) {
    hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
  }
  const int out_rows = mode == 0 ? N : K;     // column-major rows of the result = output features
  const int inner = mode == 0 ? K : N;
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.a, HIP_R_32F, K, N, ldw));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.b, HIP_R_32F, inner, M, ldx));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.c, HIP_R_32F, out_rows, M, ldc));
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.d, HIP_R_32F, out_rows, M, ldd));
  hipblasLtMatmulPreference_t pref;
  LT_CHECK(hipblasLtMatmulPreferenceCreate(&pref));
  LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES,
                                                 &ws_bytes, sizeof(ws_bytes)));
  int n = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, p.c, p.d, pref,
                                                       MAX_CAND, p.cand, &n);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || n < 1) return -2;      // caller falls back
  p.ncand = n;
  p.algo = p.cand[0].algo;
  p.ws = p.cand[0].workspaceSize;
  p.ok = true;
  return 0;
}

bool tuning_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("S2T_LT_TUNE");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// Time every candidate the heuristic returned on the call's own operands (output into a scratch
// buffer, so accumulating epilogues are not applied twice) and keep the fastest.  The library's
// first choice is tuned for large square problems; on the tall, short-K shapes of this model
// another of its kernels is often 10-30 % faster.  Runs once per distinct shape (during warm-up).
void tune(Plan& p, const float* X, const float* W, const float* C, float beta, long d_elems,
          void* workspace, size_t ws_bytes, hipStream_t st, int mode, long ldx, long ldw,
          const float* bias, long ldc, long ldd, int M, int N, int K) {
  p.tuned = true;
  if (p.ncand < 2 || !tuning_enabled()) return;
  // The candidates must be compared on an otherwise idle chip: the step's side streams (the
  // grouped weight-gradient GEMM, the Whiten statistics) run 400 us kernels beside the main
  // stream, and whichever candidate was timed under one of them lost -- at the conformer's
  // 7936 x 2048 -> 256 product that left a 31-tile kernel in the plan (239 us against 67 us).
  (void)hipDeviceSynchronize();
  float* scratch = nullptr;
  if (hipMalloc(&scratch, (size_t)d_elems * sizeof(float)) != hipSuccess) return;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const float alpha = 1.f;
  if (!C) C = scratch;                     // beta == 0: the C operand is only a placeholder
  // 4 back-to-back runs of candidate i, timed after one untimed run; -1 if the library rejects it
  auto time_cand = [&](int i) -> float {
    if (p.cand[i].workspaceSize > ws_bytes) return -1.f;
    for (int rep = 0; rep < 5; ++rep) {
      if (rep == 1) (void)hipEventRecord(e0, st);
      if (hipblasLtMatmul(g_handle, p.desc, &alpha, W, p.a, X, p.b, &beta, C, p.c, scratch, p.d,
                          &p.cand[i].algo, workspace, p.cand[i].workspaceSize,
                          st) != HIPBLAS_STATUS_SUCCESS)
        return -1.f;
    }
    float ms = 0.f;
    (void)hipEventRecord(e1, st);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
  };
  (void)time_cand(0);                      // clocks and caches up before anything is compared
  float best = 1e30f;
  int best_i = 0;
  static int max_try = -1;
  if (max_try < 0) { const char* e = getenv("S2T_LT_CANDIDATES"); max_try = e ? atoi(e) : 8; }
  const int ntry = p.ncand < max_try ? p.ncand : max_try;
  for (int i = 0; i < ntry; ++i) {
    // the better of two measurements: one pass picks a different kernel from run to run often
    // enough to move the step by a millisecond
    float ms = time_cand(i);
    if (ms > 0.f) {
      const float m2 = time_cand(i);
      if (m2 > 0.f && m2 < ms) ms = m2;
    }
    if (ms > 0.f && ms < best) {
      best = ms;
      best_i = i;
    }
  }
  // Back-to-back timing on warm caches is only a proxy for the kernel's speed inside the step, and
  // a single pass can be fooled by a clock ramp (the first shapes are tuned on a cold chip): the
  // heuristic's own choice stays unless the challenger beats it by > 8 % in two further
  // head-to-head rounds.
  if (best_i != 0) {
    for (int round = 0; round < 2 && best_i != 0; ++round) {
      const float t0 = time_cand(0), tb = time_cand(best_i);
      if (!(t0 > 0.f && tb > 0.f && tb < 0.92f * t0)) best_i = 0;
    }
  }
  p.algo = p.cand[best_i].algo;
  p.ws = p.cand[best_i].workspaceSize;
  // our kernel against the library's best: same operands, same epilogue, output into the scratch
  if (own_enabled() && (beta == 0.f || beta == 1.f)) {
    auto time_own = [&](int tile) -> float {
      for (int rep = 0; rep < 5; ++rep) {
        if (rep == 1) (void)hipEventRecord(e0, st);
        if (run_own(mode, X, ldx, W, ldw, bias, beta == 1.f ? C : nullptr, ldc, beta, scratch, ldd, M,
                    N, K, tile, st) != 0)
          return -1.f;
      }
      float ms = 0.f;
      (void)hipEventRecord(e1, st);
      (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      return ms;
    };
    // our kernel's block tiles: small tiles have the lower fixed cost (more rounds of workgroups,
    // their store phases overlap other workgroups' main loops), large ones the better rate per k
    static const int tiles[] = {21, 12, 22, 23, 11};
    float best_o = 1e30f;
    int best_t = 0;
    for (int t : tiles) {
      const float ms = time_own(t);
      if (ms > 0.f && ms < best_o) {
        best_o = ms;
        best_t = t;
      }
    }
    bool win = best_t != 0;
    for (int round = 0; round < 2 && win; ++round) {
      const float tl = time_cand(best_i), to = time_own(best_t);
      win = to > 0.f && tl > 0.f && to < own_margin() * tl;
    }
    p.own = win;
    p.own_tile = win ? best_t : 0;
    static int dump = -1;      // S2T_LT_DUMP=1: one line per timed bucket (diagnostics)
    if (dump < 0) { const char* e = getenv("S2T_LT_DUMP"); dump = e ? atoi(e) : 0; }
    if (dump)
      fprintf(stderr, "[s2t lt] mode %d M %d N %d K %d bias %d beta %g: library %.1f us, ours %.1f us (tile %d)%s\n",
              mode, M, N, K, bias ? 1 : 0, (double)beta, 1e3 * time_cand(best_i) / 4.0, 1e3 * best_o / 4.0,
              best_t, win ? " <- ours" : "");
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(scratch);
}

}  // namespace

// (plans made, buckets timed) so far: tests / diagnostics
extern "C" int s2t_linear_lt_stats(int* plans, int* timed) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (plans) *plans = (int)g_plans.size();
  if (timed) *timed = g_tunings;
  return 0;
}

// mode 0: D[M,N] = X[M,K] W[N,K]^T (+ bias[N]) (+ beta C[M,N]);   mode 1: D[M,K] = X[M,N] W[N,K] (+ beta C[M,K]).
// Row-major, leading dimensions in floats.  C may be NULL when beta == 0, and may alias D.
// Returns 0, -2 when hipBLASLt has no algorithm for the shape (caller falls back), or < -100.
extern "C" int s2t_linear_lt(int mode, const float* X, long ldx, const float* W, long ldw,
                             const float* bias, const float* C, long ldc, float beta, float* D,
                             long ldd, int M, int N, int K, void* workspace, long ws_bytes,
                             void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (mode != 0 && mode != 1)) return -1;
  if (mode == 1 && bias) return -1;
  std::lock_guard<std::mutex> lock(g_mu);
  if (!C) {
    C = D;
    ldc = ldd;
    beta = 0.f;
  }
  const Key key{mode, M, N, K, ldx, ldw, ldc, ldd, bias ? 1 : 0};
  auto it = g_plans.find(key);
  if (it == g_plans.end()) {
    if (g_plans.size() >= MAX_PLANS) {       // bounded: start over (bucket winners are kept)
      for (auto& kv : g_plans) destroy_plan(kv.second);
      g_plans.clear();
    }
    Plan p;
    const int rc = make_plan(p, mode, M, N, K, ldx, ldw, ldc, ldd, bias != nullptr, (size_t)ws_bytes);
    it = g_plans.emplace(key, p).first;
    if (rc != 0 && rc != -2) return rc;
  }
  Plan& p = it->second;
  if (!p.ok || p.ws > (size_t)ws_bytes) return -2;
  ++g_lt_calls;
  if (bias)
    LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
  if (!p.tuned) {
    const BKey bkey{mode, half_octave(M), N, K, bias ? 1 : 0};
    auto w = g_winner.find(bkey);
    if (w != g_winner.end()) {
      // the bucket was timed on another M: take its kernel if the heuristic offers it here too
      p.tuned = true;
      p.own = w->second.own;
      p.own_tile = w->second.own_tile;
      for (int i = 0; i < p.ncand; ++i)
        if (p.cand[i].workspaceSize <= (size_t)ws_bytes &&
            memcmp(&p.cand[i].algo, &w->second.algo, sizeof(hipblasLtMatmulAlgo_t)) == 0) {
          p.algo = p.cand[i].algo;
          p.ws = p.cand[i].workspaceSize;
          break;
        }
    } else if (g_tunings < tune_budget()) {
      tune(p, X, W, C == D ? nullptr : C, C == D ? 0.f : beta, (long)M * ldd, workspace,
           (size_t)ws_bytes, (hipStream_t)stream, mode, ldx, ldw, bias, ldc, ldd, M, N, K);
      g_winner.emplace(bkey, Winner{p.algo, p.own, p.own_tile});
      ++g_tunings;
    } else {
      p.tuned = true;                      // budget spent: the heuristic's first choice
    }
  }
  if (p.own && (beta == 0.f || (beta == 1.f && C != D)) &&
      run_own(mode, X, ldx, W, ldw, bias, C, ldc, beta, D, ldd, M, N, K, p.own_tile,
              (hipStream_t)stream) == 0) {
    ++g_own_calls;
    return 0;
  }
  const float alpha = 1.f;
  LT_CHECK(hipblasLtMatmul(g_handle, p.desc, &alpha, W, p.a, X, p.b, &beta, C, p.c, D, p.d, &p.algo,
                           workspace, p.ws, (hipStream_t)stream));
  return 0;
}

extern "C" long s2t_linear_lt_calls(void) { return g_lt_calls; }
// launches served by our own kernel so far (diagnostics / tests)
extern "C" long s2t_linear_lt_own_calls(void) { return g_own_calls; }

// ---- batch of independent row-major fp32 products through hipBLASLt (strided batch): the shapes the
// nonlinear attention's products (model/encoder/zipformer.py:2468-2473 and their gradients) hand to
// the library -- rows that are not 16-byte multiples (T = 495, 62) and the a^T . b form:
//   mode 0: C_b[M,N] = A_b[M,K] . B_b[N,K]^T;  mode 1: C_b = A_b[M,K] . B_b[K,N];
//   mode 2: C_b[M,N] = A_b[K,M]^T . B_b[K,N].   Dense operands, batch stride = rows * cols.
// Column-major view: C^T (N x M) = op(Bc) . op(Ac).
namespace {
struct BPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t ws = 0;
  bool ok = false;
};
using BmmKey = std::tuple<int, int, int, int, int>;
std::map<BmmKey, BPlan> g_bplans;

int set_batch(hipblasLtMatrixLayout_t l, int batch, long stride) {
  int32_t bc = batch;
  int64_t so = stride;
  LT_CHECK(hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bc, sizeof(bc)));
  LT_CHECK(hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &so, sizeof(so)));
  return 0;
}

int make_bplan(BPlan& p, int mode, int M, int N, int K, int batch, size_t ws_bytes) {
  if (!g_handle) LT_CHECK(hipblasLtCreate(&g_handle));
  LT_CHECK(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  // lt "A" = our B as a column-major matrix, lt "B" = our A
  hipblasOperation_t opa = mode == 0 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  hipblasOperation_t opb = mode == 2 ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa)));
  LT_CHECK(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb)));
  if (mode == 0) LT_CHECK(hipblasLtMatrixLayoutCreate(&p.a, HIP_R_32F, K, N, K));     // B[N,K] rows
  else LT_CHECK(hipblasLtMatrixLayoutCreate(&p.a, HIP_R_32F, N, K, N));               // B[K,N] rows
  if (mode == 2) LT_CHECK(hipblasLtMatrixLayoutCreate(&p.b, HIP_R_32F, M, K, M));     // A[K,M] rows
  else LT_CHECK(hipblasLtMatrixLayoutCreate(&p.b, HIP_R_32F, K, M, K));               // A[M,K] rows
  LT_CHECK(hipblasLtMatrixLayoutCreate(&p.d, HIP_R_32F, N, M, N));
  int rc = set_batch(p.a, batch, (long)N * K);
  if (rc) return rc;
  rc = set_batch(p.b, batch, (long)M * K);
  if (rc) return rc;
  rc = set_batch(p.d, batch, (long)M * N);
  if (rc) return rc;
  hipblasLtMatmulPreference_t pref;
  LT_CHECK(hipblasLtMatmulPreferenceCreate(&pref));
  LT_CHECK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES,
                                                 &ws_bytes, sizeof(ws_bytes)));
  hipblasLtMatmulHeuristicResult_t res[4];
  int n = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(g_handle, p.desc, p.a, p.b, p.d, p.d, pref, 4, res, &n);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || n < 1) return -2;
  p.algo = res[0].algo;
  p.ws = res[0].workspaceSize;
  p.ok = true;
  return 0;
}
}  // namespace

extern "C" int s2t_bmm_lt(int mode, const float* A, const float* B, float* C, int batch, int M, int N,
                          int K, void* workspace, long ws_bytes, void* stream) {
  if (batch <= 0 || M <= 0 || N <= 0 || K <= 0 || mode < 0 || mode > 2 || !A || !B || !C) return -1;
  std::lock_guard<std::mutex> lock(g_mu);
  const BmmKey key{mode, M, N, K, batch};
  auto it = g_bplans.find(key);
  if (it == g_bplans.end()) {
    if (g_bplans.size() >= 512) return -2;
    BPlan p;
    const int rc = make_bplan(p, mode, M, N, K, batch, (size_t)ws_bytes);
    it = g_bplans.emplace(key, p).first;
    if (rc != 0 && rc != -2) return rc;
  }
  BPlan& p = it->second;
  if (!p.ok || p.ws > (size_t)ws_bytes) return -2;
  const float alpha = 1.f, beta = 0.f;
  LT_CHECK(hipblasLtMatmul(g_handle, p.desc, &alpha, B, p.a, A, p.b, &beta, C, p.d, C, p.d, &p.algo,
                           workspace, p.ws, (hipStream_t)stream));
  return 0;
}
