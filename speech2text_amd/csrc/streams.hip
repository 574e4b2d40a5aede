// Fork / join of a side HIP stream for work that is off the critical path of backward: the
// weight-gradient GEMMs (dW = g^T x) only feed the optimizer, while the data gradients feed the
// next layer's backward.  Running them on a second stream lets them fill the CUs the small,
// latency-bound kernels of the main chain leave idle.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <vector>

namespace {
std::mutex g_mu;
std::vector<hipEvent_t> g_pool;
size_t g_next = 0;
hipStream_t g_side = nullptr;

hipEvent_t next_event() {
  if (g_pool.size() < 512) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    g_pool.push_back(e);
    return e;
  }
  hipEvent_t e = g_pool[g_next % g_pool.size()];   // ring: an event this old has long completed
  ++g_next;
  return e;
}
}  // namespace

// The library-owned side stream of the current device (created on first use).
extern "C" void* s2t_side_stream(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_side) {
    // lowest priority: the side stream's work only feeds the optimizer, the main stream is the
    // step's critical path (S2T_SIDE_PRIORITY=0: default priority)
    int lo = 0, hi = 0;
    const char* e = getenv("S2T_SIDE_PRIORITY");
    const bool low = !(e && e[0] == '0');
    if (low && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
      if (hipStreamCreateWithPriority(&g_side, hipStreamNonBlocking, lo) != hipSuccess) g_side = nullptr;
    }
    if (!g_side && hipStreamCreateWithFlags(&g_side, hipStreamNonBlocking) != hipSuccess) return nullptr;
  }
  return (void*)g_side;
}

// Work enqueued on `to` after this call starts only after everything enqueued so far on `from`.
extern "C" int s2t_stream_order(void* from, void* to) {
  std::lock_guard<std::mutex> lock(g_mu);
  hipEvent_t e = next_event();
  if (!e) return -1;
  hipError_t rc = hipEventRecord(e, (hipStream_t)from);
  if (rc != hipSuccess) return (int)rc;
  rc = hipStreamWaitEvent((hipStream_t)to, e, 0);
  return (int)rc;
}

// ---- kernel-attached timing.  A (start, stop) pair armed here is handed to the NEXT instrumented
// launch of this thread (csrc/gemm_x3p.hip: hipExtLaunchKernelGGL), which stamps the pair with
// the kernel's own begin / end times -- the same interval rocprof reports.  Events recorded AROUND
// a launch (marker packets before and after) add their dispatch latency to a ~40 us kernel:
// bench.py's in-step figure for s2t_gemm_x3p read 37-44 us where rocprof read 34-37.
thread_local hipEvent_t s2t_prof_start = nullptr, s2t_prof_stop = nullptr;

extern "C" int s2t_prof_pair_create(void** start, void** stop) {
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return -1;
  if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1; }
  *start = (void*)a;
  *stop = (void*)b;
  return 0;
}
extern "C" int s2t_prof_pair_arm(void* start, void* stop) {
  s2t_prof_start = (hipEvent_t)start;
  s2t_prof_stop = (hipEvent_t)stop;
  return 0;
}
// 1 = the armed pair was consumed by a launch (and is now disarmed), 0 = still armed (disarms it)
extern "C" int s2t_prof_pair_consumed(void) {
  const int used = s2t_prof_start == nullptr;
  s2t_prof_start = s2t_prof_stop = nullptr;
  return used;
}
extern "C" int s2t_prof_pair_ms(void* start, void* stop, float* ms) {
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1;
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? 0 : -1;
}
extern "C" int s2t_prof_pair_destroy(void* start, void* stop) {
  (void)hipEventDestroy((hipEvent_t)start);
  (void)hipEventDestroy((hipEvent_t)stop);
  return 0;
}

