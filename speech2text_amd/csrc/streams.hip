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
