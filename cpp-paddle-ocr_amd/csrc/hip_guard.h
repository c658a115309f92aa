// Allocation / synchronous-copy calls that must not run while ANOTHER host thread records a hipGraph.
// A pipeline runs two chains on two host threads (pipe.hip), detector lanes on more; each network records the hipGraph
// of a binding with hipStreamBeginCapture(ThreadLocal) the second time it sees the same input (Net::run_bound).  HIP
// invalidates a capture when, meanwhile, a synchronising call touches the device from any thread: hipFree (implicit
// device synchronisation), synchronous hipMemcpy on the legacy stream (implicit dependency on every blocking stream:
// "operation failed due to a previous error during capture" in the capturing thread, an error in the other).  Seen as
// a rare first-call failure with two chains (round 3).  So: every such call goes through these wrappers (shared lock),
// a capture takes the lock exclusively - captures are microseconds of host time - and the library's streams are
// non-blocking (no implicit ordering against the legacy stream, whatever the host application does on it).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <shared_mutex>

namespace ocr {

std::shared_mutex& capture_mutex();  // capi_net.hip
int rt_current_device();                // (the calling thread's logical device: below)

template <class T>
inline hipError_t g_malloc(T** p, size_t bytes) {
  std::shared_lock<std::shared_mutex> lk(capture_mutex());
  return hipMalloc((void**)p, bytes);
}
inline hipError_t g_free(void* p) {
  std::shared_lock<std::shared_mutex> lk(capture_mutex());
  return hipFree(p);
}
inline hipError_t g_host_malloc(void** p, size_t bytes, unsigned flags) {
  std::shared_lock<std::shared_mutex> lk(capture_mutex());
  return hipHostMalloc(p, bytes, flags);
}
inline hipError_t g_host_free(void* p) {
  std::shared_lock<std::shared_mutex> lk(capture_mutex());
  return hipHostFree(p);
}
inline hipError_t g_memcpy(void* d, const void* s, size_t n, hipMemcpyKind k) {
  std::shared_lock<std::shared_mutex> lk(capture_mutex());
  return hipMemcpy(d, s, n, k);
}
inline hipError_t g_memcpy2d(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k) {
  std::shared_lock<std::shared_mutex> lk(capture_mutex());
  return hipMemcpy2D(d, dp, s, sp, w, h, k);
}
inline hipError_t g_stream_create(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }

// How a host thread waits for its stream (round 6).  hipStreamSynchronize SPINS on the completion signal: a pipeline handle
// has three host threads doing that for most of a step (two chains + the caller), eight ranks of a node 24 busy cores - more
// than a one-GPU lease of this pool grants (16).  Mode 1 ("block") records an event created with hipEventBlockingSync behind
// the stream's work and waits on it: the thread sleeps in the driver until the interrupt.  Process-wide: ocr_rt_set_wait_mode
// (include/ocr_hip.h) or OCR_WAIT_MODE=block|spin in the environment (read at the first wait); bench.py reports images/s and
// host CPU seconds per step of both (the default follows those numbers: DESIGN.md section 7).
int rt_wait_mode();             // 0 spin, 1 block (capi_net.hip)
void rt_set_wait_mode(int m);
inline hipError_t g_stream_sync(hipStream_t s) {
  if (rt_wait_mode() == 0) return hipStreamSynchronize(s);
  thread_local hipEvent_t ev[64] = {};  // per thread and logical device
  const int d = rt_current_device();
  hipEvent_t& e = ev[d >= 0 && d < 64 ? d : 0];
  if (!e) {
    const hipError_t c = hipEventCreateWithFlags(&e, hipEventBlockingSync | hipEventDisableTiming);
    if (c != hipSuccess) { e = nullptr; (void)hipGetLastError(); return hipStreamSynchronize(s); }
  }
  const hipError_t r = hipEventRecord(e, s);
  if (r != hipSuccess) return r;
  return hipEventSynchronize(e);
}

// LOGICAL devices (round 5).  Every device id of the C-ABI is an index into a table that OCR_DEVICE_MAP can spell out
// ("0,0": two logical devices on physical GPU 0; default: the identity over the visible devices).  Per-device resources of the
// library - dynamic-LDS attribute memos, occupancy memos, priority anchors; arenas and streams belong to handles - are
// keyed by the LOGICAL id, so a one-GPU lease exercises the code a two-GPU pool runs (worker i -> device i mod n,
// /root/reference/src/gpu_worker_pool.cpp:46-59).  rt_set_device makes the logical device current for the calling
// thread (hipSetDevice on its physical device); rt_current_device is that thread's logical id (-1: none yet).
hipError_t rt_set_device(int logical);  // capi_net.hip
int rt_current_device();
int rt_physical_device(int logical);    // -1: out of range
int rt_device_count();

}  // namespace ocr
