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
