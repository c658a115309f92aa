// Per-device dynamic-LDS limit of a kernel (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "hip_guard.h"

namespace ocr {

// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device: a worker pool drives worker i on
// GPU i mod n from one process (host/paddle_ocr_hip.h), so a process-wide "done once" flag leaves every device
// but the first without the attribute and its launches fail.  The memo is the call site's (one per kernel
// instantiation): per device the byte count the attribute was last raised to (0 = not tried, -1 = refused: the
// caller takes its small-LDS path).  One instantiation can serve launches with different LDS sizes (the fused
// depthwise->pointwise kernel's parameter block grows with the column-tile count), so a request above the memoised
// value raises the attribute again instead of being answered from the memo.  Several host threads (detector lanes,
// pool workers) come through here concurrently: the memo entries are atomics, and raising twice is harmless.
struct LdsAttrMemo {
  std::atomic<int> raised[64];
};

// test hook (tests/test_host_layer.py through ocr_selftest_lds_memo): answers for hipGetDevice / hipFuncSetAttribute.
// Passed by the self-test as an argument - there is no global hook a running pipeline's launcher could pick up.
struct LdsAttrHooks {
  int (*get_device)() = nullptr;                       // < 0: failure
  bool (*set_attribute)(const void*, int) = nullptr;
};

inline bool raise_dynamic_lds(const void* kernel, int bytes, LdsAttrMemo& memo, const LdsAttrHooks* hooks = nullptr) {
  int dev = -1;
  static const LdsAttrHooks none;
  const LdsAttrHooks& hk = hooks ? *hooks : none;
  if (hk.get_device) dev = hk.get_device();
  else dev = rt_current_device();  // the calling thread's LOGICAL device (hip_guard.h): two pool workers that share a GPU keep two entries
  if (dev < 0) return false;
  const bool memoised = dev < 64;
  if (memoised) {
    const int have = memo.raised[dev].load(std::memory_order_acquire);
    if (have < 0) return false;
    if (have >= bytes) return true;
  }
  bool ok;
  if (hk.set_attribute) ok = hk.set_attribute(kernel, bytes);
  else {
    ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    if (!ok) (void)hipGetLastError();
  }
  if (memoised) {
    if (!ok) memo.raised[dev].store(-1, std::memory_order_release);
    else {
      int have = memo.raised[dev].load(std::memory_order_relaxed);
      while (have >= 0 && have < bytes && !memo.raised[dev].compare_exchange_weak(have, bytes, std::memory_order_release)) {}
    }
  }
  return ok;
}

}  // namespace ocr
