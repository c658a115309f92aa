// Per-device dynamic-LDS limit of a kernel (host side).
#pragma once
#include <hip/hip_runtime.h>

namespace ocr {

// More than 64 KB of dynamic LDS has to be allowed per kernel AND per device: a worker pool drives worker i on
// GPU i mod n from one process (host/paddle_ocr_hip.h), so a process-wide "done once" flag leaves every device
// but the first without the attribute and its launches fail.  `state` is the call site's per-device memo
// (0 = not tried, 1 = raised, 2 = refused: the caller takes its small-LDS path).
inline bool raise_dynamic_lds(const void* kernel, int bytes, unsigned char (&state)[64]) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  const bool memo = dev >= 0 && dev < 64;
  if (memo && state[dev]) return state[dev] == 1;
  const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  if (memo) state[dev] = ok ? 1 : 2;
  return ok;
}

}  // namespace ocr
