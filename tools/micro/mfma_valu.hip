// Development micro-benchmark: do VALU instructions of one wave slow the MFMA stream of another wave
// on the same SIMD?  8 waves per workgroup (2 per SIMD), 1 workgroup per CU: waves 0-3 run an MFMA
// chain, waves 4-7 a VALU fma chain (mode bit 0 = MFMA waves active, bit 1 = VALU waves active).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VKIND>
__global__ void __launch_bounds__(512, 1) k(float* out, int iters, int mode, int valu_iters) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (!(mode & 1)) return;
    floatx16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 1234.5f) out[threadIdx.x] = s;
  } else {
    if (!(mode & 2)) return;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 1e-3f + j;
    for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (VKIND == 0) v[j] = __builtin_fmaf(v[j], 1.0001f, 0.5f);
          else v[j] = __builtin_fmaxf(v[j] * 1.0001f, 0.25f);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += v[j];
    if (s == 1234.5f) out[threadIdx.x] = s;
  }
}

int main() {
  float* out; CK(hipMalloc(&out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;  // 16 MFMA per iter per wave
  for (int vi : {0, 1000, 2000, 4000, 8000}) {
    for (int mode : {1, 2, 3}) {
      if (vi == 0 && mode != 1) continue;
      if (vi != 0 && mode == 1) continue;
      k<0><<<256, 512>>>(out, iters, mode, vi);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      k<0><<<256, 512>>>(out, iters, mode, vi);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double mf = 256.0 * 4 * iters * 16 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) * 1e-12;
      printf("valu_iters %5d (=%d VALU instr/wave) mode %d: %.3f ms   mfma-equivalent %.1f TFLOP/s\n", vi, vi * 64, mode, ms, (mode & 1) ? mf : 0.0);
    }
  }
  return 0;
}
