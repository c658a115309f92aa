// Issue rate of v_mfma_f32_4x4x1_16B_f32 against v_mfma_f32_32x32x2_f32 (same FLOP/clk on paper): one wave per SIMD
// and four, 6 independent accumulators.   hipcc --offload-arch=gfx950 -O3 -o mfma_4x4 mfma_4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k4(float* out, int iters) {
  floatx4 acc[NACC];
  for (int g = 0; g < NACC; ++g) acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int g = 0; g < NACC; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[g], 0, 0, 0);
  }
  float s = 0;
  for (int g = 0; g < NACC; ++g) s += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k32(float* out, int iters) {
  floatx16 acc[2];
  for (int g = 0; g < 2; ++g) for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
  }
  float s = 0;
  for (int g = 0; g < 2; ++g) for (int r = 0; r < 16; ++r) s += acc[g][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 26);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int waves = 1; waves <= 4; waves *= 2) {
    const int blocks = 256 * 4, threads = 64 * waves, iters = 4000;
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k4, dim3(blocks), dim3(threads), 0, 0, d, iters);
        else hipLaunchKernelGGL(k32, dim3(blocks), dim3(threads), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      const double fl = which == 0 ? 8.0 * NACC * 512 : 6.0 * 4096;
      printf("%s waves/block %d: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "4x4x1  " : "32x32x2", waves, ms, (double)blocks * waves * iters * fl / ms / 1e9);
    }
  }
  return 0;
}
