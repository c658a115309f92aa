// Development microbenchmark: sustained rate of v_mfma_f32_32x32x2_f32 with NO memory traffic,
// at 1..4 accumulator chains per wave and 1..4 waves per SIMD.  Calibrates what "MFMA peak" means
// on the box's actual clocks.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int NT>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  floatx16 acc[NT];
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < NT; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT>
void run(int wgs_per_cu, int iters) {
  float* d;
  const int blocks = 256 * wgs_per_cu;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<NT><<<blocks, 256>>>(d, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NT><<<blocks, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * 4 * iters * 4 * NT * 4096.0;
  printf("NT=%d waves/SIMD=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NT, wgs_per_cu, iters, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 4; ++w) { run<1>(w, 20000); run<2>(w, 10000); run<4>(w, 5000); }
  run<4>(4, 50000);
  return 0;
}
