// Development microbenchmark: what the f32 matrix pipe sustains when its operands are DATA.  mfma_peak.hip multiplies
// constants (x = lane, y = 2) for a few milliseconds; a network multiplies activations for tens of milliseconds, and the
// chip's power management answers to the toggling: this loop runs v_mfma_f32_32x32x2_f32 back to back on operand registers
// filled with (a) zeros, (b) small integers, (c) random floats, for `ms` milliseconds, and reports TFLOP/s together with the
// average shader clock (s_memtime ticks per s_memrealtime tick, the latter at 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip ;  ./mfma_power [target ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256, 2) k(const float* __restrict__ ops, float* out, unsigned long long* clk, int iters) {
  floatx16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = ops[(i * 256 + threadIdx.x) * 2]; y[i] = ops[(i * 256 + threadIdx.x) * 2 + 1]; }
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0 && blockIdx.x == 0) { t0 = __builtin_readcyclecounter(); r0 = wall_clock64(); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[(u + t) & 7], acc[t], 0, 0, 0);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0; clk[1] = wall_clock64() - r0; }
  float s = 0;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const double target_ms = argc > 1 ? atof(argv[1]) : 30.0;
  const int blocks = 256 * 2;
  float *d, *ops; unsigned long long* clk;
  hipMalloc(&d, blocks * 256 * 4); hipMalloc(&ops, 8 * 256 * 2 * 4); hipMalloc(&clk, 16);
  const char* names[3] = {"zeros", "small integers", "random floats in [-1, 1)"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<float> h(8 * 256 * 2);
    unsigned st = 777u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = mode == 0 ? 0.f : mode == 1 ? (float)((st >> 20) & 3) : ((st >> 8) & 0xffffff) / 8388608.0f - 1.0f; }
    hipMemcpy(ops, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) {  // calibrate the iteration count to the target duration, then measure
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); k<<<blocks, 256>>>(ops, d, clk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
      const double flops = (double)blocks * 4 * iters * 32 * 4096.0;
      if (rep == 2) printf("%-26s %8.2f ms  %6.1f TFLOP/s  shader clock %.0f MHz (wave 0: %llu cycles in %llu ticks of 100 MHz)\n", names[mode], ms, flops / ms / 1e9, c[1] ? 100.0 * c[0] / c[1] : 0.0, c[0], c[1]);
      iters = (int)(iters * target_ms / ms) + 1;
    }
  }
  return 0;
}
