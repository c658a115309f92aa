// Development micro-benchmark (round 4): what does a VALU instruction cost when it sits BETWEEN the f32 MFMAs of the
// same wave?  Per MFMA (v_mfma_f32_32x32x2_f32, 4 independent accumulators round-robin) NV VALU instructions of one kind
// on 8 independent registers:  kind 1 = v_fma_f32, 2 = v_pk_fma_f32, 3 = v_pk_mul_f32, 4 = v_add_f32 (inline asm: the
// compiler neither reorders nor unpacks them).  WPS waves per SIMD.  Prints ms and the MFMA-only equivalent rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND, int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  floatx16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  float2v v[8], m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int j = 0; j < 8; ++j) v[j] = float2v{threadIdx.x * 1e-3f + j, 1.f * j};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float2v& x = v[(u * NV + j) & 7];
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x.x) : "v"(m.x), "v"(c.x));
        else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c));
        else if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(m));
        else if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.x) : "v"(c.x));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
  if (s == 1234.5f) out[threadIdx.x] = s;
}

template <int KIND, int NV>
void run(float* out, int wgs_per_cu, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  const int grid = 256 * wgs_per_cu;
  k<KIND, NV><<<grid, 256>>>(out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<KIND, NV><<<grid, 256>>>(out, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mf = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) * 1e-12;
  // cycles per (MFMA + its NV VALU) per SIMD at 2.4 GHz: ms * 2.4e6 / (iters * 16 * waves per SIMD)
  printf("%-14s x%d  waves/SIMD %d: %8.3f ms  %6.1f TFLOP/s (MFMA only)  %6.1f clk per MFMA+VALU group per wave\n", name, NV, wgs_per_cu, ms, mf,
         ms * 2.4e6 / (iters * 16.0 * wgs_per_cu));
}

int main() {
  float* out; CK(hipMalloc(&out, 4096));
  for (int w : {1, 2}) {
    run<0, 0>(out, w, "mfma only");
    run<1, 1>(out, w, "v_fma_f32"); run<1, 2>(out, w, "v_fma_f32"); run<1, 4>(out, w, "v_fma_f32"); run<1, 8>(out, w, "v_fma_f32"); run<1, 16>(out, w, "v_fma_f32");
    run<2, 1>(out, w, "v_pk_fma_f32"); run<2, 2>(out, w, "v_pk_fma_f32"); run<2, 4>(out, w, "v_pk_fma_f32"); run<2, 8>(out, w, "v_pk_fma_f32");
    run<3, 2>(out, w, "v_pk_mul_f32"); run<3, 4>(out, w, "v_pk_mul_f32");
    run<4, 4>(out, w, "v_add_f32"); run<4, 8>(out, w, "v_add_f32");
  }
  return 0;
}
