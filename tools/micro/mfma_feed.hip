// Development microbenchmark: MFMA rate when the operands of every step come from memory.
//   mode 0: operands constant (mfma_peak)          mode 1: 6 x ds_read_b128 per 32 MFMAs (no barrier)
//   mode 2: mode 1 + one __syncthreads per step    mode 3: 1 + 4 global 16-byte loads (L2-resident) per 16 MFMAs, ping-pong
// hipcc --offload-arch=gfx950 -O3 -o mfma_feed mfma_feed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const float4* __restrict__ g, float* out, int steps) {
  __shared__ float4 s[2][12 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 12 * 64; i += 256) (&s[0][0])[i] = g[i];
  __syncthreads();
  floatx16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float4 xf[2], wf[4];
  for (int i = 0; i < 2; ++i) xf[i] = s[0][i * 64 + lane];
  for (int j = 0; j < 4; ++j) wf[j] = s[0][(4 + j) * 64 + lane];
  const float4* gp = g + (blockIdx.x % 64) * 4096 + wave * 1024 + lane;
  for (int st = 0; st < steps; ++st) {
    if (MODE == 1 || MODE == 2) {
      const int cur = st & 1;
      for (int i = 0; i < 2; ++i) xf[i] = s[cur][((wave >> 1) * 2 + i) * 64 + lane];
      for (int j = 0; j < 4; ++j) wf[j] = s[cur][(4 + (wave & 1) * 4 + j) * 64 + lane];
    }
    if (MODE == 3) {
      xf[0] = gp[(st & 63) * 320]; xf[1] = xf[0];
      for (int j = 0; j < 4; ++j) wf[j] = gp[(st & 63) * 320 + 64 * (j + 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < (MODE == 3 ? 1 : 2); ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].x, xf[i].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].y, xf[i].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].z, xf[i].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].w, xf[i].w, acc[i][j], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 2) __syncthreads();
  }
  float r = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 16; ++q) r += acc[i][j][q];
  out[blockIdx.x * 256 + tid] = r;
}
template <int MODE>
void run(const float4* g, float* o, const char* name) {
  const int blocks = 512, steps = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(g, o, steps); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(g, o, steps); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)blocks * 4 * steps * (MODE == 3 ? 16 : 32) * 4096.0;
  printf("%-60s %.3f ms %.1f TFLOP/s\n", name, ms, mf / ms / 1e9);
}
int main() {
  float4* g; float* o;
  hipMalloc(&g, 64 * 4096 * 16 + (1 << 20)); hipMemset(g, 0, 64 * 4096 * 16 + (1 << 20)); hipMalloc(&o, 512 * 256 * 4);
  run<0>(g, o, "constant operands, 2 waves/SIMD");
  run<1>(g, o, "6 ds_read_b128 per 32 MFMAs");
  run<2>(g, o, "6 ds_read_b128 per 32 MFMAs + barrier per step");
  run<3>(g, o, "5 global 16-byte loads (L2) per 16 MFMAs, no prefetch");
  return 0;
}
