// Development micro-benchmark: what a streaming kernel gets from HBM on this chip as a function of its read : write mix.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_mix tools/micro/hbm_mix.hip && ./hbm_mix
// Every lane moves 16-byte pieces, consecutive lanes consecutive addresses, one pass over buffers far larger than the 256 MB
// of last-level cache; R pieces read and W pieces written per item (R:W = 1:0 sum, 0:1 fill, 1:1 copy, 1:4 the detector's
// 12 -> 96 expand conv, 4:1 a reducing conv).  Prints GB/s of read + written bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int R, int W>
__global__ void __launch_bounds__(256) mix_kernel(const float4* __restrict__ in, float4* __restrict__ out, long items, float4* sink) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < items; i += (long)gridDim.x * 256) {
    float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4 x = in[(long)r * items + i];
      v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
    }
    if (W == 0) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
    for (int w = 0; w < W; ++w) out[(long)w * items + i] = v;
  }
  if (W == 0 && acc.x == 12345.f) *sink = acc;
}

template <int R, int W>
static void run(const char* name, float4* in, float4* out, long bytes_total, float4* sink) {
  const long items = bytes_total / 16 / (R + W);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {2048, 8192, 65536}) {
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((mix_kernel<R, W>), dim3(blocks), dim3(256), 0, 0, in, out, items, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it && ms < best) best = ms;
    }
    printf("%-28s read:write %d:%d  %6d workgroups  %7.3f ms  %7.1f GB/s\n", name, R, W, blocks, best, items * 16.0 * (R + W) / best / 1e6);
  }
}

int main() {
  const long bytes = 4L << 30;  // per direction
  float4 *in, *out, *sink;
  if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&sink, 16) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(in, 0, bytes); hipMemset(out, 0, bytes);
  run<1, 0>("read only", in, out, bytes, sink);
  run<0, 1>("write only", in, out, bytes, sink);
  run<1, 1>("copy", in, out, bytes, sink);
  run<1, 4>("expand (1 read, 4 written)", in, out, bytes, sink);
  run<4, 1>("reduce (4 read, 1 written)", in, out, bytes, sink);
  run<2, 1>("2 read, 1 written", in, out, bytes, sink);
  return 0;
}
