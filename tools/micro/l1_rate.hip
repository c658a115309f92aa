// Development microbenchmark: rate of dwordx4 global loads per CU for the access shapes of the conv K loop.
//   W-like: 64 lanes x 16 B contiguous (1 KB per instruction);  X-like: 32 rows (stride 960 B) x 2 x 16 B.
//   footprint per workgroup chosen to sit in L1 (16 KB), in L2 (1 MB) or to stream from HBM.
// hipcc --offload-arch=gfx950 -O3 -o l1_rate l1_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ base, float* out, int iters, int mode, long foot16, long wg_stride16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4* p = base + (long)blockIdx.x * wg_stride16;
  float4 acc = make_float4(0, 0, 0, 0);
  long off;
  if (mode == 0) off = wave * 64 + lane;                       // contiguous 1 KB per wave
  else off = (long)(wave * 32 + (lane & 31)) * 60 + (lane >> 5);  // 32 rows x 960 B, two 16-B halves
  const long step = mode == 0 ? 256 : 2;                        // next 4 KB block / next octet of the rows
  long o = off;
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    const float4 v = p[o];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    o += step;
    if (o >= foot16) o = off;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  const long total = 1L << 30;  // 1 GiB buffer
  float4* d; float* o;
  hipMalloc(&d, total); hipMemset(d, 0, total); hipMalloc(&o, 4096 * 256 * 4);
  const int blocks = 256 * 4, iters = 4096;
  struct Cfg { const char* name; int mode; long foot; long stride; } cfgs[] = {
    {"W-like, 16 KB/WG (L1)", 0, 16 << 10, 16 << 10}, {"W-like, 512 KB/WG (L2)", 0, 512 << 10, 512 << 10},
    {"W-like, shared 128 KB by all WGs (L2 hot)", 0, 128 << 10, 0},
    {"X-like, 120 KB/WG rows of 960 B, first 480 B (L1/L2)", 1, 30, 128 << 10},
    {"X-like, rows of 960 B, full row walk", 1, 60, 128 << 10},
  };
  for (auto& c : cfgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(d, o, iters, c.mode, c.foot / (c.mode == 0 ? 16 : 1), c.stride / 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<blocks, 256>>>(d, o, iters, c.mode, c.foot / (c.mode == 0 ? 16 : 1), c.stride / 16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * 16;
    printf("%-55s %.3f ms  %.2f TB/s  = %.1f B/clk/CU @2.4GHz\n", c.name, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4e3 * 1e0);
  }
  return 0;
}
