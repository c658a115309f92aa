// Development probe: srv_mlp_kernel with per-phase shader-cycle stamps (-DSRV_MLP_CLOCKS): where a stage's time goes.
//   make -C tools/micro srv_mlp_probe ;  srv_mlp_probe M C
#define SRV_MLP_CLOCKS 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "srv_kernels.hip"
namespace ocr {
std::shared_mutex& capture_mutex() { static std::shared_mutex m; return m; }
int rt_current_device() { return 0; }
hipError_t rt_set_device(int) { return hipSetDevice(0); }
int rt_physical_device(int) { return 0; }
int rt_device_count() { return 1; }
int rt_wait_mode() { return 0; }
void rt_set_wait_mode(int) {}
}
using namespace ocr; using namespace ocr::srv;
int main(int argc, char** argv) {
  const long M = argc > 1 ? atol(argv[1]) : 983040;
  const int C = argc > 2 ? atoi(argv[2]) : 192, H = 4 * C;
  auto pad = [](int n) { return (n + 255) & ~255; };
  std::vector<_Float16> hx((size_t)M * C), w1((size_t)(C / 64) * pad(H) * 64), w2((size_t)(H / 64) * pad(C) * 64);
  unsigned s = 1u; auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) / 1000.0f; };
  for (auto& v : hx) v = (_Float16)rnd();
  for (auto& v : w1) v = (_Float16)(rnd() * 0.05f);
  for (auto& v : w2) v = (_Float16)(rnd() * 0.05f);
  _Float16 *dx, *dw1, *dw2, *dy; float *b1, *b2; unsigned long long* clk;
  hipMalloc(&dx, hx.size() * 2); hipMalloc(&dy, hx.size() * 2); hipMalloc(&dw1, w1.size() * 2); hipMalloc(&dw2, w2.size() * 2);
  hipMalloc(&b1, pad(H) * 4); hipMalloc(&b2, pad(C) * 4); hipMemset(b1, 0, pad(H) * 4); hipMemset(b2, 0, pad(C) * 4);
  const size_t nb = (M + 127) / 128;
  hipMalloc(&clk, nb * 8 * 8 * 8); hipMemset(clk, 0, nb * 8 * 8 * 8);
  hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w1.data(), w1.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice);
  MlpArgs a{dx, hx.size() * 2, dw1, w1.size() * 2, pad(H), dw2, w2.size() * 2, pad(C), b1, b2, dy, M};
  a.clocks = clk;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&] {
    if (C == 192) hipLaunchKernelGGL((srv_mlp_kernel<192, false>), dim3(nb), dim3(512), MlpGeom<192>::LDS, 0, a);
    else if (C == 256) hipLaunchKernelGGL((srv_mlp_kernel<256, false>), dim3(nb), dim3(512), MlpGeom<256>::LDS, 0, a);
    else hipLaunchKernelGGL((srv_mlp_kernel<512, false>), dim3(nb), dim3(512), MlpGeom<512>::LDS, 0, a);
  };
  hipFuncSetAttribute((const void*)srv_mlp_kernel<192, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MlpGeom<192>::LDS);
  hipFuncSetAttribute((const void*)srv_mlp_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MlpGeom<256>::LDS);
  hipFuncSetAttribute((const void*)srv_mlp_kernel<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MlpGeom<512>::LDS);
  go(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 3; ++i) go(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  std::vector<unsigned long long> hc(nb * 8 * 8);
  hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost);
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = 0; i < nb * 8; ++i) for (int k = 0; k < 8; ++k) t[k] += hc[i * 8 + k];
  const double n = nb * 8.0, stages = (H / 128) * (C / 64 + 2 * (C > 256 ? 2 : 1));
  printf("M=%ld C=%d: %.3f ms, %.1f TFLOP/s; per wave: life %.0f cycles over %g stages; per stage: dma wait %.0f, barrier %.0f, issue %.0f, compute %.0f; per CHUNK: fc1 matrix part %.0f, gelu epilogue %.0f\n",
         M, C, ms, 16.0 * M * C * C / ms / 1e9, t[4] / n, stages, t[0] / n / stages, t[1] / n / stages, t[2] / n / stages, t[3] / n / stages, t[5] / n / (H / 128), t[6] / n / (H / 128));
  return 0;
}
