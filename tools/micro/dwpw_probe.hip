// Development probe: the product's fused depthwise->pointwise kernel (included from csrc) on synthetic tensors of the
// shapes the plans bind, with phases knocked out by -D switches (OCR_DWPW_NO_G / _NO_TAPS / _NO_MMA, OCR_PROBE_NOSTORE):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../cpp-paddle-ocr_amd/csrc [-D...] -o dwpw_probe dwpw_probe.hip
//   dwpw_probe N H W K SH SW Cin Cout      (H, W = depthwise INPUT size)
// -DOCR_TU_H16=1: the precision "fp16" build of the kernel (f16 tensors and fragments)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels_dwpw.hip"
using namespace ocr;
#ifndef OCR_TU_H16
namespace ocr { inline namespace h16 { bool launch_dwpw_h16(const DwPwArgs&, hipStream_t, bool) { return false; } } }  // (the f32 launcher's twin: not linked into this probe)
#endif
#ifdef OCR_TU_H16
static bool probe_launch_h16(const DwPwArgs& a, hipStream_t s, bool query = false) { return launch_dwpw_h16(a, s, query); }
#define launch_dwpw probe_launch_h16
#endif
namespace ocr {
const RtOptions& rt_options() { static RtOptions o; return o; }
std::shared_mutex& capture_mutex() { static std::shared_mutex m; return m; }
int rt_current_device() { return 0; }  // (the probes run on device 0 without the library's logical-device table)
int rt_physical_device(int d) { return d; }
int rt_device_count() { return 1; }
hipError_t rt_set_device(int d) { return hipSetDevice(d); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void run(int N, int H, int W, int K, int SH, int SW, int cin, int cout) {
  const int P = K / 2, OH = (H + 2 * P - K) / SH + 1, OW = (W + 2 * P - K) / SW + 1;
  const int tiles = (cout + 31) / 32, nt = conv_nt_for(tiles), NTtot = (tiles + nt - 1) / nt * nt, C8 = cin / 8;
  const long Min = (long)N * H * W, M = (long)N * OH * OW;
  float *x, *y, *w, *dw, *vec, *dwq, *dwq32;
  CK(hipMalloc(&x, Min * cin * 4)); CK(hipMalloc(&y, M * cout * 4)); CK(hipMalloc(&w, (size_t)C8 * NTtot * 64 * 16));
  CK(hipMalloc(&dw, (size_t)K * K * cin * 4)); CK(hipMalloc(&vec, 8192 * 4 + 65536)); CK(hipMalloc(&dwq, (size_t)(K * K + 1) * cin * 4)); CK(hipMemset(dwq, 0, (size_t)(K * K + 1) * cin * 4));
  CK(hipMalloc(&dwq32, (size_t)(K * K + 1) * cin * 4)); CK(hipMemset(dwq32, 0, (size_t)(K * K + 1) * cin * 4));
  CK(hipMemset(x, 0, Min * cin * 4)); CK(hipMemset(w, 0, (size_t)C8 * NTtot * 64 * 16)); CK(hipMemset(dw, 0, (size_t)K * K * cin * 4)); CK(hipMemset(vec, 0, 8192 * 4 + 65536));
  if (getenv("PROBE_RANDOM")) {  // seeded data: the output checksum below must not depend on the kernel form
    std::vector<float> hx((size_t)Min * cin), hw((size_t)C8 * NTtot * 64 * 4), hd((size_t)K * K * cin), hv(8192), hq((size_t)(K * K + 1) * cin);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.2f;
    for (auto& v : hd) v = rnd() * 0.5f;
    for (auto& v : hv) v = rnd() * 0.1f;
    for (int ch = 0; ch < cin / 16; ++ch)
      for (int t = 0; t <= K * K; ++t)
        for (int i = 0; i < 16; ++i) hq[((size_t)ch * (K * K + 1) + t) * 16 + i] = t < K * K ? hd[(size_t)t * cin + ch * 16 + i] : hv[ch * 16 + i];
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hd.data(), hd.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vec, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    if (cin % 32 == 0) {
      for (int ch = 0; ch < cin / 32; ++ch)
        for (int t = 0; t <= K * K; ++t)
          for (int i = 0; i < 32; ++i) hq[((size_t)ch * (K * K + 1) + t) * 32 + i] = t < K * K ? hd[(size_t)t * cin + ch * 32 + i] : hv[ch * 32 + i];
      CK(hipMemcpy(dwq32, hq.data(), hq.size() * 4, hipMemcpyHostToDevice));
    }
  }
  DwPwArgs a{};
  a.c.out = y; a.c.wfrag = w; a.c.M = M; a.c.N = N; a.c.H = OH; a.c.W = OW; a.c.Cs_in = cin; a.c.C8 = C8; a.c.OH = OH; a.c.OW = OW;
  a.c.zeros = vec + 8192; a.c.Cs_out = cout; a.c.Cout = cout; a.c.CoutPadded = cout; a.c.ColsStore = cout; a.c.NTtot = NTtot; a.c.KH = a.c.KW = 1; a.c.out_mode = OUT_C8I;
  a.dw_in = x; a.dw_w = dw;
  if (!getenv("PROBE_FORM1")) { a.dw_wq16 = dwq; if (cin % 32 == 0) a.dw_wq32 = dwq32; }  /* (the LDS-DMA form where it is instantiated) */
  a.H = H; a.W = W; a.K = K; a.SH = SH; a.SW = SW; a.PH = a.PW = P;
  a.dw_ep = LabEp{vec, 0.f, 0.f, 0};           // bias | hsw6 (the folded chain of an absorbed depthwise conv)
  a.pw_ep = LabEp{vec, 0.99f / 6.f, 0.01f, 1};  // bias | hsw6 | sfma
#ifdef OCR_TU_H16
  a.c.half = 1;  // (the buffers above keep their f32 sizes: zeros are zeros in either format)
#endif
#ifndef OCR_TU_H16
  if (getenv("PROBE_ATTR")) {
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, (const void*)dwpw2_kernel<5, 1, 1, 16, true, 4, 3, 2, 2, false, false>);
    printf("dwpw2 attributes: %s  regs %d  static lds %zu  max threads %d\n", hipGetErrorString(e), fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
    e = hipFuncGetAttributes(&fa, (const void*)dwpw_kernel<5, 1, 1, 16, true, 4, true, 2, 2, 2, false, false>);
    printf("dwpw attributes: %s  regs %d\n", hipGetErrorString(e), fa.numRegs);
  }
#endif
  if (!launch_dwpw(a, 0, true)) { printf("shape not on the fused path (%s)\n", hipGetErrorString(hipGetLastError())); return; }
  launch_dwpw(a, 0);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef OCR_DWPW_CLOCKS
  { unsigned long long z[8] = {}; CK(hipMemcpyToSymbol(HIP_SYMBOL(ocr_dwpw_clk), z, sizeof z)); }
#endif
  CK(hipEventRecord(e0));
  const int iters = getenv("PROBE_ITERS") ? atoi(getenv("PROBE_ITERS")) : 5;  // (long runs: for the power / clock sampler, tools/power_sample.py)
  for (int i = 0; i < iters; ++i) launch_dwpw(a, 0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  const double fl = 2.0 * M * cin * cout + 2.0 * M * K * K * cin, by = 4.0 * (Min * cin + M * cout);
  printf("N=%d %dx%d dw%dx%d s%d%d %d->%d: %.3f ms  %.1f TFLOP/s  %.0f GB/s\n", N, H, W, K, K, SH, SW, cin, cout, ms, fl / ms / 1e9, by / ms / 1e6);
#ifdef OCR_DWPW_CLKRATE
  { unsigned long long z[2]; CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(ocr_dwpw_clkrate), sizeof z));
    if (z[1]) printf("   average shader clock over the workgroups' lives: %.0f MHz\n", 100.0 * z[0] / z[1]);
    z[0] = z[1] = 0; CK(hipMemcpyToSymbol(HIP_SYMBOL(ocr_dwpw_clkrate), z, sizeof z)); }
#endif
  if (getenv("PROBE_RANDOM")) {
    std::vector<unsigned> hy((size_t)M * cout);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long cs = 0; double sum = 0;
    for (size_t i = 0; i < hy.size(); ++i) { cs = cs * 1099511628211ull ^ hy[i]; float f; memcpy(&f, &hy[i], 4); sum += f; }
    printf("   checksum %016llx  sum %.6f\n", cs, sum);
  }
#ifdef OCR_DWPW_CLOCKS
  { unsigned long long z[8]; CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(ocr_dwpw_clk), sizeof z));
    if (z[6]) printf("   clocks per wave-item: steps %.0f  advB %.0f  S %.0f  G %.0f  barrier %.0f  finish %.0f  (wave-items %llu)\n",
                     (double)z[0] / z[6], (double)z[1] / z[6], (double)z[2] / z[6], (double)z[3] / z[6], (double)z[4] / z[6], (double)z[5] / z[6], z[6]); }
#endif
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(w); (void)hipFree(dw); (void)hipFree(vec); (void)hipFree(dwq); (void)hipFree(dwq32);
}

int main(int argc, char** argv) {
  if (argc >= 9) { run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), atoi(argv[8])); return 0; }
  run(1872, 24, 160, 3, 1, 1, 16, 32);    // rec.01
  run(1872, 24, 160, 3, 1, 1, 32, 64);    // rec.03
  run(1872, 24, 160, 3, 1, 1, 64, 64);    // rec.05
  run(1872, 24, 160, 3, 2, 1, 64, 128);   // rec.07
  run(1872, 12, 160, 3, 1, 1, 128, 128);  // rec.09
  run(1872, 12, 160, 3, 1, 2, 128, 240);  // rec.11
  run(1872, 12, 80, 5, 1, 1, 240, 240);   // rec.13
  run(64, 480, 480, 3, 1, 1, 16, 32);     // det.01
  run(64, 240, 240, 3, 1, 1, 48, 48);     // det.05
  run(64, 60, 60, 5, 1, 1, 192, 192);     // det.13
  return 0;
}
