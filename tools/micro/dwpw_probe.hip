// Development probe: the product's fused depthwise->pointwise kernel (included from csrc) on synthetic tensors of the
// shapes the plans bind, with phases knocked out by -D switches (OCR_DWPW_NO_G / _NO_TAPS / _NO_MMA, OCR_PROBE_NOSTORE):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../cpp-paddle-ocr_amd/csrc [-D...] -o dwpw_probe dwpw_probe.hip
//   dwpw_probe N H W K SH SW Cin Cout      (H, W = depthwise INPUT size)
// -DOCR_TU_H16=1: the precision "fp16" build of the kernel (f16 tensors and fragments)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void run(int N, int H, int W, int K, int SH, int SW, int cin, int cout) {
  const int P = K / 2, OH = (H + 2 * P - K) / SH + 1, OW = (W + 2 * P - K) / SW + 1;
  const int tiles = (cout + 31) / 32, nt = conv_nt_for(tiles), NTtot = (tiles + nt - 1) / nt * nt, C8 = cin / 8;
  const long Min = (long)N * H * W, M = (long)N * OH * OW;
  float *x, *y, *w, *dw, *vec;
  CK(hipMalloc(&x, Min * cin * 4)); CK(hipMalloc(&y, M * cout * 4)); CK(hipMalloc(&w, (size_t)C8 * NTtot * 64 * 16));
  CK(hipMalloc(&dw, (size_t)K * K * cin * 4)); CK(hipMalloc(&vec, 8192 * 4));
  CK(hipMemset(x, 0, Min * cin * 4)); CK(hipMemset(w, 0, (size_t)C8 * NTtot * 64 * 16)); CK(hipMemset(dw, 0, (size_t)K * K * cin * 4)); CK(hipMemset(vec, 0, 8192 * 4));
  DwPwArgs a{};
  a.c.out = y; a.c.wfrag = w; a.c.M = M; a.c.N = N; a.c.H = OH; a.c.W = OW; a.c.Cs_in = cin; a.c.C8 = C8; a.c.OH = OH; a.c.OW = OW;
  a.c.zeros = vec; a.c.Cs_out = cout; a.c.Cout = cout; a.c.CoutPadded = cout; a.c.ColsStore = cout; a.c.NTtot = NTtot; a.c.KH = a.c.KW = 1; a.c.out_mode = OUT_C8I;
  a.dw_in = x; a.dw_w = dw; a.H = H; a.W = W; a.K = K; a.SH = SH; a.SW = SW; a.PH = a.PW = P;
  a.dw_ep = LabEp{vec, 0.f, 0.f, 0};           // bias | hsw6 (the folded chain of an absorbed depthwise conv)
  a.pw_ep = LabEp{vec, 0.99f / 6.f, 0.01f, 1};  // bias | hsw6 | sfma
#ifdef OCR_TU_H16
  a.c.half = 1;  // (the buffers above keep their f32 sizes: zeros are zeros in either format)
#endif
  if (!launch_dwpw(a, 0, true)) { printf("shape not on the fused path\n"); return; }
  launch_dwpw(a, 0);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef OCR_DWPW_CLOCKS
  { unsigned long long z[8] = {}; CK(hipMemcpyToSymbol(HIP_SYMBOL(ocr_dwpw_clk), z, sizeof z)); }
#endif
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) launch_dwpw(a, 0);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double fl = 2.0 * M * cin * cout + 2.0 * M * K * K * cin, by = 4.0 * (Min * cin + M * cout);
  printf("N=%d %dx%d dw%dx%d s%d%d %d->%d: %.3f ms  %.1f TFLOP/s  %.0f GB/s\n", N, H, W, K, K, SH, SW, cin, cout, ms, fl / ms / 1e9, by / ms / 1e6);
#ifdef OCR_DWPW_CLOCKS
  { unsigned long long z[8]; CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(ocr_dwpw_clk), sizeof z));
    if (z[6]) printf("   clocks per wave-item: steps %.0f  advB %.0f  S %.0f  G %.0f  barrier %.0f  finish %.0f  (wave-items %llu)\n",
                     (double)z[0] / z[6], (double)z[1] / z[6], (double)z[2] / z[6], (double)z[3] / z[6], (double)z[4] / z[6], (double)z[5] / z[6], z[6]); }
#endif
  (void)hipFree(x); (void)hipFree(y); (void)hipFree(w); (void)hipFree(dw); (void)hipFree(vec);
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
