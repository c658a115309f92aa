// Development probe: times the product's conv_mfma_kernel (included from csrc) on synthetic 1x1 convs with non-trivial
// data, nothing added to the kernel (conv_probe.hip's clock stamps change the compiler's vmcnt waits: not representative).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../cpp-paddle-ocr_amd/csrc -o conv_time conv_time.hip
//   conv_time M cin cout gate(0|1) [nt [mt(1|2) [lds(0|1)]]]      prints a checksum of the output: variants must agree
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels_net.hip"
using namespace ocr;
namespace ocr {
inline namespace h16 {  // (the precision "fp16" twins of the launchers: not linked into this probe)
bool launch_conv_mfma_h16(const ConvArgs&, const Epilogue&, int, hipStream_t) { return false; }
bool launch_conv_mfma_mt2_h16(const ConvArgs&, const Epilogue&, int, hipStream_t) { return false; }
bool launch_conv3x3_tile_h16(const ConvArgs&, const Epilogue&, int, hipStream_t, bool) { return false; }
bool launch_conv_rowsum_h16(const ConvRowsumArgs&, hipStream_t) { return false; }
void launch_stem_h16(const StemArgs&, const Epilogue&, hipStream_t) {}
void launch_dw_h16(const DwArgs&, const Epilogue&, hipStream_t) {}
void launch_ew_h16(const float*, float*, long, int, int, int, const Epilogue&, hipStream_t, int, RagLevel, bool) {}
void launch_gap_h16(const float*, float*, float*, int, int, int, int, hipStream_t, RagLevel, long, bool) {}
void launch_concat_h16(const ConcatArgs&, hipStream_t) {}
void launch_pool_h16(const PoolArgs&, hipStream_t) {}
void launch_ln_h16(const float*, float*, long, int, int, float, const float*, const float*, hipStream_t, bool) {}
void launch_attn_h16(const float*, float*, int, int, int, int, int, int, float, hipStream_t, RagLevel, bool) {}
void launch_det_tail_h16(const DetTailArgs&, hipStream_t) {}
bool launch_db_head_h16(const DbHeadArgs&, int, hipStream_t) { return false; }
void launch_c8i_to_plain_h16(const float*, float*, long, int, int, hipStream_t, bool) {}
}
const RtOptions& rt_options() { static RtOptions o; return o; }
std::string rt_refuse_launch() { return std::string(); }
std::shared_mutex& capture_mutex() { static std::shared_mutex m; return m; }
int rt_current_device() { return 0; }  // (the probes run on device 0 without the library's logical-device table)
int rt_physical_device(int d) { return d; }
int rt_device_count() { return 1; }
hipError_t rt_set_device(int d) { return hipSetDevice(d); }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  p[i] = ((int)(h & 0xffff) - 32768) * (1.0f / 65536.0f);
}
static void fill(float* p, size_t n, unsigned seed) { hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, seed); }
int main(int argc, char** argv) {
  if (argc < 5) { printf("conv_time M cin cout gate [nt]\n"); return 1; }
  const long M = atol(argv[1]);
  const int cin = atoi(argv[2]), cout = atoi(argv[3]), gate = atoi(argv[4]);
  const int cs_in = (cin + 7) / 8 * 8, cs_out = (cout + 7) / 8 * 8;
  const int tiles = (cs_out + 31) / 32, nt = argc > 5 ? atoi(argv[5]) : conv_nt_for(tiles), NTtot = (tiles + nt - 1) / nt * nt, C8 = cs_in / 8;
  float *x, *y, *w, *z, *bn, *g;
  CK(hipMalloc(&x, (M * cs_in + 64) * 4)); CK(hipMalloc(&y, M * cs_out * 4)); CK(hipMalloc(&w, (size_t)C8 * NTtot * 64 * 4 * 4));
  CK(hipMalloc(&z, 4096 * 4)); CK(hipMalloc(&bn, 8192 * 4)); CK(hipMalloc(&g, 64 * cs_in * 4));
  fill(x, M * cs_in + 64, 1); fill(w, (size_t)C8 * NTtot * 64 * 4, 2); fill(bn, 8192, 3); fill(g, 64 * cs_in, 4);
  CK(hipMemset(z, 0, 4096 * 4));
  ConvArgs a{};
  a.in = x; a.out = y; a.wfrag = w; a.zeros = z; a.M = M; a.N = 64; a.H = 1; a.W = (int)(M / 64); a.Cs_in = cs_in; a.C8 = C8;
  a.OH = 1; a.OW = a.W; a.Cs_out = cs_out; a.Cout = cout; a.CoutPadded = cs_out; a.ColsStore = cs_out; a.NTtot = NTtot;
  a.KH = a.KW = 1; a.PH = a.PW = 0; a.out_mode = OUT_C8I;
  if (gate) { a.gate = g; a.gate_hw = (int)((M + 63) / 64); }
  Epilogue ep{};
  ep.n = 2;
  ep.st[0].kind = EP_BN; ep.st[0].v0 = bn; ep.st[0].v1 = bn + 4096;
  ep.st[1].kind = EP_ACT; ep.st[1].act = ACT_HSWISH; ep.st[1].p0 = 1.f; ep.st[1].p1 = 0.f;
  const int mt = argc > 6 ? atoi(argv[6]) : 1;
  const int lds = argc > 7 ? atoi(argv[7]) : 0;
  auto launch = [&]() { if (lds) { launch_conv_lds(a, ep, nt, 0); return; } if (!(mt == 2 ? launch_conv_mfma_mt2(a, ep, nt, 0) : launch_conv_mfma(a, ep, nt, 0))) { printf("refused\n"); exit(1); } };
  launch(); launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  const int inner = getenv("CONV_TIME_ITERS") ? atoi(getenv("CONV_TIME_ITERS")) : 4;  // (long runs: warm clocks)
#ifdef OCR_CONV_CLKRATE
  { unsigned long long z[2] = {0, 0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(ocr_conv_clkrate), z, sizeof z)); }
#endif
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < inner; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= inner; sum += ms; if (ms < best) best = ms;
  }
#ifdef OCR_CONV_CLKRATE
  { unsigned long long z[2]; CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(ocr_conv_clkrate), sizeof z));
    if (z[1]) printf("   average shader clock over the workgroups' lives: %.0f MHz\n", 100.0 * z[0] / z[1]); }
#endif
  // checksum of the output so that variants can be compared for identical results
  std::vector<float> hy(1 << 16);
  CK(hipMemcpy(hy.data(), y + (M / 2) * cs_out, hy.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long cks = 1469598103934665603ull;
  for (float v : hy) { unsigned u; memcpy(&u, &v, 4); cks = (cks ^ u) * 1099511628211ull; }
  const double fl = 2.0 * M * cin * cout;
  printf("M=%ld %d->%d gate=%d nt=%d mt=%d lds=%d: best %.3f ms (avg %.3f) %.1f TFLOP/s  checksum %016llx\n", M, cin, cout, gate, nt, mt, lds, best, sum / 5, fl / best / 1e9, cks);
  return 0;
}
