// Development probe: runs the product's conv_mfma_kernel (included from csrc) on a synthetic 1x1 conv
// and reports TFLOP/s plus the per-wave phase breakdown from clock64() stamps.
// NOTE (round 3): the stamps' stores make the compiler replace the K loop's counted vmcnt waits by full drains, so
// TIMES and A/B comparisons from this probe are not the product kernel's - use conv_time.hip for those; this probe is
// only good for the relative length of a wave's phases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DOCR_CONV_PROBE -I../../cpp-paddle-ocr_amd/csrc -o conv_probe conv_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "kernels_net.hip"
using namespace ocr;
namespace ocr {
const RtOptions& rt_options() { static RtOptions o; return o; }
std::shared_mutex& capture_mutex() { static std::shared_mutex m; return m; }
}
static int g_gate = 0, g_nt = 0;  // argv[5] = 1: gated input (rec op 30); argv[6]: NT override
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static void run(long M, int cin, int cout, int variant, int epi) {
  const int cs_in = (cin + 7) / 8 * 8, cs_out = (cout + 7) / 8 * 8;
  const int tiles = (cs_out + 31) / 32, nt = g_nt ? g_nt : conv_nt_for(tiles), NTtot = (tiles + nt - 1) / nt * nt, C8 = cs_in / 8;
  float *x, *y, *w, *z, *bn;
  CK(hipMalloc(&x, (M * cs_in + 64) * 4)); CK(hipMalloc(&y, M * cs_out * 4)); CK(hipMalloc(&w, (size_t)C8 * NTtot * 64 * 4 * 4));
  CK(hipMalloc(&z, 4096 * 4)); CK(hipMalloc(&bn, 8192 * 4));
  CK(hipMemset(x, 0, (M * cs_in + 64) * 4)); CK(hipMemset(w, 0, (size_t)C8 * NTtot * 64 * 16)); CK(hipMemset(z, 0, 4096 * 4)); CK(hipMemset(bn, 0, 8192 * 4));
  ConvArgs a{};
  a.in = x; a.out = y; a.wfrag = w; a.zeros = z; a.M = M; a.N = 1; a.H = 1; a.W = (int)M; a.Cs_in = cs_in; a.C8 = C8;
  a.OH = 1; a.OW = (int)M; a.Cs_out = cs_out; a.Cout = cout; a.CoutPadded = cs_out; a.ColsStore = cs_out; a.NTtot = NTtot;
  a.KH = a.KW = 1; a.PH = a.PW = 0; a.out_mode = OUT_C8I;
  if (g_gate) {
    float* g;
    CK(hipMalloc(&g, 64 * cs_in * 4)); CK(hipMemset(g, 0, 64 * cs_in * 4));
    a.gate = g; a.gate_hw = (int)((M + 63) / 64);
  }
  Epilogue ep{};
  if (epi) {
    ep.n = epi >= 2 ? 2 : 1;
    ep.st[0].kind = EP_BN; ep.st[0].v0 = bn; ep.st[0].v1 = bn + 4096;
    ep.st[1].kind = EP_ACT; ep.st[1].act = epi == 2 ? ACT_RELU : ACT_HSWISH; ep.st[1].p0 = 1.f; ep.st[1].p1 = 0.f;
  }
  const long nwaves = ((M + 127) / 128) * (NTtot / nt) * 4;
  long long* probe;
  CK(hipMalloc(&probe, nwaves * 8 * 8));
  CK(hipMemset(probe, 0, nwaves * 8 * 8));
  long long* nullp = nullptr;
  auto launch = [&]() { launch_conv_mfma(a, ep, nt, 0); };
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_probe), &nullp, sizeof(nullp)));
  launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double fl = 2.0 * M * cin * cout;
  printf("M=%ld %d->%d nt=%d variant=%d epi=%d: %.3f ms %.1f TFLOP/s\n", M, cin, cout, nt, variant, (int)epi, ms, fl / ms / 1e9);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_probe), &probe, sizeof(probe)));
  launch();
  CK(hipDeviceSynchronize());
  std::vector<long long> h8(nwaves * 8);
  CK(hipMemcpy(h8.data(), probe, nwaves * 64, hipMemcpyDeviceToHost));
  std::vector<long long> h(nwaves * 4);
  double p04 = 0, p45 = 0, p51 = 0; long pc = 0;
  for (long i = 0; i < nwaves; ++i) { for (int k = 0; k < 4; ++k) h[i * 4 + k] = h8[i * 8 + k]; if (h8[i * 8 + 3]) { p04 += h8[i * 8 + 4] - h8[i * 8]; p45 += h8[i * 8 + 5] - h8[i * 8 + 4]; p51 += h8[i * 8 + 1] - h8[i * 8 + 5]; ++pc; } }
  if (pc) printf("   prologue split: entry->staging %.0f  staging+barrier %.0f  setup+first loads %.0f\n", p04 / pc, p45 / pc, p51 / pc);
  long long* chk = nullptr;
  CK(hipMemcpyFromSymbol(&chk, HIP_SYMBOL(g_conv_probe), sizeof(chk)));
  printf("   probe ptr %p (symbol %p) first %lld %lld %lld %lld\n", (void*)probe, (void*)chk, h[0], h[1], h[2], h[3]);
  double s01 = 0, s12 = 0, s23 = 0;
  long cnt = 0;
  long long tmin = -1, tmax = 0;
  for (long i = 0; i < nwaves; ++i) {
    if (!h[i * 4 + 3]) continue;
    s01 += h[i * 4 + 1] - h[i * 4]; s12 += h[i * 4 + 2] - h[i * 4 + 1]; s23 += h[i * 4 + 3] - h[i * 4 + 2];
    if (tmin < 0 || h[i * 4] < tmin) tmin = h[i * 4];
    tmax = std::max(tmax, h[i * 4 + 3]);
    ++cnt;
  }
  // distribution of the loop phase and concurrency: waves alive at the median time
  if (variant == 0 && cnt) {
    std::vector<long long> lp;
    for (long i = 0; i < nwaves; ++i) if (h[i * 4 + 3]) lp.push_back(h[i * 4 + 2] - h[i * 4 + 1]);
    std::sort(lp.begin(), lp.end());
    const long long tmid = (tmin + tmax) / 2;
    long alive = 0;
    for (long i = 0; i < nwaves; ++i) if (h[i * 4 + 3] && h[i * 4] <= tmid && h[i * 4 + 3] >= tmid) ++alive;
    printf("   loop clk p10 %lld p50 %lld p90 %lld; waves alive at mid-kernel %ld\n", lp[lp.size() / 10], lp[lp.size() / 2], lp[lp.size() * 9 / 10], alive);
  }
  if (variant == 0)
    printf("   waves %ld  prologue %.0f  loop %.0f  epilogue %.0f clk(avg/wave)  kernel span %lld clk  (clock64 units)\n", cnt, s01 / cnt, s12 / cnt,
           s23 / cnt, tmax - tmin);
  hipFree(x); hipFree(y); hipFree(w); hipFree(z); hipFree(bn); hipFree(probe);
}

int main(int argc, char** argv) {
  if (argc >= 4) {  // conv_probe M cin cout [epi]
    if (argc > 5) g_gate = atoi(argv[5]);
    if (argc > 6) g_nt = atoi(argv[6]);
    run(atol(argv[1]), atoi(argv[2]), atoi(argv[3]), 0, argc > 4 ? atoi(argv[4]) : 3);
    return 0;
  }
  run(1966080, 240, 240, 0, 0);
  run(1966080, 240, 240, 0, 3);
  run(491520, 480, 480, 0, 3);
  return 0;
}
