// Development probe: the server networks' GEMM kernels (srv_kernels.hip included as it stands) on one 1x1 / linear shape with
// seeded f16 data, every tile configuration in turn, interleaved rounds in ONE process (cdna guide rule 24).
//   make -C tools/micro srv_gemm_probe          (or: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../cpp-paddle-ocr_amd/csrc -o srv_gemm_probe srv_gemm_probe.hip)
//   srv_gemm_probe M K N [res 0|1] [rounds]      prints ms, TFLOP/s, algorithmic GB/s per configuration and a checksum (configurations must agree)
// Knock-outs (build with -D...): SRV_PROBE_NOSTORE (no output stores), SRV_PROBE_NOEPI (no epilogue at all)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "srv_kernels.hip"

namespace ocr {  // what the library provides beside the kernels
std::shared_mutex& capture_mutex() { static std::shared_mutex m; return m; }
int rt_current_device() { return 0; }
hipError_t rt_set_device(int) { return hipSetDevice(0); }
int rt_physical_device(int) { return 0; }
int rt_device_count() { return 1; }
int rt_wait_mode() { return 0; }
void rt_set_wait_mode(int) {}
}  // namespace ocr

using namespace ocr;
using namespace ocr::srv;

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: srv_gemm_probe M K N [res] [rounds]\n"); return 2; }
  const long M = atol(argv[1]);
  const int K = atoi(argv[2]), N = atoi(argv[3]);
  const int res = argc > 4 ? atoi(argv[4]) : 0, rounds = argc > 5 ? atoi(argv[5]) : 5;
  const int npad = (N + 255) & ~255, BK = 64, nkt = (K + BK - 1) / BK;
  std::vector<_Float16> hx((size_t)M * K), hw((size_t)nkt * npad * BK, (_Float16)0.f), hr((size_t)M * N);
  unsigned s = 12345u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) / 1000.0f; };
  for (auto& v : hx) v = (_Float16)rnd();
  for (auto& v : hr) v = (_Float16)rnd();
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const int kt = k / BK, kk = k % BK, g = kk / 8, e = kk % 8, slot = g ^ ((n >> 1) & 7);
      hw[(((size_t)kt * npad + n) * 8 + slot) * 8 + e] = (_Float16)(rnd() * 0.1f);
    }
  _Float16 *dx, *dw, *dr, *dy;
  float* dbias;
  hipMalloc(&dx, hx.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dr, hr.size() * 2); hipMalloc(&dy, hr.size() * 2);
  hipMalloc(&dbias, npad * 4);
  hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dr, hr.data(), hr.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dbias, 0, npad * 4);
  GemmArgs a;
  a.x = dx; a.x_bytes = hx.size() * 2; a.w = dw; a.w_bytes = hw.size() * 2; a.y = dy; a.y_bytes = hr.size() * 2;
  a.M = M; a.K = K; a.nkt = nkt; a.Npad = npad; a.Ncols = N; a.Cs_out = N;
  a.N = 1; a.H = 1; a.W = (int)M; a.Cin = K; a.OH = 1; a.OW = (int)M;
  a.x1 = 1; a.bias = dbias; a.act = SACT_RELU;
  if (res) { a.res = dr; a.res_bytes = hr.size() * 2; a.res_up = 1; }
  const int nc = gemm_num_configs();
  std::vector<double> best(nc, 1e30);
  std::vector<double> sum(nc, 0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::string err;
  std::vector<_Float16> hy(hr.size());
  const int only = getenv("PROBE_CFG") ? atoi(getenv("PROBE_CFG")) : -1;  // (one configuration: profiler runs)
  for (int r = 0; r < rounds + 1; ++r)
    for (int c = 0; c < nc; ++c) {
      if (only >= 0 && c != only) continue;
      if (!gemm_config_ok(a, true, c)) continue;
      hipEventRecord(e0, nullptr);
      for (int it = 0; it < 3; ++it) launch_gemm(a, true, c, nullptr, err);
      hipEventRecord(e1, nullptr);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (r) best[c] = std::min(best[c], (double)ms / 3);
      if (r == 1) {
        hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost);
        double cs = 0;
        for (size_t i = 0; i < hy.size(); i += 97) cs += (double)(float)hy[i] * (1 + (i % 13));
        sum[c] = cs;
      }
    }
  const double fl = 2.0 * M * K * N, by = 2.0 * ((double)M * K + (double)M * N * (res ? 2 : 1) + (double)K * N);
  printf("M=%ld K=%d N=%d res=%d: %.2f GFLOP, %.1f MB algorithmic\n", M, K, N, res, fl / 1e9, by / 1e6);
  for (int c = 0; c < nc; ++c)
    if (best[c] < 1e29) printf("  %-22s %8.4f ms %8.1f TFLOP/s %8.0f GB/s   checksum %.4f\n", gemm_config_name(c), best[c], fl / best[c] / 1e9, by / best[c] / 1e6, sum[c]);
  return 0;
}
