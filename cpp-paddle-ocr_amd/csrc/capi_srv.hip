// C-ABI of the server networks' raw taps (BASELINE configs[4]; include/ocr_hip.h "server networks").
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <sstream>

#include "capi_common.h"
#include "srv_net.h"

using namespace ocr;

namespace ocr {
// the parameter table of a server plan: its `# param <name> <dims>` lines (there is no .pdmodel to take names from)
static std::vector<std::string> srv_param_names(const char* plan) {
  std::vector<std::string> names;
  std::stringstream ss(plan);
  std::string line;
  while (std::getline(ss, line))
    if (line.rfind("# param ", 0) == 0) {
      std::stringstream ls(line.substr(8));
      std::string n;
      ls >> n;
      names.push_back(n);
    }
  std::sort(names.begin(), names.end());  // the order .pdiparams records are written in
  return names;
}
std::string server_arch(const std::string& model_dir) {
  if (file_exists(model_dir + "/inference.pdmodel") || file_exists(model_dir + "/model.pdmodel")) return "";
  FILE* f = fopen((model_dir + "/arch.txt").c_str(), "r");
  if (!f) return "";
  char buf[64] = {0};
  const size_t n = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  std::string t(buf, n);
  while (!t.empty() && (t.back() == '\n' || t.back() == '\r' || t.back() == ' ')) t.pop_back();
  return (t == "srv_det" || t == "srv_rec") ? t : "";
}
bool load_server_model_dir(const std::string& model_dir, const char* kind, WeightMap& w, std::string& err) {
  const char* plan = embedded_plan(kind);
  if (!plan) { err = std::string("no server plan ") + kind; return false; }
  std::string params;
  for (const char* n : {"/inference.pdiparams", "/model.pdiparams", "/synthetic.pdiparams"})
    if (file_exists(model_dir + n)) { params = model_dir + n; break; }
  if (params.empty()) { err = "No parameter file found in " + model_dir; return false; }
  return pdiparams_read(params, srv_param_names(plan), w, err);
}
}  // namespace ocr

struct ocr_srv_net {
  SrvNet net;
  hipStream_t stream = nullptr;
  float* x_dev = nullptr;
  size_t x_cap = 0;
  int device = 0;
};

extern "C" {

int ocr_srv_net_create(const char* kind, const char* model_dir, int device_id, const char* precision, ocr_srv_net** out) {
  if (!kind || !model_dir || !out || !precision) return fail(OCR_ERR_ARG, "null argument");
  const bool half = !strcmp(precision, "fp16");
  if (!half && strcmp(precision, "fp32")) return fail(OCR_ERR_ARG, "precision must be fp16 (the mode BASELINE configs[4] names) or fp32 (the parity twin)");
  const std::string pk = std::string("srv_") + kind;
  const char* plan = embedded_plan(pk.c_str());
  if (!plan) return fail(OCR_ERR_ARG, "kind must be det or rec");
  int rc = ocr_rt_init(device_id);
  if (rc) return rc;
  WeightMap w;
  std::string err;
  if (!load_server_model_dir(model_dir, pk.c_str(), w, err)) return fail(OCR_ERR_MODEL, err);
  std::unique_ptr<ocr_srv_net> h(new ocr_srv_net());
  h->device = device_id;
  if (!h->net.load(plan, w, half, err)) return fail(OCR_ERR_MODEL, err);
  CAPI_HIP(g_stream_create(&h->stream));
  *out = h.release();
  return OCR_OK;
}

void ocr_srv_net_destroy(ocr_srv_net* h) {
  if (!h) return;
  (void)rt_set_device(h->device);
  if (h->x_dev) (void)g_free(h->x_dev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int ocr_srv_net_forward(ocr_srv_net* h, const float* x, int N, int H, int W, int keep_all) {
  if (!h || !x || N <= 0 || H <= 0 || W <= 0) return fail(OCR_ERR_ARG, "bad argument");
  CAPI_HIP(rt_set_device(h->device));
  const size_t n = (size_t)N * H * W * 3;
  if (n > h->x_cap) {
    if (h->x_dev) (void)g_free(h->x_dev);
    h->x_dev = nullptr;
    h->x_cap = 0;
    CAPI_HIP(g_malloc(&h->x_dev, n * sizeof(float)));
    h->x_cap = n;
  }
  CAPI_HIP(hipMemcpyAsync(h->x_dev, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  h->net.set_keep_all(keep_all != 0);
  std::string err;
  if (!h->net.run(h->x_dev, N, H, W, h->stream, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(g_stream_sync(h->stream));
  h->net.collect_timings();
  return OCR_OK;
}
/* the same network run again on the input already uploaded (timing loops) */
int ocr_srv_net_rerun(ocr_srv_net* h, int N, int H, int W, int iters) {
  if (!h || !h->x_dev || (size_t)N * H * W * 3 > h->x_cap) return fail(OCR_ERR_ARG, "no input resident");
  CAPI_HIP(rt_set_device(h->device));
  std::string err;
  for (int i = 0; i < iters; ++i)
    if (!h->net.run(h->x_dev, N, H, W, h->stream, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(g_stream_sync(h->stream));
  h->net.collect_timings();
  return OCR_OK;
}

int ocr_srv_net_num_tensors(ocr_srv_net* h) { return h ? h->net.ntensors() : 0; }

int ocr_srv_net_fetch(ocr_srv_net* h, int tid, float* out, size_t cap_floats, int dims[4]) {
  if (!h || !out || !dims) return fail(OCR_ERR_ARG, "null argument");
  CAPI_HIP(rt_set_device(h->device));
  std::vector<float> host;
  std::string err;
  if (!h->net.fetch_logical(tid, host, dims, h->stream, err)) return fail(OCR_ERR_ARG, err);
  if (host.size() > cap_floats) return fail(OCR_ERR_CAPACITY, "output buffer too small");
  memcpy(out, host.data(), host.size() * sizeof(float));
  return OCR_OK;
}

int ocr_srv_net_timing(ocr_srv_net* h, int enable) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  h->net.enable_timing(enable != 0);
  h->net.reset_timings();
  return OCR_OK;
}

int ocr_srv_net_timing_report(ocr_srv_net* h, char* buf, size_t cap) {
  if (!h || !buf) return fail(OCR_ERR_ARG, "null argument");
  size_t off = 0;
  for (auto& kv : h->net.timings()) {
    int n = snprintf(buf + off, cap > off ? cap - off : 0, "%s %.6f %ld %.0f %.0f\n", kv.first.c_str(), kv.second.ms, kv.second.count,
                     kv.second.flops, kv.second.bytes);
    if (n < 0 || off + n >= cap) return fail(OCR_ERR_CAPACITY, "report buffer too small");
    off += n;
  }
  if (off < cap) buf[off] = 0;
  return OCR_OK;
}

}  // extern "C"
