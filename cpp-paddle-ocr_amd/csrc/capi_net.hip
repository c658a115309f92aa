// C-ABI: runtime init, raw network taps, numerics probe.
#include <cstdio>
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "capi_common.h"
#include "plans_embedded.inc"

// The pipeline keeps a dozen independent launch chains in flight (rec lanes for odd tensor widths, det lanes for mixed
// image sizes, copy stream).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default:
// more chains than queues serialise behind each other (measured: BASELINE configs[2] det 531 -> 357 ms per 512 images
// with 16).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the process; this runs
// when the library is loaded.  A value the user has set is left alone.
__attribute__((constructor)) static void ocr_runtime_env_defaults() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

namespace ocr {

const RtOptions& rt_options() {
  static RtOptions o;
  static std::once_flag once;
  std::call_once(once, [] {
    auto off = [](const char* n) { const char* e = getenv(n); return e && e[0] == '0'; };
    auto num = [](const char* n, long def) { const char* e = getenv(n); return e && *e ? atol(e) : def; };
    o.fuse = !off("OCR_FUSE");
    o.fuse_gap_min = num("OCR_FUSE_GAP_MIN", 64 * 1024);
    o.conv_small_nt = !off("OCR_CONV_SMALL_NT");
    o.conv_mt2 = !off("OCR_CONV_MT2");
    if (const char* e = getenv("OCR_CONV_MT2")) o.conv_mt2_force = e[0] == 'f';
    o.conv_c24 = !off("OCR_CONV_C24");
    { const char* e = getenv("OCR_DW_LDS"); o.dw_lds = !(e && e[0] == '0'); }
    { const char* e = getenv("OCR_DWPW2"); o.dwpw2 = !(e && e[0] == '0'); }
    { const char* e = getenv("OCR_XDW"); o.xdw = !(e && e[0] == '0'); }
    { const char* e = getenv("OCR_MFMA_X16"); o.mfma_x16 = !(e && e[0] == '0'); }
    { const char* e = getenv("OCR_ATTN_LINE"); o.attn_line = !(e && e[0] == '0'); }
    o.dwpw_items = (int)num("OCR_DWPW_ITEMS", 32);
    o.dwpw_force_upw = (int)num("OCR_DWPW_FORCE_UPW", 0);
    if (const char* e = getenv("OCR_DWPW_T4")) o.dwpw_t4_thin = e[0] == 't';
    o.trace_slice = (int)num("OCR_TRACE_SLICE", 0);
  });
  return o;
}
static std::mutex g_refuse_mu;
static std::string g_refuse;
static std::atomic<bool> g_refuse_on{false};
// a COPY, taken under the lock: chain and lane threads read this on every run while a test may set it (the flag keeps
// the common case - no fault injection - to one atomic load)
std::string rt_refuse_launch() {
  if (!g_refuse_on.load(std::memory_order_acquire)) return std::string();
  std::lock_guard<std::mutex> lk(g_refuse_mu);
  return g_refuse;
}
void rt_set_refuse_launch(const char* substr) {
  std::lock_guard<std::mutex> lk(g_refuse_mu);
  g_refuse = substr ? substr : "";
  g_refuse_on.store(!g_refuse.empty(), std::memory_order_release);
}

static std::atomic<int> g_wait_mode{-1};
int rt_wait_mode() {
  int m = g_wait_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("OCR_WAIT_MODE");
    m = e && e[0] == 's' ? 0 : 1;  // default: block (same images/s, a fraction of the host CPU)
    g_wait_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}
void rt_set_wait_mode(int m) { g_wait_mode.store(m ? 1 : 0, std::memory_order_relaxed); }

std::shared_mutex& capture_mutex() {
  static std::shared_mutex m;
  return m;
}

// ---- logical devices (hip_guard.h): OCR_DEVICE_MAP=p0,p1,... read once; default the identity over the visible devices
static const std::vector<int>& device_map() {
  static std::vector<int> map;
  static std::once_flag once;
  std::call_once(once, [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    if (const char* e = getenv("OCR_DEVICE_MAP")) {
      std::vector<int> m;
      bool ok = n > 0 && *e;
      for (const char* p = e; ok && *p;) {
        char* end = nullptr;
        const long v = strtol(p, &end, 10);
        ok = end != p && v >= 0 && v < n && m.size() < 64;
        if (ok) m.push_back((int)v);
        p = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') ok = false;
      }
      if (ok && !m.empty()) { map = m; return; }  // (a malformed map is ignored: the identity)
    }
    for (int i = 0; i < n; ++i) map.push_back(i);
  });
  return map;
}
static thread_local int g_logical_device = -1;
int rt_device_count() { return (int)device_map().size(); }
int rt_physical_device(int logical) { return logical >= 0 && logical < rt_device_count() ? device_map()[logical] : -1; }
int rt_current_device() { return g_logical_device; }
hipError_t rt_set_device(int logical) {
  const int phys = rt_physical_device(logical);
  if (phys < 0) return hipErrorInvalidDevice;
  const hipError_t e = hipSetDevice(phys);
  if (e == hipSuccess) g_logical_device = logical;
  return e;
}

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

const char* embedded_plan(const char* kind) {
  if (!strcmp(kind, "det")) return kPlanDet;
  if (!strcmp(kind, "cls")) return kPlanCls;
  if (!strcmp(kind, "rec")) return kPlanRec;
  if (!strcmp(kind, "srv_det")) return kPlanSrv_det;  // BASELINE configs[4]: hand-written plans, NOT reference artifacts
  if (!strcmp(kind, "srv_rec")) return kPlanSrv_rec;
  return nullptr;
}

bool load_model_dir(const std::string& model_dir, const char* weights_override, const char* kind, WeightMap& w, std::string& err) {
  std::string model;
  for (const char* n : {"/inference.pdmodel", "/model.pdmodel"})
    if (file_exists(model_dir + n)) { model = model_dir + n; break; }
  if (model.empty()) { err = "No valid model file found in " + model_dir; return false; }
  // the graph must be the one the compiled-in plan was generated from (`plan <kind> ... graph_ops= graph_fnv=`)
  if (const char* plan = embedded_plan(kind)) {
    const char* po = strstr(plan, "graph_ops=");
    const char* pf = strstr(plan, "graph_fnv=");
    if (po && pf) {
      int nops = 0;
      unsigned long long fnv = 0;
      if (!pdmodel_graph_signature(model, nops, fnv, err)) return false;
      const int want_ops = atoi(po + 10);
      const unsigned long long want_fnv = strtoull(pf + 10, nullptr, 16);
      if (nops != want_ops || fnv != want_fnv) {
        char msg[320];
        snprintf(msg, sizeof msg, "%s is not the %s graph this library was built for (%d ops, signature %016llx; the plan expects %d, %016llx): "
                 "regenerate the plans with tools/make_plan.py and rebuild", model.c_str(), kind, nops, fnv, want_ops, want_fnv);
        err = msg;
        return false;
      }
    }
  }
  std::string params;
  if (weights_override) params = weights_override;
  else
    for (const char* n : {"/inference.pdiparams", "/model.pdiparams", "/synthetic.pdiparams"})
      if (file_exists(model_dir + n)) { params = model_dir + n; break; }
  if (params.empty() || !file_exists(params)) { err = "No parameter file found in " + model_dir; return false; }
  std::vector<std::string> names;
  if (!pdmodel_persistable_names(model, names, err)) return false;
  return pdiparams_read(params, names, w, err);
}

// Idle high-priority streams, kept for the life of the process.  Measured (round 3, tools/ab_cfg3.sh, ROCm 7.2 runtime,
// 256 mixed-size images): BASELINE configs[2]'s detector - host threads driving normal-priority streams ("lanes") with
// one small launch chain per image size - is bimodal: 145-150 ms or 190-270 ms per step, and which one depends on how
// many streams of which priority class the process created BEFORE the lanes' streams.  Rounds 1-2 were in the fast mode
// by accident (the recognizer's odd-width lanes were high-priority streams).  With one chain per pipeline: no such
// stream 193 ms, one created after the recognizer's stream 151 ms, one created before every other stream 196 ms, five
// more NORMAL streams 197 ms.  With two chains (the default): none 195 ms, one after all stage streams 266 ms, two
// 148 ms.  GPU_MAX_HW_QUEUES = 8 / 32 and DEBUG_HIP_DYNAMIC_QUEUES = 0 / 1 do not move the slow mode.  The runtime's
// mapping of streams to hardware queues is not documented; ocr_pipe_create uses the configuration that measured fast
// (two such streams after its stage objects).  The structural fix is a
// ragged detector batch (one launch list for all sizes, as the recognizer's), which needs no lanes at all.
// At most `want` such streams exist per device for the life of the process (ocr_pipe_create asks for two, every further
// pipeline of the process finds them there): nothing is created per handle, nothing leaks.
void priority_anchor(int device_id, int want) {
  static std::mutex mu;
  static hipStream_t anchor[64][2] = {};
  std::lock_guard<std::mutex> lk(mu);
  if (device_id < 0 || device_id >= 64) return;
  for (int k = 0; k < std::min(want, 2); ++k) {
    if (anchor[device_id][k]) continue;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest ||
        hipStreamCreateWithPriority(&anchor[device_id][k], hipStreamDefault, greatest) != hipSuccess) {
      anchor[device_id][k] = nullptr;
      (void)hipGetLastError();
      return;
    }
    (void)hipMemsetAsync(nullptr, 0, 0, anchor[device_id][k]);  // (touch the stream: the runtime creates its queue on first use)
    (void)hipGetLastError();
  }
}

}  // namespace ocr

using namespace ocr;

struct ocr_net {
  Net net;
  hipStream_t stream = nullptr;
  float* x_dev = nullptr;
  size_t x_cap = 0;
  int device = 0;
};

extern "C" {

const char* ocr_last_error(void) { return ocr::g_last_error.c_str(); }

int ocr_rt_device_count(void) { return rt_device_count(); }
int ocr_rt_set_wait_mode(int mode) { if (mode != 0 && mode != 1) return fail(OCR_ERR_ARG, "wait mode: 0 (spin) or 1 (block)"); rt_set_wait_mode(mode); return OCR_OK; }
int ocr_rt_get_wait_mode(void) { return rt_wait_mode(); }

int ocr_rt_init(int device_id) {
  (void)rt_options();  // the environment is read here, once
  const int n = rt_device_count();
  if (n <= 0) return fail(OCR_ERR_DEVICE, "no HIP device visible: this library has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(OCR_ERR_ARG, "device_id out of range");
  CAPI_HIP(rt_set_device(device_id));
  hipDeviceProp_t prop;
  CAPI_HIP(hipGetDeviceProperties(&prop, rt_physical_device(device_id)));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(OCR_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this build targets gfx950 (MI355X) only");
  return OCR_OK;
}

int ocr_net_create(const char* kind, const char* model_dir, const char* weights, int device_id, ocr_net** out) {
  return ocr_net_create_precision(kind, model_dir, weights, device_id, "fp32", out);
}

int ocr_net_create_precision(const char* kind, const char* model_dir, const char* weights, int device_id, const char* precision,
                             ocr_net** out) {
  if (!kind || !model_dir || !out || !precision) return fail(OCR_ERR_ARG, "null argument");
  const bool half = !strcmp(precision, "fp16");
  if (!half && strcmp(precision, "fp32")) return fail(OCR_ERR_ARG, "precision must be fp32 or fp16");
  const char* plan = embedded_plan(kind);
  if (!plan) return fail(OCR_ERR_ARG, "kind must be det, cls or rec");
  int rc = ocr_rt_init(device_id);
  if (rc) return rc;
  WeightMap w;
  std::string err;
  if (!load_model_dir(model_dir, weights, kind, w, err)) return fail(OCR_ERR_MODEL, err);
  std::unique_ptr<ocr_net> h(new ocr_net());
  h->device = device_id;
  if (!h->net.load(plan, w, err, half)) return fail(OCR_ERR_MODEL, err);
  CAPI_HIP(g_stream_create(&h->stream));
  *out = h.release();
  return OCR_OK;
}

void ocr_net_destroy(ocr_net* h) {
  if (!h) return;
  (void)rt_set_device(h->device);
  if (h->x_dev) (void)g_free(h->x_dev);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int ocr_net_forward(ocr_net* h, const float* x, int N, int H, int W, int keep_all) {
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
  h->net.set_keep_all(keep_all < 0 || keep_all > 2 ? 1 : keep_all);
  std::string err;
  if (!h->net.run(h->x_dev, N, H, W, h->stream, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(g_stream_sync(h->stream));
  h->net.collect_timings();
  return OCR_OK;
}

int ocr_net_forward_ragged(ocr_net* h, const float* x, int N, int H, const int* widths, int keep_all) {
  if (!h || !x || !widths || N <= 0 || H <= 0) return fail(OCR_ERR_ARG, "bad argument");
  CAPI_HIP(rt_set_device(h->device));
  size_t n = 0;
  for (int i = 0; i < N; ++i) {
    if (widths[i] <= 0) return fail(OCR_ERR_ARG, "bad line width");
    n += (size_t)H * widths[i] * 3;
  }
  if (n > h->x_cap) {
    if (h->x_dev) (void)g_free(h->x_dev);
    h->x_dev = nullptr;
    h->x_cap = 0;
    CAPI_HIP(g_malloc(&h->x_dev, n * sizeof(float)));
    h->x_cap = n;
  }
  CAPI_HIP(hipMemcpyAsync(h->x_dev, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  h->net.set_keep_all(keep_all < 0 || keep_all > 2 ? 1 : keep_all);
  std::string err;
  if (!h->net.run_ragged(h->x_dev, H, widths, N, h->stream, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(g_stream_sync(h->stream));
  h->net.collect_timings();
  return OCR_OK;
}

int ocr_net_forward_ragged_images(ocr_net* h, const float* x, int N, const int* heights, const int* widths, int keep_all) {
  if (!h || !x || !widths || !heights || N <= 0) return fail(OCR_ERR_ARG, "bad argument");
  if (keep_all != 0 && keep_all != 2) return fail(OCR_ERR_ARG, "a ragged batch of images runs the production launch list (keep_all 0 or 2)");
  CAPI_HIP(rt_set_device(h->device));
  size_t n = 0;
  for (int i = 0; i < N; ++i) {
    if (widths[i] <= 0 || heights[i] <= 0) return fail(OCR_ERR_ARG, "bad image size");
    n += (size_t)heights[i] * widths[i] * 3;
  }
  if (n > h->x_cap) {
    if (h->x_dev) (void)g_free(h->x_dev);
    h->x_dev = nullptr;
    h->x_cap = 0;
    CAPI_HIP(g_malloc(&h->x_dev, n * sizeof(float)));
    h->x_cap = n;
  }
  CAPI_HIP(hipMemcpyAsync(h->x_dev, x, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
  h->net.set_keep_all(keep_all);
  std::string err;
  if (!h->net.run_ragged_images(h->x_dev, heights, widths, N, h->stream, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(g_stream_sync(h->stream));
  h->net.collect_timings();
  return OCR_OK;
}

int ocr_net_num_tensors(ocr_net* h) { return h ? h->net.ntensors() : 0; }
int ocr_net_tensor_exists(ocr_net* h, int tid) { return h && h->net.materialised(tid) ? 1 : 0; }

int ocr_net_fetch(ocr_net* h, int tid, float* out, size_t cap_floats, int dims[4]) {
  if (!h || !out || !dims) return fail(OCR_ERR_ARG, "null argument");
  CAPI_HIP(rt_set_device(h->device));
  std::vector<float> host;
  std::string err;
  if (!h->net.fetch_logical(tid, host, dims, h->stream, err)) return fail(OCR_ERR_DEVICE, err);
  if (host.size() > cap_floats) return fail(OCR_ERR_CAPACITY, "output buffer too small");
  memcpy(out, host.data(), host.size() * sizeof(float));
  return OCR_OK;
}

int ocr_net_timing(ocr_net* h, int enable) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  h->net.enable_timing(enable != 0);
  h->net.reset_timings();
  return OCR_OK;
}

int ocr_net_timing_report(ocr_net* h, char* buf, size_t cap) {
  if (!h || !buf) return fail(OCR_ERR_ARG, "null argument");
  size_t off = 0;
  for (auto& kv : h->net.timings()) {
    int n = snprintf(buf + off, cap > off ? cap - off : 0, "%s %.6f %ld %.0f %.0f\n", kv.first.c_str(), kv.second.ms,
                     kv.second.count, kv.second.flops, kv.second.bytes);
    if (n < 0 || off + n >= cap) return fail(OCR_ERR_CAPACITY, "report buffer too small");
    off += n;
  }
  if (off < cap) buf[off] = 0;
  return OCR_OK;
}

int ocr_selftest_refuse_launch(const char* substr) {
  rt_set_refuse_launch(substr);
  return OCR_OK;
}

// The per-device dynamic-LDS memo of the kernel launchers (lds_attr.h) on faked devices: no HIP call is made, so this
// runs without a GPU - the one-process-several-devices path of the worker pool otherwise needs a multi-GPU lease.
namespace {
int g_fake_dev = 0;
int g_fake_calls = 0, g_fake_last_bytes = 0;
bool g_fake_refuse = false;
}  // namespace
int ocr_selftest_lds_memo(void) {
  LdsAttrHooks hk;  // handed to raise_dynamic_lds explicitly: the launchers of running pipelines never see the fakes
  hk.get_device = [] { return g_fake_dev; };
  hk.set_attribute = [](const void*, int bytes) { ++g_fake_calls; g_fake_last_bytes = bytes; return !g_fake_refuse; };
  static LdsAttrMemo memo;  // (a fresh one per process: the test runs once)
  const void* k = (const void*)&g_fake_dev;
  std::string why;
  auto expect = [&](bool cond, const char* what) { if (!cond && why.empty()) why = what; };
  g_fake_calls = 0;
  g_fake_dev = 0;
  expect(raise_dynamic_lds(k, 70000, memo, &hk) && g_fake_calls == 1, "first request on device 0 must raise the attribute");
  expect(raise_dynamic_lds(k, 70000, memo, &hk) && g_fake_calls == 1, "the same request again is answered from the memo");
  expect(raise_dynamic_lds(k, 65000, memo, &hk) && g_fake_calls == 1, "a smaller request is covered by the raised limit");
  expect(raise_dynamic_lds(k, 72704, memo, &hk) && g_fake_calls == 2 && g_fake_last_bytes == 72704, "a larger request must raise again");
  g_fake_dev = 1;
  expect(raise_dynamic_lds(k, 70000, memo, &hk) && g_fake_calls == 3, "another device has its own memo entry");
  g_fake_dev = 2;
  g_fake_refuse = true;
  expect(!raise_dynamic_lds(k, 70000, memo, &hk) && g_fake_calls == 4, "a refusal is reported");
  g_fake_refuse = false;
  expect(!raise_dynamic_lds(k, 70000, memo, &hk) && g_fake_calls == 4, "and remembered: the caller keeps its small-LDS path on that device");
  g_fake_dev = 0;
  expect(raise_dynamic_lds(k, 72704, memo, &hk) && g_fake_calls == 4, "device 0 is unaffected by the other devices' entries");
  g_fake_dev = -1;
  expect(!raise_dynamic_lds(k, 1, memo, &hk), "no current device");
  return why.empty() ? OCR_OK : fail(OCR_ERR_DEVICE, why);
}

int ocr_probe(const float* a, const float* b, float* out, int n) {
  if (!a || !b || !out || n <= 0) return fail(OCR_ERR_ARG, "bad argument");
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  CAPI_HIP(g_malloc(&da, n * sizeof(float)));
  CAPI_HIP(g_malloc(&db, n * sizeof(float)));
  CAPI_HIP(g_malloc(&dout, 8 * (size_t)n * sizeof(float)));
  CAPI_HIP(g_memcpy(da, a, n * sizeof(float), hipMemcpyHostToDevice));
  CAPI_HIP(g_memcpy(db, b, n * sizeof(float), hipMemcpyHostToDevice));
  launch_probe(da, db, dout, n, nullptr);
  CAPI_HIP(g_memcpy(out, dout, 8 * (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  (void)g_free(da); (void)g_free(db); (void)g_free(dout);
  return OCR_OK;
}

}  // extern "C"
