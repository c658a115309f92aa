// C-ABI: detector / classifier / recognizer handles (include/ocr_hip.h).
#include <cstring>
#include <memory>

#include "capi_common.h"
#include "stages.h"

using namespace ocr;

struct ocr_det { DetStage s; };
struct ocr_cls { ClsStage s; };
struct ocr_rec { RecStage s; };

extern "C" {

void ocr_det_cfg_default(ocr_det_cfg* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  // literals of OCRWorker::OCRWorker, /root/reference/src/ocr_worker.cpp:21-35
  c->model_dir = nullptr;
  c->device_id = 0;
  c->limit_type = "max";
  c->limit_side_len = 512;
  c->det_db_thresh = 0.2;
  c->det_db_box_thresh = 0.4;
  c->det_db_unclip_ratio = 1.8;
  c->det_db_score_mode = "fast";
  c->use_dilation = 0;
  c->precision = "fp32";
  c->max_batch = 1;
}

int ocr_det_create(const ocr_det_cfg* c, ocr_det** out) {
  if (!c || !out || !c->model_dir) return fail(OCR_ERR_ARG, "null argument");
  DetConfig cfg;
  cfg.model_dir = c->model_dir;
  cfg.device = c->device_id;
  if (c->limit_type) cfg.limit_type = c->limit_type;
  cfg.limit_side_len = c->limit_side_len;
  cfg.thresh = c->det_db_thresh;
  cfg.box_thresh = c->det_db_box_thresh;
  cfg.unclip_ratio = c->det_db_unclip_ratio;
  if (c->det_db_score_mode) cfg.score_mode = c->det_db_score_mode;
  cfg.use_dilation = c->use_dilation;
  if (c->precision) cfg.precision = c->precision;
  cfg.max_batch = c->max_batch > 0 ? c->max_batch : 1;
  cfg.cv_compat = resolve_cv_compat(c->cv_compat);
  std::unique_ptr<ocr_det> h(new ocr_det());
  std::string err;
  int code = 0;
  if (!h->s.create(cfg, err, code)) return fail(code, err);
  *out = h.release();
  return OCR_OK;
}
void ocr_det_destroy(ocr_det* h) { delete h; }

int ocr_det_run_batch(ocr_det* h, const ocr_img* imgs, int count, int32_t* boxes, int cap, int* n, double times[3]) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  std::string err;
  const int rc = h->s.run(imgs, count, boxes, cap, n, times, err);
  return rc ? fail(rc, err) : OCR_OK;
}
int ocr_det_run(ocr_det* h, const ocr_img* img, int32_t* boxes, int cap, int* n, double times[3]) {
  return ocr_det_run_batch(h, img, 1, boxes, cap, n, times);
}
int ocr_det_last_shape(ocr_det* h, int* count, int* rows, int* cols) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  if (count) *count = h->s.last_count;
  if (rows) *rows = h->s.last_h;
  if (cols) *cols = h->s.last_w;
  return OCR_OK;
}
static int det_tap(ocr_det* h, int index, const void* base, size_t elem, size_t per_px, void* out, size_t cap_elems) {
  if (!h || !out) return fail(OCR_ERR_ARG, "null argument");
  if (index < 0 || index >= h->s.last_count || !base) return fail(OCR_ERR_ARG, "no such image in the last run");
  const size_t cnt = (size_t)h->s.last_h * h->s.last_w * per_px;
  if (cnt > cap_elems) return fail(OCR_ERR_CAPACITY, "output buffer too small");
  CAPI_HIP(g_memcpy(out, (const char*)base + (size_t)index * cnt * elem, cnt * elem, hipMemcpyDeviceToHost));
  return OCR_OK;
}
int ocr_det_prob_map(ocr_det* h, int index, float* out, size_t cap) {
  return det_tap(h, index, h ? h->s.prob_dev() : nullptr, sizeof(float), 1, out, cap);
}
int ocr_det_bitmap(ocr_det* h, int index, uint8_t* out, size_t cap) {
  return det_tap(h, index, h ? h->s.bitmap_dev() : nullptr, 1, 1, out, cap);
}
int ocr_det_resized(ocr_det* h, int index, uint8_t* out, size_t cap) {
  return det_tap(h, index, h ? h->s.resized_dev() : nullptr, 1, 3, out, cap);
}
int ocr_det_post(ocr_det* h, const float* prob, int rows, int cols, int src_rows, int src_cols, int32_t* boxes, int cap,
                 int* n) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  std::string err;
  const int rc = h->s.post_only(prob, rows, cols, src_rows, src_cols, boxes, cap, n, err);
  return rc ? fail(rc, err) : OCR_OK;
}

// ---- self-test taps: the device's ClipperOffset / UnClip in front of vectors the caller holds (tests/golden/unclip_ref*:
// outputs of the reference's own compiled src/clipper.cpp).  The calling thread's current device (ocr_rt_init).
int ocr_selftest_unclip(const int32_t* quads, const double* deltas, int n, int64_t* out, int cap, int* counts, double* trig) {
  if (!quads || !deltas || !out || !counts || n <= 0 || cap <= 0) return fail(OCR_ERR_ARG, "bad argument");
  int *dq = nullptr, *dc = nullptr;
  double *dd = nullptr, *dt = nullptr;
  if (trig) CAPI_HIP(g_malloc(&dt, (size_t)n * 3 * sizeof(double)));
  long long* dout = nullptr;
  CAPI_HIP(g_malloc(&dq, (size_t)n * 8 * sizeof(int)));
  CAPI_HIP(g_malloc(&dd, (size_t)n * sizeof(double)));
  CAPI_HIP(g_malloc(&dc, (size_t)n * sizeof(int)));
  CAPI_HIP(g_malloc(&dout, (size_t)n * cap * 2 * sizeof(long long)));
  CAPI_HIP(g_memcpy(dq, quads, (size_t)n * 8 * sizeof(int), hipMemcpyHostToDevice));
  CAPI_HIP(g_memcpy(dd, deltas, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
  CAPI_HIP(hipMemset(dout, 0, (size_t)n * cap * 2 * sizeof(long long)));
  launch_selftest_clipper(dq, dd, n, dout, cap, dc, dt, nullptr);
  CAPI_HIP(hipGetLastError());
  if (trig) { CAPI_HIP(g_memcpy(trig, dt, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost)); (void)g_free(dt); }
  CAPI_HIP(g_memcpy(out, dout, (size_t)n * cap * 2 * sizeof(long long), hipMemcpyDeviceToHost));
  CAPI_HIP(g_memcpy(counts, dc, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  (void)g_free(dq); (void)g_free(dd); (void)g_free(dc); (void)g_free(dout);
  return OCR_OK;
}
int ocr_selftest_unclip_box(const float* boxes, float unclip_ratio, int n, float* out14, int* status) {
  if (!boxes || !out14 || !status || n <= 0) return fail(OCR_ERR_ARG, "bad argument");
  float *db = nullptr, *dout = nullptr;
  int* dst = nullptr;
  CAPI_HIP(g_malloc(&db, (size_t)n * 8 * sizeof(float)));
  CAPI_HIP(g_malloc(&dout, (size_t)n * 14 * sizeof(float)));
  CAPI_HIP(g_malloc(&dst, (size_t)n * 2 * sizeof(int)));
  CAPI_HIP(g_memcpy(db, boxes, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice));
  launch_selftest_unclip_box(db, unclip_ratio, n, dout, dst, nullptr);
  CAPI_HIP(hipGetLastError());
  CAPI_HIP(g_memcpy(out14, dout, (size_t)n * 14 * sizeof(float), hipMemcpyDeviceToHost));
  CAPI_HIP(g_memcpy(status, dst, (size_t)n * 2 * sizeof(int), hipMemcpyDeviceToHost));
  (void)g_free(db); (void)g_free(dout); (void)g_free(dst);
  return OCR_OK;
}

void ocr_cls_cfg_default(ocr_cls_cfg* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->cls_thresh = 0.98;   // ocr_worker.cpp:45
  c->cls_batch_num = 8;   // ocr_worker.cpp:47
  c->precision = "fp32";
}
int ocr_cls_create(const ocr_cls_cfg* c, ocr_cls** out) {
  if (!c || !out || !c->model_dir) return fail(OCR_ERR_ARG, "null argument");
  ClsConfig cfg;
  cfg.model_dir = c->model_dir;
  cfg.device = c->device_id;
  cfg.thresh = c->cls_thresh;
  cfg.batch_num = c->cls_batch_num > 0 ? c->cls_batch_num : 1;
  if (c->precision) cfg.precision = c->precision;
  std::unique_ptr<ocr_cls> h(new ocr_cls());
  std::string err;
  int code = 0;
  if (!h->s.create(cfg, err, code)) return fail(code, err);
  *out = h.release();
  return OCR_OK;
}
void ocr_cls_destroy(ocr_cls* h) { delete h; }
int ocr_cls_run(ocr_cls* h, const ocr_img* imgs, int n, int* labels, float* scores, double times[3]) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  std::string err;
  const int rc = h->s.run(imgs, n, labels, scores, times, err);
  return rc ? fail(rc, err) : OCR_OK;
}
int ocr_cls_probs(ocr_cls* h, float* out, size_t cap) {
  if (!h || !out) return fail(OCR_ERR_ARG, "null argument");
  if (h->s.tap_probs.size() > cap) return fail(OCR_ERR_CAPACITY, "output buffer too small");
  memcpy(out, h->s.tap_probs.data(), h->s.tap_probs.size() * sizeof(float));
  return OCR_OK;
}

void ocr_rec_cfg_default(ocr_rec_cfg* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  c->rec_batch_num = 16;  // ocr_worker.cpp:60-62
  c->rec_img_h = 28;
  c->rec_img_w = 192;
  c->precision = "fp32";
}
int ocr_rec_create(const ocr_rec_cfg* c, ocr_rec** out) {
  if (!c || !out || !c->model_dir || !c->label_path) return fail(OCR_ERR_ARG, "null argument");
  RecConfig cfg;
  cfg.model_dir = c->model_dir;
  cfg.label_path = c->label_path;
  cfg.device = c->device_id;
  cfg.batch_num = c->rec_batch_num;
  cfg.sort_mode = c->sort_mode;
  cfg.img_h = c->rec_img_h;
  cfg.img_w = c->rec_img_w;
  if (c->precision) cfg.precision = c->precision;
  std::unique_ptr<ocr_rec> h(new ocr_rec());
  std::string err;
  int code = 0;
  if (!h->s.create(cfg, err, code)) return fail(code, err);
  *out = h.release();
  return OCR_OK;
}
void ocr_rec_destroy(ocr_rec* h) { delete h; }
int ocr_rec_run(ocr_rec* h, const ocr_img* imgs, int n, int32_t* ids, int max_len, int* lens, float* scores,
                double times[3]) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  std::string err;
  const int rc = h->s.run(imgs, n, ids, max_len, lens, scores, times, err);
  return rc ? fail(rc, err) : OCR_OK;
}
const char* ocr_rec_label(ocr_rec* h, int id) {
  if (!h || id < 0 || id >= (int)h->s.labels().size()) return nullptr;
  return h->s.labels()[id].c_str();
}
int ocr_rec_num_classes(ocr_rec* h) { return h ? (int)h->s.labels().size() : 0; }
int ocr_rec_steps(ocr_rec* h, int index, int32_t* amax, float* pmax, int cap, int* T) {
  if (!h || !T) return fail(OCR_ERR_ARG, "null argument");
  if (index < 0 || index >= (int)h->s.tap_T.size()) return fail(OCR_ERR_ARG, "no such line in the last run");
  const int t = h->s.tap_T[index];
  *T = t;
  if (t > cap) return fail(OCR_ERR_CAPACITY, "output buffer too small");
  if (amax) memcpy(amax, h->s.tap_amax.data() + h->s.tap_off[index], t * sizeof(int));
  if (pmax) memcpy(pmax, h->s.tap_pmax.data() + h->s.tap_off[index], t * sizeof(float));
  return OCR_OK;
}

}  // extern "C"
