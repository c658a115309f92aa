// Fused det -> crop -> [cls -> rotate] -> rec pipeline over device-resident images
// (OCRWorker::processRequest, /root/reference/src/ocr_worker.cpp:213-311) and its C-ABI.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>

#include "capi_common.h"
#include "crop.h"
#include "stages.h"

using namespace ocr;

namespace {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

struct ocr_pipe {
  DetStage det;
  RecStage rec;
  std::unique_ptr<ClsStage> cls;
  int device = 0;
  DevBuf<uint8_t> upload;
  DevBuf<RotDesc> rot_desc;
  DevBuf<int> rot_seg;
  int crop_mode = 0;  // OCR_CROP_BOUNDING_RECT | OCR_CROP_ROTATE
  DevBuf<uint8_t> crop_arena;
  DevBuf<WarpDesc> warp_desc;
  std::vector<int32_t> boxes;
  std::vector<int> nbox;

  // one det pass over `count` same-size images living at dev (+ optional prob maps), then crops/cls/rec
  int run_group(uint8_t* dev, int rows, int cols, int count, const float* dev_prob, std::vector<std::vector<ocr_word>>& out_words,
                std::vector<std::vector<int32_t>>& out_ids, double times[3], std::string& err) {
    const int cap = 1000;  // max_candidates bounds the boxes of one image (postprocess_op.cpp:260)
    boxes.resize((size_t)count * cap * 8);
    nbox.resize(count);
    const size_t row = (size_t)cols * 3, img_bytes = row * rows;
    double t0 = now_ms();
    int rc = det.run_device(dev, img_bytes, row, rows, cols, count, boxes.data(), cap, nbox.data(), nullptr, err, dev_prob);
    if (rc) return rc;
    double t1 = now_ms();
    times[0] += t1 - t0;
    // crop rectangles: cv::boundingRect(points) & image rect (ocr_worker.cpp:245-258)
    std::vector<LineSrc> lines;
    std::vector<int> seg(1, 0);
    std::vector<int> line_box;  // box index (within its image) of every line
    if (crop_mode == OCR_CROP_ROTATE) {
      rc = rotate_crops(dev, rows, cols, count, cap, lines, seg, line_box, err);
      if (rc) return rc;
    } else
    for (int i = 0; i < count; ++i) {
      for (int j = 0; j < nbox[i]; ++j) {
        const int32_t* b = &boxes[((size_t)i * cap + j) * 8];
        int x0 = b[0], x1 = b[0], y0 = b[1], y1 = b[1];
        for (int k = 1; k < 4; ++k) {
          x0 = std::min(x0, b[2 * k]); x1 = std::max(x1, b[2 * k]);
          y0 = std::min(y0, b[2 * k + 1]); y1 = std::max(y1, b[2 * k + 1]);
        }
        const int ix0 = std::max(x0, 0), iy0 = std::max(y0, 0);
        const int ix1 = std::min(x1 + 1, cols), iy1 = std::min(y1 + 1, rows);
        if (ix1 - ix0 > 0 && iy1 - iy0 > 0) {
          // `word.box = det_boxes[i]` pairs text k with box k of the image, even when an empty crop was
          // skipped before it (reference quirk, ocr_worker.cpp:293-299) — kept.
          line_box.push_back((int)lines.size() - seg.back());
          lines.push_back(LineSrc{dev + img_bytes * i, row, ix0, iy0, ix1 - ix0, iy1 - iy0});
        }
      }
      seg.push_back((int)lines.size());
    }
    out_words.assign(count, {});
    out_ids.assign(count, {});
    if (lines.empty()) return OCR_OK;
    if (cls) {
      std::vector<int> labels(lines.size());
      std::vector<float> scores(lines.size());
      rc = cls->run_lines(lines, labels.data(), scores.data(), err);
      if (rc) return rc;
      // rotate in request order, in place on the device copy (cv::rotate on ROI views aliasing the image):
      // one workgroup per image walks that image's rotations sequentially
      std::vector<RotDesc> rd;
      std::vector<int> rseg(1, 0);
      for (int i = 0; i < count; ++i) {
        for (int k = seg[i]; k < seg[i + 1]; ++k)
          if (labels[k] == 1)
            rd.push_back(RotDesc{const_cast<uint8_t*>(lines[k].img), lines[k].stride, lines[k].x, lines[k].y, lines[k].w, lines[k].h});
        rseg.push_back((int)rd.size());
      }
      if (!rd.empty()) {
        if (!rot_desc.ensure(rd.size(), err) || !rot_seg.ensure(rseg.size(), err)) return OCR_ERR_DEVICE;
        if (hipMemcpyAsync(rot_desc.p, rd.data(), rd.size() * sizeof(RotDesc), hipMemcpyHostToDevice, cls->stream()) != hipSuccess ||
            hipMemcpyAsync(rot_seg.p, rseg.data(), rseg.size() * sizeof(int), hipMemcpyHostToDevice, cls->stream()) != hipSuccess) {
          err = "rotation list upload failed";
          return OCR_ERR_DEVICE;
        }
        launch_rotate180_list(rot_desc.p, rot_seg.p, count, cls->stream());
      }
      if (hipStreamSynchronize(cls->stream()) != hipSuccess) { err = "cls stream sync failed"; return OCR_ERR_DEVICE; }
    }
    double t2 = now_ms();
    times[1] += t2 - t1;
    const int max_len = 256;
    std::vector<int32_t> ids(lines.size() * max_len);
    std::vector<int> lens(lines.size());
    std::vector<float> scores(lines.size());
    rc = rec.run_lines(lines, seg, ids.data(), max_len, lens.data(), scores.data(), err);
    if (rc) return rc;
    times[2] += now_ms() - t2;
    for (int i = 0; i < count; ++i) {
      for (int k = seg[i]; k < seg[i + 1]; ++k) {
        ocr_word w;
        memcpy(w.box, &boxes[((size_t)i * cap + line_box[k]) * 8], sizeof(w.box));
        w.ids_off = (int32_t)out_ids[i].size();
        w.ids_len = lens[k];
        w.confidence = scores[k];
        out_ids[i].insert(out_ids[i].end(), ids.begin() + (size_t)k * max_len, ids.begin() + (size_t)k * max_len + lens[k]);
        out_words[i].push_back(w);
      }
    }
    return OCR_OK;
  }

  // crop_mode OCR_CROP_ROTATE: every box becomes its own perspective-rectified image
  // (Utility::GetRotateCropImage, utility.cpp:137-190) in the crop arena
  int rotate_crops(const uint8_t* dev, int rows, int cols, int count, int cap, std::vector<LineSrc>& lines, std::vector<int>& seg,
                   std::vector<int>& line_box, std::string& err) {
    const size_t row = (size_t)cols * 3, img_bytes = row * rows;
    std::vector<WarpDesc> wd;
    std::vector<size_t> off;
    size_t total = 0;
    int max_px = 0;
    for (int i = 0; i < count; ++i) {
      for (int j = 0; j < nbox[i]; ++j) {
        CropPlan p;
        if (!plan_rotate_crop(rows, cols, &boxes[((size_t)i * cap + j) * 8], p)) continue;
        WarpDesc d;
        d.src = dev + img_bytes * i + (size_t)p.top * row + (size_t)p.left * 3;
        d.sstride = row; d.sw = p.sw; d.sh = p.sh; d.dst = nullptr; d.dw = p.dw; d.dh = p.dh; d.rot = p.rot; d.bw0 = p.bw0;
        memcpy(d.m, p.minv, sizeof(d.m));
        wd.push_back(d);
        off.push_back(total);
        total += ((size_t)p.dw * p.dh * 3 + 15) & ~(size_t)15;
        max_px = std::max(max_px, p.dw * p.dh);
        line_box.push_back(j);
        lines.push_back(LineSrc{nullptr, (size_t)p.ocols * 3, 0, 0, p.ocols, p.orows});
      }
      seg.push_back((int)lines.size());
    }
    if (wd.empty()) return OCR_OK;
    if (!crop_arena.ensure(total, err) || !warp_desc.ensure(wd.size(), err)) return OCR_ERR_DEVICE;
    for (size_t k = 0; k < wd.size(); ++k) {
      wd[k].dst = crop_arena.p + off[k];
      lines[k].img = wd[k].dst;
    }
    if (hipMemcpyAsync(warp_desc.p, wd.data(), wd.size() * sizeof(WarpDesc), hipMemcpyHostToDevice, det.stream()) != hipSuccess) {
      err = "crop list upload failed";
      return OCR_ERR_DEVICE;
    }
    launch_warp_crops(warp_desc.p, (int)wd.size(), max_px, det.stream());
    if (hipStreamSynchronize(det.stream()) != hipSuccess) { err = "crop kernel failed"; return OCR_ERR_DEVICE; }
    return OCR_OK;
  }
};

static int emit(const std::vector<std::vector<ocr_word>>& W, const std::vector<std::vector<int32_t>>& I, const std::vector<int>& order,
                ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids) {
  int wo = 0, io = 0;
  for (size_t i = 0; i < order.size(); ++i) {
    const auto& w = W[order[i]];
    const auto& d = I[order[i]];
    word_off[i] = wo;
    nwords[i] = (int)w.size();
    if (wo + (int)w.size() > cap_words || io + (int)d.size() > cap_ids) return fail(OCR_ERR_CAPACITY, "result buffers too small");
    for (size_t k = 0; k < w.size(); ++k) {
      words[wo + k] = w[k];
      words[wo + k].ids_off += io;
    }
    if (!d.empty()) memcpy(ids + io, d.data(), d.size() * sizeof(int32_t));
    wo += (int)w.size();
    io += (int)d.size();
  }
  return OCR_OK;
}

extern "C" {

void ocr_pipe_cfg_default(ocr_pipe_cfg* c) {
  if (!c) return;
  memset(c, 0, sizeof(*c));
  ocr_det_cfg_default(&c->det);
  ocr_cls_cfg_default(&c->cls);
  ocr_rec_cfg_default(&c->rec);
  c->enable_cls = 0;
  c->crop_mode = OCR_CROP_BOUNDING_RECT;
}

int ocr_pipe_create(const ocr_pipe_cfg* c, ocr_pipe** out) {
  if (!c || !out || !c->det.model_dir || !c->rec.model_dir || !c->rec.label_path) return fail(OCR_ERR_ARG, "null argument");
  if (c->enable_cls && !c->cls.model_dir) return fail(OCR_ERR_ARG, "cls model_dir missing");
  std::unique_ptr<ocr_pipe> h(new ocr_pipe());
  h->device = c->det.device_id;
  if (c->crop_mode != OCR_CROP_BOUNDING_RECT && c->crop_mode != OCR_CROP_ROTATE) return fail(OCR_ERR_ARG, "unknown crop_mode");
  h->crop_mode = c->crop_mode;
  std::string err;
  int code = 0;
  DetConfig d;
  d.model_dir = c->det.model_dir; d.device = c->det.device_id;
  if (c->det.limit_type) d.limit_type = c->det.limit_type;
  d.limit_side_len = c->det.limit_side_len; d.thresh = c->det.det_db_thresh; d.box_thresh = c->det.det_db_box_thresh;
  d.unclip_ratio = c->det.det_db_unclip_ratio;
  if (c->det.det_db_score_mode) d.score_mode = c->det.det_db_score_mode;
  d.use_dilation = c->det.use_dilation;
  if (c->det.precision) d.precision = c->det.precision;
  d.max_batch = c->det.max_batch > 0 ? c->det.max_batch : 1;
  if (!h->det.create(d, err, code)) return fail(code, err);
  RecConfig r;
  r.model_dir = c->rec.model_dir; r.label_path = c->rec.label_path; r.device = c->det.device_id;
  r.batch_num = c->rec.rec_batch_num; r.img_h = c->rec.rec_img_h; r.img_w = c->rec.rec_img_w; r.sort_mode = c->rec.sort_mode;
  if (c->rec.precision) r.precision = c->rec.precision;
  if (!h->rec.create(r, err, code)) return fail(code, err);
  h->rec.want_taps = false;
  if (c->enable_cls) {
    ClsConfig k;
    k.model_dir = c->cls.model_dir; k.device = c->det.device_id; k.thresh = c->cls.cls_thresh;
    k.batch_num = c->cls.cls_batch_num > 0 ? c->cls.cls_batch_num : 1;
    if (c->cls.precision) k.precision = c->cls.precision;
    h->cls.reset(new ClsStage());
    if (!h->cls->create(k, err, code)) return fail(code, err);
  }
  *out = h.release();
  return OCR_OK;
}
void ocr_pipe_destroy(ocr_pipe* h) { delete h; }

int ocr_pipe_run_device(ocr_pipe* h, const void* dev_bgr, int rows, int cols, int count, const float* dev_prob, ocr_word* words,
                        int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3]) {
  if (!h || !dev_bgr || rows <= 0 || cols <= 0 || count < 1 || !words || !word_off || !nwords || !ids)
    return fail(OCR_ERR_ARG, "bad argument");
  CAPI_HIP(hipSetDevice(h->device));
  double t[3] = {0, 0, 0};
  std::vector<std::vector<ocr_word>> W;
  std::vector<std::vector<int32_t>> I;
  std::string err;
  // the request's clone (OCRRequest copies the Mat, ocr_worker.h:28-29): cls rotation is in place on it
  const size_t bytes = (size_t)rows * cols * 3 * count;
  if (!h->upload.ensure(bytes, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(hipMemcpyAsync(h->upload.p, dev_bgr, bytes, hipMemcpyDeviceToDevice, h->det.stream()));
  const int rc = h->run_group(h->upload.p, rows, cols, count, dev_prob, W, I, t, err);
  if (rc) return fail(rc, err);
  if (times) { times[0] = t[0]; times[1] = t[1]; times[2] = t[2]; }
  std::vector<int> order(count);
  for (int i = 0; i < count; ++i) order[i] = i;
  return emit(W, I, order, words, cap_words, word_off, nwords, ids, cap_ids);
}

int ocr_pipe_run(ocr_pipe* h, const ocr_img* imgs, int count, ocr_word* words, int cap_words, int* word_off, int* nwords,
                 int32_t* ids, int cap_ids, double times[3]) {
  if (!h || !imgs || count < 1 || !words || !word_off || !nwords || !ids) return fail(OCR_ERR_ARG, "bad argument");
  for (int i = 0; i < count; ++i)
    if (!imgs[i].data || imgs[i].rows <= 0 || imgs[i].cols <= 0) return fail(OCR_ERR_ARG, "Empty image data provided");
  CAPI_HIP(hipSetDevice(h->device));
  // images of one size share a det pass
  std::map<std::pair<int, int>, std::vector<int>> by_size;
  for (int i = 0; i < count; ++i) by_size[{imgs[i].rows, imgs[i].cols}].push_back(i);
  std::vector<std::vector<ocr_word>> W(count);
  std::vector<std::vector<int32_t>> I(count);
  double t[3] = {0, 0, 0};
  std::string err;
  for (auto& kv : by_size) {
    const int rows = kv.first.first, cols = kv.first.second, n = (int)kv.second.size();
    const size_t row = (size_t)cols * 3, img_bytes = row * rows;
    if (!h->upload.ensure(img_bytes * n, err)) return fail(OCR_ERR_DEVICE, err);
    for (int k = 0; k < n; ++k) {
      const ocr_img& im = imgs[kv.second[k]];
      const size_t stride = im.row_stride ? im.row_stride : row;
      CAPI_HIP(hipMemcpy2DAsync(h->upload.p + img_bytes * k, row, im.data, stride, row, rows, hipMemcpyHostToDevice, h->det.stream()));
    }
    std::vector<std::vector<ocr_word>> w;
    std::vector<std::vector<int32_t>> d;
    const int rc = h->run_group(h->upload.p, rows, cols, n, nullptr, w, d, t, err);
    if (rc) return fail(rc, err);
    for (int k = 0; k < n; ++k) { W[kv.second[k]] = std::move(w[k]); I[kv.second[k]] = std::move(d[k]); }
  }
  if (times) { times[0] = t[0]; times[1] = t[1]; times[2] = t[2]; }
  std::vector<int> order(count);
  for (int i = 0; i < count; ++i) order[i] = i;
  return emit(W, I, order, words, cap_words, word_off, nwords, ids, cap_ids);
}

const char* ocr_pipe_label(ocr_pipe* h, int id) {
  if (!h || id < 0 || id >= (int)h->rec.labels().size()) return nullptr;
  return h->rec.labels()[id].c_str();
}

int ocr_pipe_det_shape(ocr_pipe* h, int rows, int cols, int* net_rows, int* net_cols) {
  if (!h || !net_rows || !net_cols) return fail(OCR_ERR_ARG, "null argument");
  float a, b;
  DetStage::resize_shape(rows, cols, h->det.cfg().limit_type, h->det.cfg().limit_side_len, *net_rows, *net_cols, a, b);
  return OCR_OK;
}

static std::vector<Net*> pipe_nets(ocr_pipe* h) {
  std::vector<Net*> v{&h->det.net()};
  for (int i = 0; i < h->rec.num_lanes(); ++i) v.push_back(&h->rec.lane_net(i));
  return v;
}
int ocr_pipe_timing(ocr_pipe* h, int enable) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  for (Net* net : pipe_nets(h)) {
    net->enable_timing(enable != 0);
    net->reset_timings();
  }
  return OCR_OK;
}
int ocr_pipe_timing_filter(ocr_pipe* h, const char* substr) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  const std::string f = substr ? substr : "";
  for (Net* net : pipe_nets(h)) net->set_timing_filter(f);
  return OCR_OK;
}
int ocr_pipe_timing_report(ocr_pipe* h, char* buf, size_t cap) {
  if (!h || !buf) return fail(OCR_ERR_ARG, "null argument");
  size_t off = 0;
  // the halves of a split rec launch (and equal shapes on different lanes) have the same instance names: one row per name
  std::map<std::string, KernelTiming> all;
  for (Net* net : pipe_nets(h))
    for (auto& kv : net->timings()) {
      KernelTiming& t = all[kv.first];
      t.ms += kv.second.ms; t.count += kv.second.count; t.flops += kv.second.flops; t.bytes += kv.second.bytes;
    }
  for (auto& kv : all) {
    int n = snprintf(buf + off, cap > off ? cap - off : 0, "%s %.6f %ld %.0f %.0f\n", kv.first.c_str(), kv.second.ms,
                     kv.second.count, kv.second.flops, kv.second.bytes);
    if (n < 0 || off + n >= cap) return fail(OCR_ERR_CAPACITY, "report buffer too small");
    off += n;
  }
  if (off < cap) buf[off] = 0;
  return OCR_OK;
}

int ocr_rotate_crop_shape(int rows, int cols, const int32_t* box, int* out_rows, int* out_cols) {
  if (!box || !out_rows || !out_cols) return fail(OCR_ERR_ARG, "null argument");
  CropPlan p;
  if (!plan_rotate_crop(rows, cols, box, p)) return fail(OCR_ERR_ARG, "box has no crop inside the image");
  *out_rows = p.orows;
  *out_cols = p.ocols;
  return OCR_OK;
}

int ocr_rotate_crop(const uint8_t* bgr, int rows, int cols, size_t row_stride, const int32_t* boxes, int n, uint8_t* out,
                    size_t out_cap, size_t* out_off, int* out_rows, int* out_cols) {
  if (!bgr || rows <= 0 || cols <= 0 || !boxes || n < 1 || !out || !out_off || !out_rows || !out_cols)
    return fail(OCR_ERR_ARG, "bad argument");
  const size_t row = (size_t)cols * 3, stride = row_stride ? row_stride : row;
  std::vector<WarpDesc> wd(n);
  std::vector<size_t> src_off(n);
  size_t total = 0;
  int max_px = 0;
  for (int k = 0; k < n; ++k) {
    CropPlan p;
    if (!plan_rotate_crop(rows, cols, boxes + 8 * k, p)) return fail(OCR_ERR_ARG, "box has no crop inside the image");
    WarpDesc& d = wd[k];
    d.sstride = row; d.sw = p.sw; d.sh = p.sh; d.dw = p.dw; d.dh = p.dh; d.rot = p.rot; d.bw0 = p.bw0;
    src_off[k] = (size_t)p.top * row + (size_t)p.left * 3;
    memcpy(d.m, p.minv, sizeof(d.m));
    out_off[k] = total;
    out_rows[k] = p.orows;
    out_cols[k] = p.ocols;
    total += (size_t)p.dw * p.dh * 3;
    max_px = std::max(max_px, p.dw * p.dh);
  }
  out_off[n] = total;
  if (total > out_cap) return fail(OCR_ERR_CAPACITY, "crop buffer too small");
  std::string err;
  DevBuf<uint8_t> dimg, dout;
  DevBuf<WarpDesc> ddesc;
  if (!dimg.ensure(row * rows, err) || !dout.ensure(total, err) || !ddesc.ensure(n, err)) return fail(OCR_ERR_DEVICE, err);
  for (int k = 0; k < n; ++k) {
    wd[k].src = dimg.p + src_off[k];
    wd[k].dst = dout.p + out_off[k];
  }
  CAPI_HIP(hipMemcpy2D(dimg.p, row, bgr, stride, row, rows, hipMemcpyHostToDevice));
  CAPI_HIP(hipMemcpy(ddesc.p, wd.data(), n * sizeof(WarpDesc), hipMemcpyHostToDevice));
  launch_warp_crops(ddesc.p, n, max_px, 0);
  CAPI_HIP(hipGetLastError());
  CAPI_HIP(hipMemcpy(out, dout.p, total, hipMemcpyDeviceToHost));
  return OCR_OK;
}

int ocr_dev_alloc(void** p, size_t bytes) {
  if (!p) return fail(OCR_ERR_ARG, "null argument");
  CAPI_HIP(hipMalloc(p, bytes));
  return OCR_OK;
}
int ocr_dev_free(void* p) { CAPI_HIP(hipFree(p)); return OCR_OK; }
int ocr_dev_upload(void* dst, const void* src, size_t bytes) { CAPI_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return OCR_OK; }
int ocr_dev_download(void* dst, const void* src, size_t bytes) { CAPI_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return OCR_OK; }
int ocr_dev_sync(void) { CAPI_HIP(hipDeviceSynchronize()); return OCR_OK; }

}  // extern "C"
