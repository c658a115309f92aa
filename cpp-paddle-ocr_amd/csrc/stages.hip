#include "stages.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>

#include "capi_common.h"

namespace ocr {

#define ST_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      err = std::string(#expr) + ": " + hipGetErrorString(_e);                    \
      return OCR_ERR_DEVICE;                                                      \
    }                                                                             \
  } while (0)

bool StageTimer::init(std::string& err) {
  for (int i = 0; i < 4; ++i)
    if (hipEventCreate(&ev[i]) != hipSuccess) { err = "hipEventCreate failed"; return false; }
  return true;
}
StageTimer::~StageTimer() {
  for (int i = 0; i < 4; ++i)
    if (ev[i]) (void)hipEventDestroy(ev[i]);
}
void StageTimer::read(double times[3]) {
  if (!times) return;
  for (int i = 0; i < 3; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[i], ev[i + 1]) != hipSuccess) ms = 0.f;
    times[i] = ms;
  }
}

std::vector<float> make_norm_lut(const float mean[3], const float scale[3]) {
  // Normalize::Run (/root/reference/src/preprocess_op.cpp:40-55): convertTo(CV_32FC3, 1/255), then per
  // channel convertTo(alpha = scale, beta = -mean*scale); both are x*a+b in float.
  std::vector<float> lut(3 * 256);
  const float e = (float)(1.0 / 255.0);
  for (int c = 0; c < 3; ++c) {
    const float a = (float)(1.0 * (double)scale[c]);
    const float b = (float)((0.0 - (double)mean[c]) * (double)scale[c]);
    for (int v = 0; v < 256; ++v) lut[c * 256 + v] = fmaf((float)v * e, a, b);
  }
  return lut;
}

bool upload_lines(const ocr_img* imgs, int n, DevBuf<uint8_t>& staging, std::vector<LineSrc>& out, hipStream_t s,
                  std::string& err) {
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    if (!imgs[i].data || imgs[i].rows <= 0 || imgs[i].cols <= 0) { err = "empty text-line image"; return false; }
    total += ((size_t)imgs[i].rows * imgs[i].cols * 3 + 15) & ~(size_t)15;
  }
  if (!staging.ensure(total, err)) return false;
  size_t off = 0;
  out.resize(n);
  for (int i = 0; i < n; ++i) {
    const size_t row = (size_t)imgs[i].cols * 3;
    const size_t stride = imgs[i].row_stride ? imgs[i].row_stride : row;
    if (hipMemcpy2DAsync(staging.p + off, row, imgs[i].data, stride, row, imgs[i].rows, hipMemcpyHostToDevice, s) !=
        hipSuccess) { err = "hipMemcpy2DAsync failed"; return false; }
    out[i] = LineSrc{staging.p + off, row, 0, 0, imgs[i].cols, imgs[i].rows};
    off += ((size_t)imgs[i].rows * row + 15) & ~(size_t)15;
  }
  return true;
}

// ================================================================= detector
DetStage::~DetStage() {
  if (mixed_done_) (void)hipEventDestroy(mixed_done_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void DetStage::resize_shape(int h, int w, const std::string& limit_type, int limit_side_len, int& rh, int& rw,
                            float& ratio_h, float& ratio_w) {
  // ResizeImgType0::Run, /root/reference/src/preprocess_op.cpp:57-93
  float ratio = 1.f;
  if (limit_type == "min") {
    const int min_wh = std::min(h, w);
    if (min_wh < limit_side_len) ratio = h < w ? float(limit_side_len) / float(h) : float(limit_side_len) / float(w);
  } else {
    const int max_wh = std::max(h, w);
    if (max_wh > limit_side_len) ratio = h > w ? float(limit_side_len) / float(h) : float(limit_side_len) / float(w);
  }
  int resize_h = int(float(h) * ratio);
  int resize_w = int(float(w) * ratio);
  resize_h = std::max(int(round(float(resize_h) / 32) * 32), 32);
  resize_w = std::max(int(round(float(resize_w) / 32) * 32), 32);
  rh = resize_h;
  rw = resize_w;
  ratio_h = float(resize_h) / float(h);
  ratio_w = float(resize_w) / float(w);
}

bool DetStage::create(const DetConfig& cfg, std::string& err, int& code) {
  cfg_ = cfg;
  code = OCR_ERR_ARG;
  // the reference's constructor parameter (TensorRT precision, ocr_det.cpp:50-56 / ocr_cls.cpp:135-140 / ocr_rec.cpp:167-172):
  // "fp32" = the bit-exact contract, "fp16" = f16 matrix products with f32 accumulation (Net::load); "int8" is refused
  if (cfg.precision != "fp32" && cfg.precision != "fp16") { err = "precision '" + cfg.precision + "' is not implemented (fp32 | fp16)"; return false; }
  if (cfg.score_mode != "fast" && cfg.score_mode != "slow") { err = "det_db_score_mode must be fast or slow"; return false; }
  if (cfg.limit_type != "max" && cfg.limit_type != "min") { err = "limit_type must be max or min"; return false; }
  if (cfg.max_batch < 1) { err = "max_batch must be >= 1"; return false; }
  if (cfg.cv_compat != OCR_CV_45 && cfg.cv_compat != OCR_CV_410) { err = "cv_compat must be OCR_CV_45 or OCR_CV_410"; return false; }
  code = ocr_rt_init(cfg.device);
  if (code) { err = ocr_last_error(); return false; }
  WeightMap w;
  code = OCR_ERR_MODEL;
  if (server_arch(cfg.model_dir) == "srv_det") {
    // BASELINE configs[4]: the server detector (ResNet50-vd DB; hand-written plan, NOT a reference artifact) on the f16
    // implicit-GEMM family of srv_kernels.hip; "fp32" is its parity twin
    if (!load_server_model_dir(cfg.model_dir, "srv_det", w, err)) return false;
    srv_.reset(new SrvNet());
    if (!srv_->load(embedded_plan("srv_det"), w, cfg.precision == "fp16", err)) return false;
  } else {
    if (!load_model_dir(cfg.model_dir, nullptr, "det", w, err)) return false;
    if (!net_.load(embedded_plan("det"), w, err, cfg.precision == "fp16")) return false;
  }
  code = OCR_ERR_DEVICE;
  if (g_stream_create(&stream_) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
  if (!timer_.init(err)) return false;
  const float mean[3] = {0.485f, 0.456f, 0.406f};                      // ocr_det.h:121
  const float scale[3] = {1 / 0.229f, 1 / 0.224f, 1 / 0.225f};        // ocr_det.h:122
  const auto lut = make_norm_lut(mean, scale);
  if (!lut_.ensure(lut.size(), err)) return false;
  if (g_memcpy(lut_.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { err = "lut upload failed"; return false; }
  ithresh_ = (int)std::floor(cfg.thresh * 255);  // cv::threshold on 8U floors the threshold (ocr_det.cpp:151-154)
  code = OCR_OK;
  return true;
}

bool DetStage::ensure_post(int count, size_t px, int H, int W, int cap, std::string& err) {
  const int max_cand = 1000;
  // per image: a key pool for the traced borders and (slow score mode) a word pool for the polygon masks; sized for
  // ordinary maps, grown by post_launch when a map needs more (never kept: every call starts from the default sizing)
  pool_cap_ = std::max(1 << 16, (H * W) / 2);
  mask_words_ = mask_words(H, W);
  return labels_.ensure(px, err) && touch_.ensure(px, err) && chunk_cnt_.ensure((size_t)count * 128, err) && ncont_all_.ensure(count, err) && ncont_.ensure(count, err) &&
         starts_.ensure((size_t)count * max_cand, err) && npts_.ensure((size_t)count * max_cand, err) &&
         poff_.ensure((size_t)count * max_cand, err) && pool_.ensure((size_t)count * pool_cap_, err) &&
         iscratch_.ensure((size_t)count * pool_cap_ * 4, err) && cand_boxes_.ensure((size_t)count * max_cand * 8, err) &&
         cand_valid_.ensure((size_t)count * max_cand, err) && status_.ensure(1, err) && pool_need_.ensure(count, err) &&
         out_boxes_.ensure((size_t)count * cap * 8, err) && out_n_.ensure(count, err) &&
         (cfg_.use_dilation ? bitmap2_.ensure(px, err) : true) &&
         (cfg_.score_mode == "slow" ? (mask_pool_.ensure((size_t)count * mask_words_, err) && mask_top_.ensure(count, err)) : true);
}

// Runs the post-processing kernels for the `count` maps `a` describes and brings boxes and counts back.  The reference
// answers every map (findContours / minAreaRect / fillPoly allocate what they need, postprocess_op.cpp:255-331), so
// exhausted scratch is not an error here either: a border that outgrows the small LDS working set of the per-border
// stage re-runs that stage with the large one, and a map whose first max_candidates borders hold more vertices than the
// default key pool (or whose polygon masks need more words than the default mask pool) re-runs the pass with pools of
// the size the device reported (pool_need / the mask pool's bump cursor count what WOULD have been needed).
int DetStage::post_launch(PostArgs a, int count, int32_t* boxes, int cap, int* n, std::string& err) {
  a.labels = labels_.p; a.touch = touch_.p; a.chunk_cnt = chunk_cnt_.p; a.ncont_all = ncont_all_.p; a.ncont = ncont_.p;
  a.starts = starts_.p; a.npts = npts_.p; a.poff = poff_.p; a.cand_boxes = cand_boxes_.p; a.cand_valid = cand_valid_.p;
  a.status = status_.p; a.pool_need = pool_need_.p; a.max_cand = 1000;
  a.box_thresh = (float)cfg_.box_thresh; a.unclip_ratio = (float)cfg_.unclip_ratio;
  a.slow = cfg_.score_mode == "slow";
  a.fill_shifted = cfg_.cv_compat != OCR_CV_45;
  a.probe_stop = 0;
#ifdef OCR_DEV_PROBES  // development probe (tools/post_kernel_times.sh builds with -DOCR_DEV_PROBES): timing only, results are wrong
  { static const char* stop = getenv("OCR_POST_STOP"); a.probe_stop = stop ? atoi(stop) : 0; }
#endif
  int status = 0;
  auto fetch = [&]() -> int {
    ST_HIP(hipMemcpyAsync(n, out_n_.p, count * sizeof(int), hipMemcpyDeviceToHost, stream_));
    ST_HIP(hipMemcpyAsync(boxes, out_boxes_.p, (size_t)count * cap * 8 * sizeof(int), hipMemcpyDeviceToHost, stream_));
    ST_HIP(hipMemcpyAsync(&status, status_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    return OCR_OK;
  };
  for (int attempt = 0;; ++attempt) {
    a.pool = pool_.p; a.iscratch = iscratch_.p; a.pool_cap = pool_cap_;
    if (a.slow) {
      a.mask_pool = mask_pool_.p; a.mask_pool_top = mask_top_.p; a.mask_pool_words = (unsigned)mask_words_;
      ST_HIP(hipMemsetAsync(mask_top_.p, 0, count * sizeof(unsigned), stream_));
    }
    ST_HIP(hipMemsetAsync(status_.p, 0, sizeof(int), stream_));
    launch_post(a, count, out_boxes_.p, cap, out_n_.p, stream_);
    ST_HIP(hipGetLastError());  // a refused launch (e.g. LDS limit on this device) must not leave stale borders behind
    if (int rc = fetch()) return rc;
    if (attempt == 0) timer_.mark(3, stream_);
    ST_HIP(g_stream_sync(stream_));
    if (status && !(status & ~(POST_ERR_HULL | POST_ERR_UNCLIP))) {
      // a border outgrew the small LDS working set of the per-border stage: run that stage again with the large one
      ST_HIP(hipMemsetAsync(status_.p, 0, sizeof(int), stream_));
      if (a.slow) ST_HIP(hipMemsetAsync(mask_top_.p, 0, count * sizeof(unsigned), stream_));
      launch_post_large(a, count, out_boxes_.p, cap, out_n_.p, stream_);
      ST_HIP(hipGetLastError());
      if (int rc = fetch()) return rc;
      ST_HIP(g_stream_sync(stream_));
    }
    if (!(status & POST_ERR_POOL) || attempt == 4) break;
    // what the maps would have needed (largest image decides: the pools are [count][cap])
    std::vector<int> need(count);
    std::vector<unsigned> mneed(a.slow ? count : 0);
    ST_HIP(hipMemcpyAsync(need.data(), pool_need_.p, count * sizeof(int), hipMemcpyDeviceToHost, stream_));
    if (a.slow) ST_HIP(hipMemcpyAsync(mneed.data(), mask_top_.p, count * sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    ST_HIP(g_stream_sync(stream_));
    const long want_keys = *std::max_element(need.begin(), need.end());
    const size_t want_words = a.slow ? (size_t)*std::max_element(mneed.begin(), mneed.end()) : 0;
    bool grew = false;
    if (want_keys > pool_cap_) {
      if (want_keys > (1L << 28)) break;  // 2 GB of keys per image: not a map this library is sized for
      pool_cap_ = (int)((want_keys + 65535) & ~65535L);
      if (!pool_.ensure((size_t)count * pool_cap_, err) || !iscratch_.ensure((size_t)count * pool_cap_ * 4, err)) return OCR_ERR_DEVICE;
      grew = true;
    }
    if (want_words > mask_words_) {
      if (want_words >= (1ul << 31)) break;
      mask_words_ = (want_words + 65535) & ~(size_t)65535;
      if (!mask_pool_.ensure((size_t)count * mask_words_, err)) return OCR_ERR_DEVICE;
      grew = true;
    }
    if (!grew) break;
  }
  if (status) {
    err = "det post-processing scratch exhausted (status " + std::to_string(status) + ")";
    return OCR_ERR_CAPACITY;
  }
  for (int i = 0; i < count; ++i)
    if (n[i] > cap) { err = "more boxes than the caller's capacity"; return OCR_ERR_CAPACITY; }
  return OCR_OK;
}

int DetStage::run_post(int count, int H, int W, const float* prob, float ratio_h, float ratio_w, int src_h, int src_w,
                       int32_t* boxes, int cap, int* n, std::string& err, const uint8_t* bitmap) {
  if (!ensure_post(count, (size_t)count * H * W, H, W, cap, err)) return OCR_ERR_DEVICE;
  const uint8_t* bm = bitmap ? bitmap : bitmap_.p;
  if (cfg_.use_dilation) {
    launch_dilate2(bm, bitmap2_.p, count, H, W, stream_);
    bm = bitmap2_.p;
  }
  PostArgs a{};
  a.bitmap = bm; a.pred = prob;
  a.H = H; a.W = W;
  a.ratio_h = ratio_h; a.ratio_w = ratio_w; a.src_h = src_h; a.src_w = src_w;
  return post_launch(a, count, boxes, cap, n, err);
}

int DetStage::run(const ocr_img* imgs, int count, int32_t* boxes, int cap, int* n, double times[3], std::string& err) {
  if (!imgs || count < 1 || !boxes || !n || cap < 1) { err = "bad argument"; return OCR_ERR_ARG; }
  const int rows = imgs[0].rows, cols = imgs[0].cols;
  for (int i = 0; i < count; ++i) {
    if (!imgs[i].data || imgs[i].rows <= 0 || imgs[i].cols <= 0) { err = "Empty image data provided"; return OCR_ERR_ARG; }
    if (imgs[i].rows != rows || imgs[i].cols != cols) { err = "ocr_det_run_batch needs images of one size"; return OCR_ERR_ARG; }
  }
  ST_HIP(rt_set_device(cfg_.device));
  const size_t row = (size_t)cols * 3, img_bytes = row * rows;
  if (!src_.ensure(img_bytes * count, err)) return OCR_ERR_DEVICE;
  timer_.mark(0, stream_);
  for (int i = 0; i < count; ++i) {
    const size_t stride = imgs[i].row_stride ? imgs[i].row_stride : row;
    ST_HIP(hipMemcpy2DAsync(src_.p + img_bytes * i, row, imgs[i].data, stride, row, rows, hipMemcpyHostToDevice, stream_));
  }
  src_rows_ = rows;
  src_cols_ = cols;
  return run_device(src_.p, img_bytes, row, rows, cols, count, boxes, cap, n, times, err);
}

int DetStage::run_device(const uint8_t* dev_imgs, size_t img_bytes, size_t stride, int rows, int cols, int count,
                         int32_t* boxes, int cap, int* n, double times[3], std::string& err, const float* prob_override) {
  ST_HIP(rt_set_device(cfg_.device));
  int rh, rw;
  float ratio_h, ratio_w;
  resize_shape(rows, cols, cfg_.limit_type, cfg_.limit_side_len, rh, rw, ratio_h, ratio_w);
  const size_t px = (size_t)count * rh * rw;
  const uint8_t* old_bm = bitmap_.p;
  if (!x_.ensure(px * 3, err) || !resized_.ensure(px * 3, err) || !bitmap_.ensure(px, err)) return OCR_ERR_DEVICE;
  if (dev_imgs != src_.p) timer_.mark(0, stream_);
  DetPreArgs pa{};
  pa.src = dev_imgs; pa.src_image_bytes = img_bytes; pa.src_stride = stride;
  pa.N = count; pa.sh = rows; pa.sw = cols; pa.dh = rh; pa.dw = rw; pa.lut = lut_.p; pa.out = x_.p; pa.resized = resized_.p;
  launch_det_pre(pa, stream_);
  timer_.mark(1, stream_);
  if (srv_) {
    if (!srv_->run(x_.p, count, rh, rw, stream_, err)) return OCR_ERR_DEVICE;
    if (!prob_override) launch_bitmap(prob_dev(), bitmap_.p, (long)px, ithresh_, stream_);  // (the mobile network's head writes it itself)
  } else {
    if (bitmap_.p != old_bm || bm_n_ == 0) { net_.set_det_bitmap(bitmap_.p, ithresh_); bm_n_ = 1; }
    if (!net_.run(x_.p, count, rh, rw, stream_, err)) return OCR_ERR_DEVICE;
  }
  timer_.mark(2, stream_);
  last_count = count; last_h = rh; last_w = rw;
  const float* pred = prob_dev();
  if (prob_override) {
    launch_bitmap(prob_override, bitmap_.p, (long)px, ithresh_, stream_);
    pred = prob_override;
  }
  src_rows_ = rows;
  src_cols_ = cols;
  const int rc = run_post(count, rh, rw, pred, ratio_h, ratio_w, rows, cols, boxes, cap, n, err);
  net_.collect_timings();
  if (srv_) srv_->collect_timings();
  timer_.read(times);
  return rc;
}

int DetStage::mixed_net(const uint8_t* base, const MixedGroup* groups, int ngroups, const float* prob_override, std::string& err) {
  ST_HIP(rt_set_device(cfg_.device));
  if (srv_) { err = "the server detector runs batches of one image size (a ragged batch of mixed sizes is not built for it)"; return OCR_ERR_ARG; }
  if (!mixed_done_) ST_HIP(hipEventCreateWithFlags(&mixed_done_, hipEventDisableTiming));
  mixed_pix_.assign(ngroups + 1, 0);
  std::vector<int> hs, ws;
  std::vector<std::array<int, 2>> shape(ngroups);
  for (int gi = 0; gi < ngroups; ++gi) {
    float a, b;
    resize_shape(groups[gi].rows, groups[gi].cols, cfg_.limit_type, cfg_.limit_side_len, shape[gi][0], shape[gi][1], a, b);
    mixed_pix_[gi + 1] = mixed_pix_[gi] + (size_t)groups[gi].count * shape[gi][0] * shape[gi][1];
    for (int k = 0; k < groups[gi].count; ++k) { hs.push_back(shape[gi][0]); ws.push_back(shape[gi][1]); }
  }
  const size_t px = mixed_pix_[ngroups];
  const uint8_t* old_bm = bitmap_.p;
  if (!x_.ensure(px * 3, err) || !resized_.ensure(px * 3, err) || !bitmap_.ensure(px, err)) return OCR_ERR_DEVICE;
  timer_.mark(0, stream_);
  for (int gi = 0; gi < ngroups; ++gi) {  // ResizeImgType0 + Normalize + Permute per size group, into the ragged input
    const MixedGroup& g = groups[gi];
    DetPreArgs pa{};
    pa.src = base + g.off; pa.src_image_bytes = (size_t)g.rows * g.cols * 3; pa.src_stride = (size_t)g.cols * 3;
    pa.N = g.count; pa.sh = g.rows; pa.sw = g.cols; pa.dh = shape[gi][0]; pa.dw = shape[gi][1]; pa.lut = lut_.p;
    pa.out = x_.p + mixed_pix_[gi] * 3; pa.resized = resized_.p + mixed_pix_[gi] * 3;
    launch_det_pre(pa, stream_);
  }
  timer_.mark(1, stream_);
  if (bitmap_.p != old_bm || bm_n_ == 0) { net_.set_det_bitmap(bitmap_.p, ithresh_); bm_n_ = 1; }
  if (!net_.run_ragged_images(x_.p, hs.data(), ws.data(), (int)hs.size(), stream_, err)) return OCR_ERR_DEVICE;
  timer_.mark(2, stream_);
  mixed_probs_.assign(ngroups, nullptr);
  mixed_prob_base_ = prob_override ? prob_override : prob_dev();
  for (int gi = 0; gi < ngroups; ++gi) {
    mixed_probs_[gi] = prob_dev() + mixed_pix_[gi];
    if (prob_override) {  // the benchmark protocol: thresholding and scoring read these maps (group gi at its prob_off)
      mixed_probs_[gi] = prob_override + groups[gi].prob_off;
      launch_bitmap(mixed_probs_[gi], bitmap_.p + mixed_pix_[gi], (long)(mixed_pix_[gi + 1] - mixed_pix_[gi]), ithresh_, stream_);
    }
  }
  ST_HIP(hipEventRecord(mixed_done_, stream_));
  last_count = (int)hs.size(); last_h = 0; last_w = 0;
  return OCR_OK;
}

int DetStage::post_mixed(const MixedGroup* groups, int ngroups, int32_t* boxes, int cap, int* n, std::string& err) {
  ST_HIP(rt_set_device(cfg_.device));
  if (cfg_.use_dilation) { err = "post_mixed: not with use_dilation"; return OCR_ERR_ARG; }
  std::vector<PostImg> im;
  int Hm = 0, Wm = 0;
  for (int gi = 0; gi < ngroups; ++gi) {
    int rh, rw;
    float ratio_h, ratio_w;
    resize_shape(groups[gi].rows, groups[gi].cols, cfg_.limit_type, cfg_.limit_side_len, rh, rw, ratio_h, ratio_w);
    Hm = std::max(Hm, rh); Wm = std::max(Wm, rw);
    for (int k = 0; k < groups[gi].count; ++k) {
      PostImg q;
      q.h = rh; q.w = rw;
      // 32-bit offsets on the device: the END of every image's maps must stay below 2^31 (the probability maps may sit
      // far into a staging slot when prob_override is used), checked before anything is narrowed
      const size_t off = mixed_pix_[gi] + (size_t)k * rh * rw;
      const size_t poff = (size_t)(mixed_probs_[gi] - mixed_prob_base_) + (size_t)k * rh * rw;
      if (mixed_probs_[gi] < mixed_prob_base_ || off + (size_t)rh * rw >= (1ul << 31) || poff + (size_t)rh * rw >= (1ul << 31)) {
        err = "post_mixed: maps too large for 32-bit offsets";
        return OCR_ERR_CAPACITY;
      }
      q.off = (int)off;
      q.poff = (int)poff;
      q.src_h = groups[gi].rows; q.src_w = groups[gi].cols; q.ratio_h = ratio_h; q.ratio_w = ratio_w;
      im.push_back(q);
    }
  }
  const int count = (int)im.size();
  // scratch: per pixel for the whole chunk, per image sized for the largest map
  if (!ensure_post(count, mixed_pix_[ngroups], Hm, Wm, cap, err) || !post_img_.ensure(count, err)) return OCR_ERR_DEVICE;
  ST_HIP(hipMemcpyAsync(post_img_.p, im.data(), (size_t)count * sizeof(PostImg), hipMemcpyHostToDevice, stream_));
  PostArgs a{};
  a.img = post_img_.p;
  a.bitmap = bitmap_.p; a.pred = mixed_prob_base_;
  a.H = Hm; a.W = Wm;
  return post_launch(a, count, boxes, cap, n, err);
}

int DetStage::post_group(const float* prob, const uint8_t* bitmap, const MixedGroup& g, hipEvent_t wait_for, int32_t* boxes, int cap,
                         int* n, std::string& err) {
  ST_HIP(rt_set_device(cfg_.device));
  if (wait_for) ST_HIP(hipStreamWaitEvent(stream_, wait_for, 0));
  int rh, rw;
  float ratio_h, ratio_w;
  resize_shape(g.rows, g.cols, cfg_.limit_type, cfg_.limit_side_len, rh, rw, ratio_h, ratio_w);
  src_rows_ = g.rows;
  src_cols_ = g.cols;
  return run_post(g.count, rh, rw, prob, ratio_h, ratio_w, g.rows, g.cols, boxes, cap, n, err, bitmap);
}

int DetStage::post_only(const float* prob, int rows, int cols, int src_rows, int src_cols, int32_t* boxes, int cap, int* n,
                        std::string& err) {
  if (!prob || rows <= 0 || cols <= 0 || !boxes || !n || cap < 1) { err = "bad argument"; return OCR_ERR_ARG; }
  ST_HIP(rt_set_device(cfg_.device));
  const size_t px = (size_t)rows * cols;
  if (!prob_in_.ensure(px, err) || !bitmap_.ensure(px, err)) return OCR_ERR_DEVICE;
  bm_n_ = 0;  // the fused-bitmap pointer may have moved: re-arm on the next network run
  ST_HIP(hipMemcpyAsync(prob_in_.p, prob, px * sizeof(float), hipMemcpyHostToDevice, stream_));
  launch_bitmap(prob_in_.p, bitmap_.p, (long)px, ithresh_, stream_);
  last_count = 1; last_h = rows; last_w = cols;
  const float ratio_h = float(rows) / float(src_rows), ratio_w = float(cols) / float(src_cols);
  return run_post(1, rows, cols, prob_in_.p, ratio_h, ratio_w, src_rows, src_cols, boxes, cap, n, err);
}

// ================================================================= recognizer
RecStage::~RecStage() {
  if (stream_) (void)hipStreamDestroy(stream_);
}

bool RecStage::create(const RecConfig& cfg, std::string& err, int& code) {
  cfg_ = cfg;
  code = OCR_ERR_ARG;
  // the reference's constructor parameter (TensorRT precision, ocr_det.cpp:50-56 / ocr_cls.cpp:135-140 / ocr_rec.cpp:167-172):
  // "fp32" = the bit-exact contract, "fp16" = f16 matrix products with f32 accumulation (Net::load); "int8" is refused
  if (cfg.precision != "fp32" && cfg.precision != "fp16") { err = "precision '" + cfg.precision + "' is not implemented (fp32 | fp16)"; return false; }
  if (cfg.batch_num < 1 || cfg.img_h < 1 || cfg.img_w < 1) { err = "bad rec shape"; return false; }
  if (cfg.sort_mode != OCR_SORT_STD && cfg.sort_mode != OCR_SORT_STABLE && cfg.sort_mode != OCR_SORT_MSVC_STRICT) { err = "unknown sort_mode"; return false; }
  code = ocr_rt_init(cfg.device);
  if (code) { err = ocr_last_error(); return false; }
  code = OCR_ERR_MODEL;
  // Utility::ReadDict + "#" / " " (ocr_rec.h:82-84)
  std::ifstream in(cfg.label_path);
  if (!in) { err = "no such label file: " + cfg.label_path; return false; }
  labels_.clear();
  labels_.push_back("#");
  for (std::string line; std::getline(in, line);) labels_.push_back(line);
  labels_.push_back(" ");
  WeightMap w;
  if (server_arch(cfg.model_dir) == "srv_rec") {
    // BASELINE configs[4]: SVTR-large (hand-written plan, NOT a reference artifact).  Its position embedding and local-mixing
    // windows belong to ONE token grid: every line is resized into rec_img_h x rec_img_w = 48 x 320 (CrnnResizeImg with the
    // batch's width ratio held at rec_img_w / rec_img_h instead of growing with the widest line, ocr_rec.cpp:47-57)
    if (cfg.img_h != 48 || cfg.img_w != 320) { err = "the server recognizer takes 48 x 320 lines (rec_img_h, rec_img_w)"; code = OCR_ERR_ARG; return false; }
    if (!load_server_model_dir(cfg.model_dir, "srv_rec", w, err)) return false;
    srv_.reset(new SrvNet());
    if (!srv_->load(embedded_plan("srv_rec"), w, cfg.precision == "fp16", err)) return false;
    {  // the pipeline needs arg max and its probability, never the logits: the CTC head in partial mode (f16 build; OCR_SRV_CTC=0: logits + one pass over them)
      static const bool ctc_on = [] { const char* e = getenv("OCR_SRV_CTC"); return !(e && e[0] == '0'); }();
      srv_->set_ctc_partials(ctc_on);
    }
  } else {
    if (!load_model_dir(cfg.model_dir, nullptr, "rec", w, err)) return false;
    if (!net_.load(embedded_plan("rec"), w, err, cfg.precision == "fp16")) return false;
  }
  code = OCR_ERR_DEVICE;
  if (g_stream_create(&stream_) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
  if (!timer_.init(err)) return false;
  const float mean[3] = {0.5f, 0.5f, 0.5f}, scale[3] = {1 / 0.5f, 1 / 0.5f, 1 / 0.5f};  // ocr_rec.h:108-109
  const auto lut = make_norm_lut(mean, scale);
  if (!lut_.ensure(lut.size(), err)) return false;
  if (g_memcpy(lut_.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { err = "lut upload failed"; return false; }
  code = OCR_OK;
  return true;
}

int RecStage::run(const ocr_img* imgs, int n, int32_t* ids, int max_len, int* lens, float* scores, double times[3],
                  std::string& err) {
  if (n < 0 || (n > 0 && (!imgs || !ids || !lens || !scores)) || max_len < 1) { err = "bad argument"; return OCR_ERR_ARG; }
  if (times) times[0] = times[1] = times[2] = 0;
  if (n == 0) return OCR_OK;
  ST_HIP(rt_set_device(cfg_.device));
  timer_.mark(0, stream_);
  std::vector<LineSrc> lines;
  if (!upload_lines(imgs, n, staging_, lines, stream_, err)) return OCR_ERR_DEVICE;
  timer_.mark(1, stream_);
  const std::vector<int> seg = {0, n};
  const int rc = run_lines(lines, seg, ids, max_len, lens, scores, err);
  timer_.mark(2, stream_);
  timer_.mark(3, stream_);
  (void)g_stream_sync(stream_);
  timer_.read(times);
  return rc;
}

int RecStage::run_lines(const std::vector<LineSrc>& lines, const std::vector<int>& seg, int32_t* ids, int max_len,
                        int* lens, float* scores, std::string& err) {
  const int n = (int)lines.size();
  const int imgH = cfg_.img_h, imgW = cfg_.img_w;
  struct Item { int line, resize_w, tensor_w; };
  std::vector<Item> items;
  items.reserve(n);
  for (size_t sg = 0; sg + 1 < seg.size(); ++sg) {
    // CRNNRecognizer::Run batching, /root/reference/src/ocr_rec.cpp:34-57, on one image's lines
    const int s0 = seg[sg], sn = seg[sg + 1] - seg[sg];
    std::vector<float> width_list(sn);
    for (int i = 0; i < sn; ++i) width_list[i] = float(lines[s0 + i].w) / lines[s0 + i].h;
    std::vector<size_t> indices(sn);
    for (int i = 0; i < sn; ++i) indices[i] = i;
    // Utility::argsort is std::sort: the order of equal ratios is the host library's
    if (cfg_.sort_mode == OCR_SORT_MSVC_STRICT && sn > 32) {
      // MSVC's std::sort is an insertion sort (ties in input order) only up to _ISORT_MAX = 32 elements; its quicksort's order
      // of ties beyond that is not restated here - refuse rather than answer in an order the reference's build may not produce
      std::vector<float> sorted_w(width_list);
      std::sort(sorted_w.begin(), sorted_w.end());
      if (std::adjacent_find(sorted_w.begin(), sorted_w.end()) != sorted_w.end()) {
        char msg[200];
        snprintf(msg, sizeof msg, "sort_mode OCR_SORT_MSVC_STRICT: image %d has %d crops (> 32) with tied w/h ratios - MSVC's std::sort order of ties beyond 32 elements is not restated (use OCR_SORT_STABLE or OCR_SORT_STD)", (int)sg, sn);
        err = msg;
        return OCR_ERR_ARG;
      }
    }
    if (cfg_.sort_mode != OCR_SORT_STD) std::stable_sort(indices.begin(), indices.end(), [&](size_t a, size_t b) { return width_list[a] < width_list[b]; });
    else std::sort(indices.begin(), indices.end(), [&](size_t a, size_t b) { return width_list[a] < width_list[b]; });
    for (int beg = 0; beg < sn; beg += cfg_.batch_num) {
      const int end = std::min(sn, beg + cfg_.batch_num);
      float max_wh_ratio = imgW * 1.0 / imgH;
      for (int ino = beg; ino < end; ++ino) {
        const int h = lines[s0 + indices[ino]].h, w = lines[s0 + indices[ino]].w;
        const float wh_ratio = w * 1.0 / h;
        max_wh_ratio = std::max(max_wh_ratio, wh_ratio);
      }
      if (srv_) max_wh_ratio = imgW * 1.0 / imgH;  // (the server recognizer's fixed token grid: create())
      const int bw = int(imgH * max_wh_ratio);  // CrnnResizeImg: imgW = int(imgH * wh_ratio)
      const int tensor_w = std::max(bw, imgW);
      for (int ino = beg; ino < end; ++ino) {
        const LineSrc& L = lines[s0 + indices[ino]];
        const float ratio = float(L.w) / float(L.h);
        const int resize_w = ceilf(imgH * ratio) > bw ? bw : int(ceilf(imgH * ratio));
        items.push_back({s0 + (int)indices[ino], resize_w, tensor_w});
      }
    }
  }
  tap_T.assign(n, 0);
  tap_off.assign(n, 0);
  tap_amax.clear();
  tap_pmax.clear();
  for (int i = 0; i < n; ++i) { lens[i] = 0; scores[i] = 0.f; }
  if (items.empty()) return OCR_OK;
  // One launch list for every tensor width of the call (a ragged batch: kernels_net.h, RagLevel).  In the reference each
  // batch of rec_batch_num lines is a predictor run of its own width (ocr_rec.cpp:47-81); a line's results depend on
  // its own tensor width only (padding, attention length), never on the other lines of a launch, so the lines of all
  // batches - and of all images of a request batch - share the launches.  Lines in ascending tensor width (equal
  // widths stay neighbours: a workgroup's tiles then belong to lines of one shape); a launch is cut where its
  // activation arena would pass the pixel budget; a line too wide for the ragged attention kernel's LDS working set
  // (> ~3000 px) runs as a uniform launch of its own width.
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.tensor_w < b.tensor_w; });
  struct Slot { int first, count; bool ragged; int T; size_t step_off; };
  std::vector<Slot> slots;
  const long budget = (long)max_lines_per_launch * imgH * std::max(imgW, 320);
  {
    int i = 0;
    const int ni = (int)items.size();
    while (i < ni) {
      if (srv_ || !attn_ragged_fits(items[i].tensor_w / 8 + 2)) {  // uniform launch: the lines of exactly this width
        int j = i;
        while (j < ni && items[j].tensor_w == items[i].tensor_w && j - i < max_lines_per_launch) ++j;
        slots.push_back({i, j - i, false, 0, 0});
        i = j;
        continue;
      }
      long pix = 0;
      int j = i;
      while (j < ni && attn_ragged_fits(items[j].tensor_w / 8 + 2) && (j == i || pix + (long)imgH * items[j].tensor_w <= budget)) {
        pix += (long)imgH * items[j].tensor_w;
        ++j;
      }
      slots.push_back({i, j - i, true, 0, 0});
      i = j;
    }
  }
  // per launch: bind (host only), so that the step counts are known before the line descriptors travel
  const int ni = (int)items.size();
  std::vector<LineDesc> d(ni);
  std::vector<int> widths;
  size_t step_total = 0, xneed = 0;
  for (Slot& sl : slots) {
    sl.step_off = step_total;
    if (sl.ragged) {
      widths.resize(sl.count);
      for (int j = 0; j < sl.count; ++j) widths[j] = items[sl.first + j].tensor_w;
      if (!net_.bind_ragged(imgH, widths.data(), sl.count, err)) return OCR_ERR_DEVICE;
      const TensorDesc& ot = net_.tensor(net_.output_tid());
      if (ot.h != 1) { err = "rec_img_h does not reduce to a single row"; return OCR_ERR_ARG; }
      if (ot.c != (int)labels_.size()) { err = "dictionary size does not match the CTC head"; return OCR_ERR_MODEL; }
      const std::vector<int>& T = net_.ragged_widths(net_.output_tid());
      long pix = 0;
      for (int j = 0; j < sl.count; ++j) {
        const Item& it = items[sl.first + j];
        const LineSrc& L = lines[it.line];
        LineDesc& q = d[sl.first + j];
        q = LineDesc::make(L.img, L.stride, L.x, L.y, L.w, L.h, it.resize_w, j, imgH);
        q.tensor_w = it.tensor_w;
        q.pix0 = (int)pix;
        q.step0 = (int)(step_total - sl.step_off);
        q.steps = T[j];
        pix += (long)imgH * it.tensor_w;
        step_total += T[j];
      }
      xneed = std::max(xneed, (size_t)pix * 3);
    } else {
      for (int j = 0; j < sl.count; ++j) {
        const Item& it = items[sl.first + j];
        const LineSrc& L = lines[it.line];
        d[sl.first + j] = LineDesc::make(L.img, L.stride, L.x, L.y, L.w, L.h, it.resize_w, j, imgH);
      }
      step_total += (size_t)sl.count * (items[sl.first].tensor_w / 4 + 8);  // >= count * T
      xneed = std::max(xneed, (size_t)sl.count * imgH * items[sl.first].tensor_w * 3);
    }
  }
  if (!amax_.ensure(step_total, err) || !pmax_.ensure(step_total, err) || !ids_.ensure((size_t)ni * max_len, err) ||
      !lens_.ensure(ni, err) || !scores_.ensure(ni, err) || !descs_.ensure(ni, err) || !x_.ensure(xneed, err))
    return OCR_ERR_DEVICE;
  ST_HIP(hipMemcpyAsync(descs_.p, d.data(), (size_t)ni * sizeof(LineDesc), hipMemcpyHostToDevice, stream_));
  for (Slot& sl : slots) {
    if (!srv_) net_.set_head_outputs(nullptr, amax_.p + sl.step_off, pmax_.p + sl.step_off);
    if (srv_) {
      const int Wt = items[sl.first].tensor_w;
      launch_line_pre(descs_.p + sl.first, sl.count, imgH, Wt, lut_.p, false, x_.p, stream_);
      if (!srv_->run(x_.p, sl.count, imgH, Wt, stream_, err)) return OCR_ERR_DEVICE;
      const SrvTensor& ot = srv_->tensor(srv_->output_tid());
      if (ot.h != 1) { err = "rec_img_h does not reduce to a single row"; return OCR_ERR_ARG; }
      if (ot.c != (int)labels_.size()) { err = "dictionary size does not match the CTC head"; return OCR_ERR_MODEL; }
      sl.T = ot.w;
      if (sl.T > Wt / 4 + 8) { err = "rec step buffer too small"; return OCR_ERR_CAPACITY; }
      if (srv_->ctc_partials())  // the head left per-row partials instead of logits (srv_net.h): fold them
        srv::launch_ctc_reduce((const float*)srv_->tensor_ptr(srv_->output_tid()), (long)sl.count * sl.T, srv_->ctc_slots(), srv_->ctc_step(),
                               amax_.p + sl.step_off, pmax_.p + sl.step_off, stream_);
      else
        srv::launch_argmax_softmax((const float*)srv_->tensor_ptr(srv_->output_tid()), (long)sl.count * sl.T, ot.c, ot.cs, amax_.p + sl.step_off,
                                   pmax_.p + sl.step_off, srv_->half(), stream_);
      launch_ctc(amax_.p + sl.step_off, pmax_.p + sl.step_off, sl.count, sl.T, max_len, ids_.p + (size_t)sl.first * max_len,
                 lens_.p + sl.first, scores_.p + sl.first, stream_);
    } else if (sl.ragged) {
      widths.resize(sl.count);
      long pix = 0;
      for (int j = 0; j < sl.count; ++j) { widths[j] = items[sl.first + j].tensor_w; pix += (long)imgH * widths[j]; }
      launch_line_pre_ragged(descs_.p + sl.first, sl.count, pix, imgH, lut_.p, x_.p, stream_);
      if (!net_.run_ragged(x_.p, imgH, widths.data(), sl.count, stream_, err)) return OCR_ERR_DEVICE;
      launch_ctc_ragged(amax_.p + sl.step_off, pmax_.p + sl.step_off, descs_.p + sl.first, sl.count, max_len,
                        ids_.p + (size_t)sl.first * max_len, lens_.p + sl.first, scores_.p + sl.first, stream_);
    } else {
      const int Wt = items[sl.first].tensor_w;
      launch_line_pre(descs_.p + sl.first, sl.count, imgH, Wt, lut_.p, false, x_.p, stream_);
      if (!net_.run(x_.p, sl.count, imgH, Wt, stream_, err)) return OCR_ERR_DEVICE;
      const TensorDesc& ot = net_.tensor(net_.output_tid());
      if (ot.h != 1) { err = "rec_img_h does not reduce to a single row"; return OCR_ERR_ARG; }
      if (ot.c != (int)labels_.size()) { err = "dictionary size does not match the CTC head"; return OCR_ERR_MODEL; }
      sl.T = ot.w;
      if (sl.T > Wt / 4 + 8) { err = "rec step buffer too small"; return OCR_ERR_CAPACITY; }
      launch_ctc(amax_.p + sl.step_off, pmax_.p + sl.step_off, sl.count, sl.T, max_len, ids_.p + (size_t)sl.first * max_len,
                 lens_.p + sl.first, scores_.p + sl.first, stream_);
    }
  }
  std::vector<int> h_ids((size_t)ni * max_len), h_lens(ni), h_amax;
  std::vector<float> h_scores(ni), h_pmax;
  ST_HIP(hipMemcpyAsync(h_ids.data(), ids_.p, h_ids.size() * sizeof(int), hipMemcpyDeviceToHost, stream_));
  ST_HIP(hipMemcpyAsync(h_lens.data(), lens_.p, (size_t)ni * sizeof(int), hipMemcpyDeviceToHost, stream_));
  ST_HIP(hipMemcpyAsync(h_scores.data(), scores_.p, (size_t)ni * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if (want_taps) {
    h_amax.resize(step_total);
    h_pmax.resize(step_total);
    ST_HIP(hipMemcpyAsync(h_amax.data(), amax_.p, step_total * sizeof(int), hipMemcpyDeviceToHost, stream_));
    ST_HIP(hipMemcpyAsync(h_pmax.data(), pmax_.p, step_total * sizeof(float), hipMemcpyDeviceToHost, stream_));
  }
  ST_HIP(g_stream_sync(stream_));
  net_.collect_timings();
  if (srv_) srv_->collect_timings();
  for (const Slot& sl : slots)
    for (int j = 0; j < sl.count; ++j) {
      const int q = sl.first + j, li = items[q].line;
      if (h_lens[q] > max_len) { err = "text longer than max_len"; return OCR_ERR_CAPACITY; }
      lens[li] = h_lens[q];
      scores[li] = h_scores[q];
      memcpy(ids + (size_t)li * max_len, h_ids.data() + (size_t)q * max_len, (size_t)h_lens[q] * sizeof(int));
      const int T = sl.ragged ? d[q].steps : sl.T;
      tap_T[li] = T;
      if (want_taps) {
        tap_off[li] = (int)tap_amax.size();
        const size_t o = sl.step_off + (sl.ragged ? (size_t)d[q].step0 : (size_t)j * sl.T);
        tap_amax.insert(tap_amax.end(), h_amax.begin() + o, h_amax.begin() + o + T);
        tap_pmax.insert(tap_pmax.end(), h_pmax.begin() + o, h_pmax.begin() + o + T);
      }
    }
  return OCR_OK;
}

// ================================================================= classifier
ClsStage::~ClsStage() {
  if (stream_) (void)hipStreamDestroy(stream_);
}

bool ClsStage::create(const ClsConfig& cfg, std::string& err, int& code) {
  cfg_ = cfg;
  code = OCR_ERR_ARG;
  // the reference's constructor parameter (TensorRT precision, ocr_det.cpp:50-56 / ocr_cls.cpp:135-140 / ocr_rec.cpp:167-172):
  // "fp32" = the bit-exact contract, "fp16" = f16 matrix products with f32 accumulation (Net::load); "int8" is refused
  if (cfg.precision != "fp32" && cfg.precision != "fp16") { err = "precision '" + cfg.precision + "' is not implemented (fp32 | fp16)"; return false; }
  if (cfg.batch_num < 1) { err = "cls_batch_num must be >= 1"; return false; }
  code = ocr_rt_init(cfg.device);
  if (code) { err = ocr_last_error(); return false; }
  code = OCR_ERR_MODEL;
  WeightMap w;
  if (!load_model_dir(cfg.model_dir, nullptr, "cls", w, err)) return false;
  if (!net_.load(embedded_plan("cls"), w, err, cfg.precision == "fp16")) return false;
  code = OCR_ERR_DEVICE;
  if (g_stream_create(&stream_) != hipSuccess) { err = "hipStreamCreate failed"; return false; }
  if (!timer_.init(err)) return false;
  const float mean[3] = {0.5f, 0.5f, 0.5f}, scale[3] = {1 / 0.5f, 1 / 0.5f, 1 / 0.5f};  // ocr_cls.h:93-94
  const auto lut = make_norm_lut(mean, scale);
  if (!lut_.ensure(lut.size(), err)) return false;
  if (g_memcpy(lut_.p, lut.data(), lut.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { err = "lut upload failed"; return false; }
  code = OCR_OK;
  return true;
}

int ClsStage::run(const ocr_img* imgs, int n, int* labels, float* scores, double times[3], std::string& err) {
  if (n < 0 || (n > 0 && (!imgs || !labels || !scores))) { err = "bad argument"; return OCR_ERR_ARG; }
  if (times) times[0] = times[1] = times[2] = 0;
  if (n == 0) return OCR_OK;
  ST_HIP(rt_set_device(cfg_.device));
  timer_.mark(0, stream_);
  std::vector<LineSrc> lines;
  if (!upload_lines(imgs, n, staging_, lines, stream_, err)) return OCR_ERR_DEVICE;
  timer_.mark(1, stream_);
  const int rc = run_lines(lines, labels, scores, err);
  timer_.mark(2, stream_);
  timer_.mark(3, stream_);
  (void)g_stream_sync(stream_);
  timer_.read(times);
  return rc;
}

int ClsStage::run_lines(const std::vector<LineSrc>& lines, int* labels, float* scores, std::string& err) {
  // Classifier::Run, /root/reference/src/ocr_cls.cpp:23-106: fixed 3x48x192 input, so the reference's
  // batches of cls_batch_num are independent rows of one launch here.
  const int n = (int)lines.size();
  const int imgH = 48, imgW = 192;
  std::vector<LineDesc> d(n);
  for (int i = 0; i < n; ++i) {
    const LineSrc& L = lines[i];
    const float ratio = float(L.w) / float(L.h);
    const int resize_w = ceilf(imgH * ratio) > imgW ? imgW : int(ceilf(imgH * ratio));
    d[i] = LineDesc::make(L.img, L.stride, L.x, L.y, L.w, L.h, resize_w, i, imgH);
  }
  if (!descs_.ensure(n, err) || !x_.ensure((size_t)n * imgH * imgW * 3, err)) return OCR_ERR_DEVICE;
  ST_HIP(hipMemcpyAsync(descs_.p, d.data(), n * sizeof(LineDesc), hipMemcpyHostToDevice, stream_));
  launch_line_pre(descs_.p, n, imgH, imgW, lut_.p, true, x_.p, stream_);
  const int* old_a = amax_.p;
  const float* old_p = pmax_.p;
  const float* old_q = probs_.p;
  if (!amax_.ensure(n, err) || !pmax_.ensure(n, err) || !probs_.ensure((size_t)n * 2, err)) return OCR_ERR_DEVICE;
  if (amax_.p != old_a || pmax_.p != old_p || probs_.p != old_q || !old_a) net_.set_head_outputs(probs_.p, amax_.p, pmax_.p);
  if (!net_.run(x_.p, n, imgH, imgW, stream_, err)) return OCR_ERR_DEVICE;
  tap_probs.resize((size_t)n * 2);
  ST_HIP(hipMemcpyAsync(labels, amax_.p, n * sizeof(int), hipMemcpyDeviceToHost, stream_));
  ST_HIP(hipMemcpyAsync(scores, pmax_.p, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
  ST_HIP(hipMemcpyAsync(tap_probs.data(), probs_.p, tap_probs.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
  ST_HIP(g_stream_sync(stream_));
  net_.collect_timings();
  return OCR_OK;
}

}  // namespace ocr
