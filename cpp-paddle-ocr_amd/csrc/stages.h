// Stage objects behind the C-ABI: detector, classifier, recognizer.  Each owns one HIP stream, its
// network and its device buffers ("one handle per host thread").
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "../../include/ocr_hip.h"
#include "kernels_post.h"
#include "kernels_pre.h"
#include "net.h"
#include "srv_net.h"

namespace ocr {

// growable device buffer
template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  ~DevBuf() { if (p) (void)g_free(p); }
  bool ensure(size_t n, std::string& err) {
    if (n <= cap) return true;
    if (p) (void)g_free(p);
    p = nullptr;
    cap = 0;
    hipError_t e = g_malloc(&p, n * sizeof(T));
    if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
    cap = n;
    return true;
  }
};

struct StageTimer {  // pre / infer / post in ms from HIP events on the stage's stream
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool init(std::string& err);
  ~StageTimer();
  void mark(int i, hipStream_t s) { (void)hipEventRecord(ev[i], s); }
  void read(double times[3]);
};

struct DetConfig {
  std::string model_dir, limit_type = "max", score_mode = "fast", precision = "fp32";
  int device = 0, limit_side_len = 512, use_dilation = 0, max_batch = 1;
  double thresh = 0.2, box_thresh = 0.4, unclip_ratio = 1.8;
  int cv_compat = 410;  // OCR_CV_45 | OCR_CV_410 (resolved: never 0 here)
};

class DetStage {
 public:
  ~DetStage();
  bool create(const DetConfig& cfg, std::string& err, int& code);
  // imgs: `count` same-size host images.  boxes [count][cap][8], n [count].
  int run(const ocr_img* imgs, int count, int32_t* boxes, int cap, int* n, double times[3], std::string& err);
  // the same on images that already sit in device memory (packed BGR rows, one image every img_bytes)
  // prob_override (device, [count][rh][rw] f32, may be null): the benchmark protocol of SURVEY.md §8d —
  // the network still runs, but thresholding / scoring read this map instead of the network's.
  int run_device(const uint8_t* dev_imgs, size_t img_bytes, size_t stride, int rows, int cols, int count, int32_t* boxes,
                 int cap, int* n, double times[3], std::string& err, const float* prob_override = nullptr);
  // ---- batches of MIXED sizes: one ragged network launch for all size groups, post-processing per group -----------
  // A size group = `count` images of rows x cols, packed BGR, starting `off` bytes into the batch's device buffer;
  // prob_off = floats into the benchmark protocol's probability-map buffer (prob_override of run_device).
  struct MixedGroup { int rows, cols, count; size_t off, prob_off; };
  // Enqueues resize + normalise of every group and ONE network pass over all images (Net::run_ragged_images: every
  // image keeps its own size) on this stage's stream and records `done()`; the maps of group i are then at
  // mixed_prob(i) / mixed_bitmap(i).  The caller runs post_group per group - on this stage or, concurrently, on other
  // DetStage instances (their streams wait for done()).
  int mixed_net(const uint8_t* base, const MixedGroup* groups, int ngroups, const float* prob_override, std::string& err);
  const float* mixed_prob(int gi) const { return mixed_probs_[gi]; }
  const uint8_t* mixed_bitmap(int gi) const { return bitmap_.p + mixed_pix_[gi]; }
  hipEvent_t done() const { return mixed_done_; }
  void collect_timings() { net_.collect_timings(); }  // after the stream that ran mixed_net has been synchronised
  // BoxesFromBitmap + FilterTagDetRes of EVERY group of the last mixed_net in one pass (the post-processing kernels
  // take each image's own map size: kernels_post.h, PostImg); boxes [images][cap][8] and n [images] in group order.
  // Not with use_dilation (the 2x2 dilate is per size): the caller then goes through post_group.
  int post_mixed(const MixedGroup* groups, int ngroups, int32_t* boxes, int cap, int* n, std::string& err);
  // BoxesFromBitmap + FilterTagDetRes of one size group whose maps sit in device memory (another stage's mixed_net)
  int post_group(const float* prob, const uint8_t* bitmap, const MixedGroup& g, hipEvent_t wait_for, int32_t* boxes, int cap, int* n,
                 std::string& err);
  // mutable view of the uploaded copies (the request's clone: cls rotation happens in place on it)
  uint8_t* dev_images_mut() { return src_.p; }
  int post_only(const float* prob, int rows, int cols, int src_rows, int src_cols, int32_t* boxes, int cap, int* n,
                std::string& err);
  // device-resident copy of the last uploaded images (for the fused pipeline)
  const uint8_t* dev_images() const { return src_.p; }
  size_t dev_image_bytes() const { return (size_t)src_rows_ * src_cols_ * 3; }
  hipStream_t stream() const { return stream_; }
  Net& net() { return net_; }
  int last_count = 0, last_h = 0, last_w = 0;
  const float* prob_dev() const { return srv_ ? (const float*)srv_->tensor_ptr(srv_->output_tid()) : net_.tensor_ptr(net_.output_tid()); }
  // the server detector (BASELINE configs[4]: a model directory whose arch.txt says srv_det), or null
  SrvNet* srv() { return srv_.get(); }
  const uint8_t* bitmap_dev() const { return cfg_.use_dilation ? bitmap2_.p : bitmap_.p; }
  const uint8_t* resized_dev() const { return resized_.p; }
  const DetConfig& cfg() const { return cfg_; }
  static void resize_shape(int h, int w, const std::string& limit_type, int limit_side_len, int& rh, int& rw,
                           float& ratio_h, float& ratio_w);

 private:
  bool ensure_post(int count, size_t px, int H, int W, int cap, std::string& err);
  int post_launch(PostArgs a, int count, int32_t* boxes, int cap, int* n, std::string& err);
  int run_post(int count, int H, int W, const float* prob, float ratio_h, float ratio_w, int src_h, int src_w,
               int32_t* boxes, int cap, int* n, std::string& err, const uint8_t* bitmap = nullptr);
  std::vector<size_t> mixed_pix_;            // first pixel of every group in the ragged maps
  const float* mixed_prob_base_ = nullptr;   // what post_mixed's per-image offsets are relative to
  DevBuf<PostImg> post_img_;
  std::vector<const float*> mixed_probs_;
  hipEvent_t mixed_done_ = nullptr;
  DetConfig cfg_;
  Net net_;
  std::unique_ptr<SrvNet> srv_;
  hipStream_t stream_ = nullptr;
  StageTimer timer_;
  DevBuf<float> lut_, x_, prob_in_;
  DevBuf<uint8_t> src_, resized_, bitmap_, bitmap2_, touch_;
  DevBuf<int> labels_, chunk_cnt_, ncont_all_, ncont_, starts_, npts_, poff_, iscratch_, cand_boxes_, cand_valid_, status_, out_boxes_,
      out_n_;
  DevBuf<unsigned long long> pool_;
  DevBuf<unsigned> mask_pool_, mask_top_;
  DevBuf<int> pool_need_;   // [images] keys the traced borders of an image would have needed (kernels_post.h)
  size_t mask_words_ = 0;   // words of polygon-score scratch per image in this call
  // words of polygon-score scratch per image: masks + crossing lists of all its borders (8 map areas of bits + slack)
  static size_t mask_words(int H, int W) { return (((size_t)H * W / 4 + (1u << 16)) + 31) & ~(size_t)31; }
  int src_rows_ = 0, src_cols_ = 0;
  int ithresh_ = 0;
  int pool_cap_ = 0;
  int bm_n_ = 0;  // 1 once the fused-bitmap pointer has been handed to the network
};

struct LineSrc {  // a text-line image living in device memory
  const uint8_t* img;
  size_t stride;
  int x, y, w, h;
};

struct RecConfig {
  std::string model_dir, label_path, precision = "fp32";
  int device = 0, batch_num = 16, img_h = 28, img_w = 192;
  int sort_mode = 0;  // OCR_SORT_STD | OCR_SORT_STABLE
};

class RecStage {
 public:
  ~RecStage();
  bool create(const RecConfig& cfg, std::string& err, int& code);
  int run(const ocr_img* imgs, int n, int32_t* ids, int max_len, int* lens, float* scores, double times[3],
          std::string& err);
  // lines already on the device (the stream must be ordered after whatever produced them).
  // seg: offsets of per-image segments (size nimages+1); the reference's aspect-sort / batch-of-16
  // rule is applied inside each segment, launches are shared across segments of equal tensor width.
  int run_lines(const std::vector<LineSrc>& lines, const std::vector<int>& seg, int32_t* ids, int max_len, int* lens,
                float* scores, std::string& err);
  bool want_taps = true;
  int max_lines_per_launch = 4096;  // bounds the activation arena of one launch: this many 48x320 lines' pixels (~1.2 MB each)
  const std::vector<std::string>& labels() const { return labels_; }
  hipStream_t stream() const { return stream_; }
  Net& net() { return net_; }
  SrvNet* srv() { return srv_.get(); }  // the server recognizer (arch.txt: srv_rec), or null
  // per-step taps of the last run, in input order: T per line, amax/pmax concatenated
  std::vector<int> tap_T, tap_off;
  std::vector<int> tap_amax;
  std::vector<float> tap_pmax;

 private:
  RecConfig cfg_;
  // One network instance, one stream: the lines of a call run as ONE ragged launch list whatever their tensor widths
  // (run_lines); rounds 1-2 launched once per distinct width on up to six streams ("lanes").
  Net net_;
  std::unique_ptr<SrvNet> srv_;
  hipStream_t stream_ = nullptr;
  StageTimer timer_;
  std::vector<std::string> labels_;
  DevBuf<float> lut_, pmax_, scores_, x_;
  DevBuf<int> amax_, ids_, lens_;
  DevBuf<uint8_t> staging_;
  DevBuf<LineDesc> descs_;
};

struct ClsConfig {
  std::string model_dir, precision = "fp32";
  int device = 0, batch_num = 8;
  double thresh = 0.98;
};

class ClsStage {
 public:
  ~ClsStage();
  bool create(const ClsConfig& cfg, std::string& err, int& code);
  int run(const ocr_img* imgs, int n, int* labels, float* scores, double times[3], std::string& err);
  int run_lines(const std::vector<LineSrc>& lines, int* labels, float* scores, std::string& err);
  hipStream_t stream() const { return stream_; }
  Net& net() { return net_; }
  std::vector<float> tap_probs;  // [n][2] of the last run

 private:
  ClsConfig cfg_;
  Net net_;
  hipStream_t stream_ = nullptr;
  StageTimer timer_;
  DevBuf<float> lut_, x_, pmax_, probs_;
  DevBuf<int> amax_;
  DevBuf<uint8_t> staging_;
  DevBuf<LineDesc> descs_;
};

// Uploads ROI views into one packed staging buffer and describes them as device lines.
bool upload_lines(const ocr_img* imgs, int n, DevBuf<uint8_t>& staging, std::vector<LineSrc>& out, hipStream_t s,
                  std::string& err);

// 3x256 normalisation LUT: Normalize::Run per byte value (convertTo 1/255, then scale/shift)
std::vector<float> make_norm_lut(const float mean[3], const float scale[3]);

}  // namespace ocr
