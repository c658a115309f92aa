// Host layer above the C-ABI (include/ocr_hip.h): the reference's operator interface for this path
// with the same names, argument meaning and error behaviour, over an ImageView instead of cv::Mat.
//
//   PaddleOCR::DBDetector      /root/reference/include/paddle_ocr/ocr_det.h:28-131
//   PaddleOCR::Classifier      /root/reference/include/paddle_ocr/ocr_cls.h:28-104
//   PaddleOCR::CRNNRecognizer  /root/reference/include/paddle_ocr/ocr_rec.h:28-122
//   PaddleOCR::OCRRequest / WordResult / OCRResult / OCRWorker   include/paddle_ocr/ocr_worker.h:22-94
//   PaddleOCR::GPUWorkerPool   include/paddle_ocr/gpu_worker_pool.h:14-31
//
// Differences, all forced by the missing third-party types: images are `ImageView` (the fields of a
// CV_8UC3 cv::Mat that the path reads); the worker's result is an OCRResult plus a JSON string with
// the reference's keys (src/ocr_worker.cpp:155-190) written by a 40-line emitter instead of jsoncpp.
// Header-only; link with -locr_hip.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ocr_hip.h"
#include "jpeg_decode.h"

namespace PaddleOCR {

// The part of cv::Mat this path uses: CV_8UC3 BGR rows, possibly an ROI of a larger image.
struct ImageView {
  const uint8_t* data = nullptr;
  int rows = 0, cols = 0;
  size_t step = 0;  // bytes per row
  bool empty() const { return !data || rows <= 0 || cols <= 0; }
  ImageView roi(int x, int y, int w, int h) const { return ImageView{data + (size_t)y * step + (size_t)x * 3, h, w, step}; }
  ocr_img c() const { return ocr_img{data, rows, cols, step}; }
};

// An owning image (cv::Mat::clone()).
struct Image {
  std::vector<uint8_t> pixels;
  int rows = 0, cols = 0;
  // a JPEG that has only been entropy-decoded: the worker finishes it on the device (ocr_pipe_stage_jpeg); `pixels`
  // stays empty unless someone asks for them (materialise())
  std::shared_ptr<jpeg::Coefs> jpeg;
  Image() = default;
  explicit Image(const ImageView& v) : rows(v.rows), cols(v.cols) {
    if (!v.empty()) {
      pixels.resize((size_t)rows * cols * 3);
      for (int y = 0; y < rows; ++y) memcpy(&pixels[(size_t)y * cols * 3], v.data + (size_t)y * v.step, (size_t)cols * 3);
    }
  }
  ImageView view() const { return ImageView{pixels.data(), rows, cols, (size_t)cols * 3}; }
  bool empty() const { return pixels.empty() && !jpeg; }
  bool device_decodable() const { return pixels.empty() && jpeg; }
  void materialise() {  // the pixel half of the JPEG on the host
    if (!device_decodable()) return;
    if (!jpeg::Decoder::pixels(*jpeg, pixels, rows, cols)) throw std::runtime_error("JPEG reconstruction failed");
    jpeg.reset();
  }
  ocr_jpeg_img jpeg_desc() const {
    ocr_jpeg_img d;
    memset(&d, 0, sizeof d);
    d.rows = jpeg->rows; d.cols = jpeg->cols; d.ncomp = jpeg->ncomp; d.hmax = jpeg->hmax; d.vmax = jpeg->vmax;
    for (int i = 0; i < jpeg->ncomp; ++i) {
      const auto& c = jpeg->comp[i];
      d.comp[i].coef = c.coef.data();
      memcpy(d.comp[i].quant, c.quant, sizeof c.quant);
      d.comp[i].bw = c.bw; d.comp[i].bh = c.bh; d.comp[i].dw = c.dw; d.comp[i].dh = c.dh;
    }
    return d;
  }
};

inline void check_ocr(int rc, const char* what) {
  if (rc != OCR_OK) throw std::runtime_error(std::string(what) + ": " + ocr_last_error());
}

// Utility (the hot subset of /root/reference/include/paddle_ocr/utility.h)
class Utility {
 public:
  // Utility::GetRotateCropImage (utility.cpp:137-190): perspective-rectified crop of one det box.
  // Throws where the reference would die inside cv::Mat::operator() (box outside the image / empty).
  static Image GetRotateCropImage(const ImageView& srcimage, const std::vector<std::vector<int>>& box) {
    if (srcimage.empty() || box.size() != 4) throw std::runtime_error("GetRotateCropImage: bad input");
    int32_t b[8];
    for (int i = 0; i < 4; ++i) { b[2 * i] = box[i].at(0); b[2 * i + 1] = box[i].at(1); }
    Image out;
    check_ocr(ocr_rotate_crop_shape(srcimage.rows, srcimage.cols, b, &out.rows, &out.cols), "GetRotateCropImage");
    out.pixels.resize((size_t)out.rows * out.cols * 3);
    size_t off[2];
    int r, c;
    check_ocr(ocr_rotate_crop(srcimage.data, srcimage.rows, srcimage.cols, srcimage.step, b, 1, out.pixels.data(), out.pixels.size(),
                              off, &r, &c), "GetRotateCropImage");
    return out;
  }
};

class DBDetector {
 public:
  // Same parameter list as the reference constructor (ocr_det.h:60-75).  use_gpu must be true: this
  // build is the GPU path; gpu_mem / cpu threads / mkldnn / tensorrt are accepted and ignored.
  explicit DBDetector(const std::string& model_dir, const bool& use_gpu, const int& gpu_id, const int& /*gpu_mem*/,
                      const int& /*cpu_math_library_num_threads*/, const bool& /*use_mkldnn*/, const std::string& limit_type,
                      const int& limit_side_len, const double& det_db_thresh, const double& det_db_box_thresh,
                      const double& det_db_unclip_ratio, const std::string& det_db_score_mode, const bool& use_dilation,
                      const bool& /*use_tensorrt*/, const std::string& precision) {
    if (!use_gpu) throw std::runtime_error("DBDetector: this build has no CPU path (use_gpu must be true)");
    ocr_det_cfg c;
    ocr_det_cfg_default(&c);
    c.model_dir = model_dir.c_str(); c.device_id = gpu_id; c.limit_type = limit_type.c_str();
    c.limit_side_len = limit_side_len; c.det_db_thresh = det_db_thresh; c.det_db_box_thresh = det_db_box_thresh;
    c.det_db_unclip_ratio = det_db_unclip_ratio; c.det_db_score_mode = det_db_score_mode.c_str();
    c.use_dilation = use_dilation; c.precision = precision.c_str();
    check_ocr(ocr_det_create(&c, &h_), "DBDetector");  // the reference exit(1)s here (ocr_det.cpp:41-45); we throw
  }
  ~DBDetector() { ocr_det_destroy(h_); }
  DBDetector(const DBDetector&) = delete;
  DBDetector& operator=(const DBDetector&) = delete;

  // void Run(const cv::Mat&, vector<vector<vector<int>>>& boxes, vector<double>& times)  (ocr_det.h:95-97)
  void Run(const ImageView& img, std::vector<std::vector<std::vector<int>>>& boxes, std::vector<double>& times) {
    std::vector<int32_t> flat(1000 * 8);
    int n = 0;
    double t[3] = {0, 0, 0};
    ocr_img im = img.c();
    check_ocr(ocr_det_run(h_, &im, flat.data(), 1000, &n, t), "DBDetector::Run");
    boxes.clear();
    for (int i = 0; i < n; ++i) {
      std::vector<std::vector<int>> b(4, std::vector<int>(2));
      for (int k = 0; k < 4; ++k) { b[k][0] = flat[i * 8 + 2 * k]; b[k][1] = flat[i * 8 + 2 * k + 1]; }
      boxes.emplace_back(std::move(b));
    }
    times.insert(times.end(), t, t + 3);
  }

 private:
  ocr_det* h_ = nullptr;
};

class Classifier {
 public:
  explicit Classifier(const std::string& model_dir, const bool& use_gpu, const int& gpu_id, const int& /*gpu_mem*/,
                      const int& /*cpu_math_library_num_threads*/, const bool& /*use_mkldnn*/, const double& cls_thresh,
                      const bool& /*use_tensorrt*/, const std::string& precision, const int& cls_batch_num) {
    if (!use_gpu) throw std::runtime_error("Classifier: this build has no CPU path (use_gpu must be true)");
    ocr_cls_cfg c;
    ocr_cls_cfg_default(&c);
    c.model_dir = model_dir.c_str(); c.device_id = gpu_id; c.cls_thresh = cls_thresh; c.cls_batch_num = cls_batch_num;
    c.precision = precision.c_str();
    check_ocr(ocr_cls_create(&c, &h_), "Classifier");
  }
  ~Classifier() { ocr_cls_destroy(h_); }
  Classifier(const Classifier&) = delete;
  Classifier& operator=(const Classifier&) = delete;
  // outputs are pre-sized by the caller (ocr_worker.cpp:271-272)
  void Run(const std::vector<ImageView>& img_list, std::vector<int>& cls_labels, std::vector<float>& cls_scores,
           std::vector<double>& times) {
    std::vector<ocr_img> v;
    for (auto& i : img_list) v.push_back(i.c());
    double t[3] = {0, 0, 0};
    check_ocr(ocr_cls_run(h_, v.data(), (int)v.size(), cls_labels.data(), cls_scores.data(), t), "Classifier::Run");
    times.insert(times.end(), t, t + 3);
  }

 private:
  ocr_cls* h_ = nullptr;
};

class CRNNRecognizer {
 public:
  explicit CRNNRecognizer(const std::string& model_dir, const bool& use_gpu, const int& gpu_id, const int& /*gpu_mem*/,
                          const int& /*cpu_math_library_num_threads*/, const bool& /*use_mkldnn*/,
                          const std::string& label_path, const bool& /*use_tensorrt*/, const std::string& precision,
                          const int& rec_batch_num, const int& rec_img_h, const int& rec_img_w) {
    if (!use_gpu) throw std::runtime_error("CRNNRecognizer: this build has no CPU path (use_gpu must be true)");
    ocr_rec_cfg c;
    ocr_rec_cfg_default(&c);
    c.model_dir = model_dir.c_str(); c.device_id = gpu_id; c.label_path = label_path.c_str();
    c.rec_batch_num = rec_batch_num; c.rec_img_h = rec_img_h; c.rec_img_w = rec_img_w; c.precision = precision.c_str();
    check_ocr(ocr_rec_create(&c, &h_), "CRNNRecognizer");
  }
  ~CRNNRecognizer() { ocr_rec_destroy(h_); }
  CRNNRecognizer(const CRNNRecognizer&) = delete;
  CRNNRecognizer& operator=(const CRNNRecognizer&) = delete;
  // rec_texts / rec_text_scores are pre-sized by the caller (ocr_worker.cpp:286-287); lines whose
  // score would be NaN keep their previous content, like the reference's `continue` (ocr_rec.cpp:123-125)
  void Run(const std::vector<ImageView>& img_list, std::vector<std::string>& rec_texts, std::vector<float>& rec_text_scores,
           std::vector<double>& times) {
    const int n = (int)img_list.size(), max_len = 512;
    std::vector<ocr_img> v;
    for (auto& i : img_list) v.push_back(i.c());
    std::vector<int32_t> ids((size_t)n * max_len);
    std::vector<int> lens(n);
    std::vector<float> scores(n);
    double t[3] = {0, 0, 0};
    check_ocr(ocr_rec_run(h_, v.data(), n, ids.data(), max_len, lens.data(), scores.data(), t), "CRNNRecognizer::Run");
    for (int i = 0; i < n; ++i) {
      if (lens[i] == 0) continue;
      std::string s;
      for (int k = 0; k < lens[i]; ++k) s += ocr_rec_label(h_, ids[(size_t)i * max_len + k]);
      rec_texts[i] = std::move(s);
      rec_text_scores[i] = scores[i];
    }
    times.insert(times.end(), t, t + 3);
  }

 private:
  ocr_rec* h_ = nullptr;
};

// ---------------------------------------------------------------- worker (ocr_worker.h:22-94)
struct OCRRequest {
  int request_id;
  Image image_data;  // deep copy, like `image_data(img.clone())`
  std::promise<std::string> result_promise;
  OCRRequest(int id, const ImageView& img) : request_id(id), image_data(img) {}
  OCRRequest(int id, Image&& img) : request_id(id), image_data(std::move(img)) {}  // keeps a JPEG coefficient payload
};
struct WordResult {
  std::string text;
  std::vector<std::vector<int>> box;
  float confidence;
};
struct OCRResult {
  int request_id = 0;
  bool success = false;
  int width = 0, height = 0;
  std::string error_message;
  std::vector<WordResult> words;
  double processing_time_ms = 0;
};

namespace detail {
inline void json_escape(std::string& o, const std::string& s) {
  o += '"';
  for (unsigned char ch : s) {
    switch (ch) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (ch < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", ch); o += b; }
        else o += (char)ch;  // emitUTF8 = true (ocr_worker.cpp:188)
    }
  }
  o += '"';
}
// keys and order as jsoncpp emits them (alphabetical), no indentation (ocr_worker.cpp:155-190)
inline std::string result_json(const OCRResult& r, int worker_id) {
  std::string o = "{";
  char buf[64];
  if (!r.success) { o += "\"error\":"; json_escape(o, r.error_message); o += ","; }
  snprintf(buf, sizeof buf, "\"height\":%d,", r.height); o += buf;
  snprintf(buf, sizeof buf, "\"processing_time_ms\":%.17g,", r.processing_time_ms); o += buf;
  snprintf(buf, sizeof buf, "\"request_id\":%d,", r.request_id); o += buf;
  o += std::string("\"success\":") + (r.success ? "true" : "false") + ",";
  snprintf(buf, sizeof buf, "\"width\":%d,", r.width); o += buf;
  if (r.success) {
    o += "\"words\":[";
    for (size_t i = 0; i < r.words.size(); ++i) {
      const WordResult& w = r.words[i];
      if (i) o += ",";
      o += "{\"box\":[";
      for (size_t k = 0; k < w.box.size(); ++k) {
        snprintf(buf, sizeof buf, "%s[%d,%d]", k ? "," : "", w.box[k][0], w.box[k][1]);
        o += buf;
      }
      snprintf(buf, sizeof buf, "],\"confidence\":%.17g,\"text\":", (double)w.confidence);
      o += buf;
      json_escape(o, w.text);
      o += "}";
    }
    o += "],";
  }
  snprintf(buf, sizeof buf, "\"worker_id\":%d}", worker_id); o += buf;
  return o;
}
}  // namespace detail

class OCRWorker {
 public:
  // OCRWorker(worker_id, model_dir, use_gpu, gpu_id = 0, enable_cls = false); hyper-parameters are the
  // literals of the reference constructor (ocr_worker.cpp:21-63) — they are ocr_pipe_cfg_default().
  // rotate_crops (extension, default off): crops through Utility::GetRotateCropImage instead of ROI views.
  // max_batch (extension): the worker thread takes up to this many QUEUED requests at once and runs them
  // as one ocr_pipe_run - the GPU is fed whole batches under concurrent load, a lone request is still
  // served immediately (nothing ever waits for a batch to fill).  Per-image results do not depend on what
  // else is in the batch (tests/test_gpu_parity.py, tests/test_ipc_service.py).  0 = OCR_WORKER_MAX_BATCH
  // from the environment, else 16; 1 = the reference's one-request-at-a-time loop.  (16, not 32, since round 4: under
  // 8 .. 96 closed-loop clients the saturated rate is the same, 470-476 requests/s, and the tail is shorter - p99 at 64
  // clients 186 ms against 241 - because a request that just missed a batch waits for a shorter one; profiles/r4_service_sweep.jsonl)
  OCRWorker(int worker_id, const std::string& model_dir, bool use_gpu, int gpu_id = 0, bool enable_cls = false,
            bool rotate_crops = false, int max_batch = 0)
      : worker_id_(worker_id), running_(false), is_idle_(true) {
    if (max_batch <= 0) {
      const char* e = getenv("OCR_WORKER_MAX_BATCH");
      max_batch = e ? atoi(e) : 16;
    }
    max_batch_ = max_batch < 1 ? 1 : (max_batch > 256 ? 256 : max_batch);  // result buffers: 1 MB of ids per image
    gpu_id_ = gpu_id;
    if (!use_gpu) throw std::runtime_error("OCRWorker: this build has no CPU path (use_gpu must be true)");
    det_dir_ = model_dir + "/det"; cls_dir_ = model_dir + "/cls"; rec_dir_ = model_dir + "/rec";
    dict_ = model_dir + "/rec/ppocr_keys_v1.txt";
    ocr_pipe_cfg c;
    ocr_pipe_cfg_default(&c);
    c.det.model_dir = det_dir_.c_str(); c.det.device_id = gpu_id;
    c.cls.model_dir = cls_dir_.c_str();
    c.rec.model_dir = rec_dir_.c_str(); c.rec.label_path = dict_.c_str();
    c.enable_cls = enable_cls;
    c.crop_mode = rotate_crops ? OCR_CROP_ROTATE : OCR_CROP_BOUNDING_RECT;
    // The reference hard-codes the model hyper-parameters in OCRWorker::OCRWorker (src/ocr_worker.cpp:21-63: limit 512,
    // rec 28x192, cls off) and ocr_pipe_cfg_default carries those literals.  Extension for load tests of other
    // operating points (tools/service_load.py): OCR_WORKER_DET_LIMIT, OCR_WORKER_REC_H / _W, OCR_WORKER_CLS=1.
    if (const char* e = getenv("OCR_WORKER_DET_LIMIT")) if (atoi(e) >= 32) c.det.limit_side_len = atoi(e);
    if (const char* e = getenv("OCR_WORKER_REC_H")) if (atoi(e) >= 8) c.rec.rec_img_h = atoi(e);
    if (const char* e = getenv("OCR_WORKER_REC_W")) if (atoi(e) >= 8) c.rec.rec_img_w = atoi(e);
    if (const char* e = getenv("OCR_WORKER_CLS")) if (e[0] == '1') c.enable_cls = 1;
    // (the reference's worker passes the literal "fp32" to its three stage constructors, ocr_worker.cpp:34-60; the stages'
    // own parameter takes "fp16" - OCR_WORKER_PRECISION=fp16 hands it to all three)
    if (const char* e = getenv("OCR_WORKER_PRECISION")) {
      precision_ = e;
      c.det.precision = c.cls.precision = c.rec.precision = precision_.c_str();
    }
    check_ocr(ocr_pipe_create(&c, &pipe_), "OCRWorker");
  }
  virtual ~OCRWorker() { stop(); ocr_pipe_destroy(pipe_); }

  void start() {
    if (running_) return;
    running_ = true;
    worker_thread_ = std::thread(&OCRWorker::workerLoop, this);
  }
  void stop() {
    {
      // under the queue mutex: a worker between its predicate check and its wait would otherwise miss the wake-up
      std::lock_guard<std::mutex> lock(queue_mutex_);
      if (!running_) return;
      running_ = false;
    }
    cv_.notify_all();
    if (worker_thread_.joinable()) worker_thread_.join();
    // requests still queued get an answer, not a broken promise
    std::queue<std::shared_ptr<OCRRequest>> rest;
    { std::lock_guard<std::mutex> lock(queue_mutex_); rest.swap(request_queue_); }
    for (; !rest.empty(); rest.pop()) {
      OCRResult r;
      r.request_id = rest.front()->request_id;
      r.success = false;
      r.error_message = "worker stopped before the request was processed";
      rest.front()->result_promise.set_value(detail::result_json(r, worker_id_));
    }
  }
  void addRequest(std::shared_ptr<OCRRequest> request) {
    { std::lock_guard<std::mutex> lock(queue_mutex_); request_queue_.push(request); }
    cv_.notify_one();
  }
  bool isIdle() const { return is_idle_; }
  int getWorkerId() const { return worker_id_; }
  int getGpuId() const { return gpu_id_; }
  // load figures for the pool's dispatch and its statistics (round 6): requests waiting in this worker's queue, plus the
  // batch it is running; requests it has answered
  int queueDepth() {
    std::lock_guard<std::mutex> lock(queue_mutex_);
    return (int)request_queue_.size() + in_flight_.load();
  }
  long requestsServed() const { return served_.load(); }

  // processRequest (ocr_worker.cpp:213-311) — public here so tests can call it without the thread
  OCRResult processRequest(const OCRRequest& request) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    OCRResult result;
    result.request_id = request.request_id;
    result.success = false;
    if (request.image_data.empty()) { result.error_message = "Empty image data provided"; return result; }
    result.width = request.image_data.cols;
    result.height = request.image_data.rows;
    std::vector<ocr_word> words(1000);
    std::vector<int32_t> ids(1000 * 256);
    int off = 0, n = 0;
    int rc;
    if (request.image_data.device_decodable()) {  // JPEG: pixels are produced on the device, straight into the staging slot
      const ocr_jpeg_img jd = request.image_data.jpeg_desc();
      rc = ocr_pipe_stage_jpeg(pipe_, 0, &jd, 1);
      if (rc == OCR_OK) rc = ocr_pipe_run_staged(pipe_, 0, words.data(), (int)words.size(), &off, &n, ids.data(), (int)ids.size(), nullptr);
    } else {
      ocr_img im = request.image_data.view().c();
      rc = ocr_pipe_run(pipe_, &im, 1, words.data(), (int)words.size(), &off, &n, ids.data(), (int)ids.size(), nullptr);
    }
    if (rc != OCR_OK) { result.error_message = ocr_last_error(); return result; }  // the reference's catch (...) path
    result.success = true;
    for (int i = 0; i < n; ++i) {
      WordResult w;
      for (int k = 0; k < words[i].ids_len; ++k) w.text += ocr_pipe_label(pipe_, ids[words[i].ids_off + k]);
      w.confidence = words[i].confidence;
      w.box.assign(4, std::vector<int>(2));
      for (int k = 0; k < 4; ++k) { w.box[k][0] = words[i].box[2 * k]; w.box[k][1] = words[i].box[2 * k + 1]; }
      result.words.push_back(std::move(w));
    }
    result.processing_time_ms =
        std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    return result;
  }

  // The same for several requests in one pipeline run; results[i] belongs to requests[i].  A failure of
  // the batched run (one oversized image, say) falls back to request-by-request processing, so a request
  // fails or succeeds exactly as it would alone.  processing_time_ms is the wall time of the run a request
  // was part of.
  std::vector<OCRResult> processBatch(const std::vector<const OCRRequest*>& requests) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    std::vector<OCRResult> results(requests.size());
    std::vector<ocr_img> imgs;
    std::vector<ocr_jpeg_img> jimgs;
    std::vector<size_t> owner;
    // a batch of JPEGs only is decoded on the device; a mixed batch takes the host path for its JPEGs
    bool all_jpeg = true;
    for (size_t i = 0; i < requests.size(); ++i)
      if (!requests[i]->image_data.empty() && !requests[i]->image_data.device_decodable()) all_jpeg = false;
    for (size_t i = 0; i < requests.size(); ++i) {
      results[i].request_id = requests[i]->request_id;
      if (requests[i]->image_data.empty()) { results[i].error_message = "Empty image data provided"; continue; }
      results[i].width = requests[i]->image_data.cols;
      results[i].height = requests[i]->image_data.rows;
      if (all_jpeg) jimgs.push_back(requests[i]->image_data.jpeg_desc());
      else {
        const_cast<OCRRequest*>(requests[i])->image_data.materialise();
        imgs.push_back(requests[i]->image_data.view().c());
      }
      owner.push_back(i);
    }
    if (owner.empty()) return results;
    if (owner.size() == 1) { results[owner[0]] = processRequest(*requests[owner[0]]); return results; }
    const int k = (int)owner.size();
    // result buffers live with the worker (grown once): 1000 words of up to 256 ids per image, as processRequest
    if (batch_words_.size() < (size_t)k * 1000) { batch_words_.resize((size_t)k * 1000); batch_ids_.resize((size_t)k * 1000 * 256); }
    std::vector<ocr_word>& words = batch_words_;
    std::vector<int32_t>& ids = batch_ids_;
    std::vector<int> off(k), cnt(k);
    int rc;
    if (all_jpeg) {
      rc = ocr_pipe_stage_jpeg(pipe_, 0, jimgs.data(), k);
      if (rc == OCR_OK) rc = ocr_pipe_run_staged(pipe_, 0, words.data(), k * 1000, off.data(), cnt.data(), ids.data(), k * 1000 * 256, nullptr);
    } else {
      rc = ocr_pipe_run(pipe_, imgs.data(), k, words.data(), k * 1000, off.data(), cnt.data(), ids.data(), k * 1000 * 256, nullptr);
    }
    if (rc != OCR_OK) {
      for (size_t j = 0; j < owner.size(); ++j) results[owner[j]] = processRequest(*requests[owner[j]]);
      return results;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    for (int j = 0; j < k; ++j) {
      OCRResult& r = results[owner[j]];
      r.success = true;
      for (int i = off[j]; i < off[j] + cnt[j]; ++i) {
        WordResult w;
        for (int q = 0; q < words[i].ids_len; ++q) w.text += ocr_pipe_label(pipe_, ids[words[i].ids_off + q]);
        w.confidence = words[i].confidence;
        w.box.assign(4, std::vector<int>(2));
        for (int q = 0; q < 4; ++q) { w.box[q][0] = words[i].box[2 * q]; w.box[q][1] = words[i].box[2 * q + 1]; }
        r.words.push_back(std::move(w));
      }
      r.processing_time_ms = ms;
    }
    return results;
  }
  int maxBatch() const { return max_batch_; }

 private:
  void workerLoop() {
    while (running_) {
      std::vector<std::shared_ptr<OCRRequest>> batch;
      {
        std::unique_lock<std::mutex> lock(queue_mutex_);
        cv_.wait(lock, [this] { return !request_queue_.empty() || !running_; });
        if (!running_) break;
        auto take = [&] {
          while (!request_queue_.empty() && (int)batch.size() < max_batch_) {
            batch.push_back(request_queue_.front());
            request_queue_.pop();
          }
        };
        take();
        if (!batch.empty()) { is_idle_ = false; in_flight_ = (int)batch.size(); }
      }
      if (batch.empty()) continue;
      std::vector<const OCRRequest*> ptrs;
      for (auto& b : batch) ptrs.push_back(b.get());
      std::vector<OCRResult> rs;
      try {
        rs = processBatch(ptrs);
      } catch (const std::exception& e) {
        rs.assign(batch.size(), OCRResult());
        for (size_t i = 0; i < batch.size(); ++i) {
          rs[i].request_id = batch[i]->request_id;
          rs[i].success = false;
          rs[i].error_message = e.what();
        }
      }
      for (size_t i = 0; i < batch.size(); ++i) batch[i]->result_promise.set_value(detail::result_json(rs[i], worker_id_));
      served_ += (long)batch.size();
      in_flight_ = 0;
      is_idle_ = true;
    }
  }

  int worker_id_;
  int gpu_id_ = 0;
  int max_batch_ = 1;
  std::string precision_;  // OCR_WORKER_PRECISION (kept alive for the configuration's pointers)
  std::vector<ocr_word> batch_words_;
  std::vector<int32_t> batch_ids_;
  std::atomic<bool> running_, is_idle_;
  std::atomic<int> in_flight_{0};
  std::atomic<long> served_{0};
  std::thread worker_thread_;
  std::queue<std::shared_ptr<OCRRequest>> request_queue_;
  std::mutex queue_mutex_;
  std::condition_variable cv_;
  ocr_pipe* pipe_ = nullptr;
  std::string det_dir_, cls_dir_, rec_dir_, dict_;
};

// ---------------------------------------------------------------- pool (gpu_worker_pool.h:14-31)
// The reference pins every worker to GPU 0 (gpu_worker_pool.cpp:12-16); here worker i runs on GPU
// i mod ocr_rt_device_count(): whole images are sharded over the node's GPUs, nothing else changes.
class GPUWorkerPool {
 public:
  // Dispatch: the reference's rule (first idle worker, else round robin: gpu_worker_pool.cpp:46-59) is the default.
  // LeastQueued (round 6, an option): the worker with the fewest requests queued or running - under a stream of mixed-size
  // requests round robin hands a slow worker as many requests as a fast one (profiles/r6_pool_load.json has the histograms).
  // OCR_POOL_DISPATCH=least in the environment selects it for pools that do not call setDispatch.
  enum Dispatch { IdleFirstRoundRobin = 0, LeastQueued = 1 };
  GPUWorkerPool(const std::string& model_dir, int num_workers = 2) : next_worker_index_(0) {
    const int ngpu = ocr_rt_device_count();
    if (ngpu <= 0) throw std::runtime_error("GPUWorkerPool: no HIP device visible");
    for (int i = 0; i < num_workers; ++i) workers_.emplace_back(std::make_unique<OCRWorker>(i, model_dir, true, i % ngpu));
    if (const char* e = getenv("OCR_POOL_DISPATCH")) if (e[0] == 'l') dispatch_ = LeastQueued;
    depth_hist_.assign(65, 0);
  }
  ~GPUWorkerPool() { stop(); }
  void start() { for (auto& w : workers_) w->start(); }
  void stop() { for (auto& w : workers_) w->stop(); }
  void setDispatch(Dispatch d) { dispatch_ = d; }
  std::future<std::string> submitRequest(std::shared_ptr<OCRRequest> request) {
    auto future = request->result_promise.get_future();
    getAvailableWorker()->addRequest(request);
    return future;
  }
  int getOptimalWorkerCount() { return ocr_rt_device_count(); }  // declared, never defined in the reference
  int workerDevice(int i) const { return workers_[i]->getGpuId(); }
  int workerCount() const { return (int)workers_.size(); }
  // statistics: requests answered per worker; histogram of the chosen worker's depth (queued + running) at submit time,
  // bucket 64 = 64 or more
  std::vector<long> requestsPerWorker() const { std::vector<long> v; for (auto& w : workers_) v.push_back(w->requestsServed()); return v; }
  std::vector<long> submitDepthHistogram() { std::lock_guard<std::mutex> lock(workers_mutex_); return depth_hist_; }

 private:
  OCRWorker* getAvailableWorker() {
    std::lock_guard<std::mutex> lock(workers_mutex_);
    OCRWorker* pick = nullptr;
    if (dispatch_ == LeastQueued) {
      int best = 1 << 30;
      const int n = (int)workers_.size(), start = next_worker_index_.fetch_add(1) % n;  // (ties: rotate, so that an idle pool still spreads)
      for (int k = 0; k < n; ++k) {
        OCRWorker* w = workers_[(start + k) % n].get();
        const int d = w->queueDepth();
        if (d < best) { best = d; pick = w; }
      }
    } else {  // first idle, else round robin (gpu_worker_pool.cpp:46-59)
      for (auto& w : workers_) if (w->isIdle()) { pick = w.get(); break; }
      if (!pick) pick = workers_[next_worker_index_.fetch_add(1) % (int)workers_.size()].get();
    }
    const int d = pick->queueDepth();
    depth_hist_[d > 64 ? 64 : d]++;
    return pick;
  }
  std::vector<std::unique_ptr<OCRWorker>> workers_;
  std::mutex workers_mutex_;
  std::atomic<int> next_worker_index_;
  Dispatch dispatch_ = IdleFirstRoundRobin;
  std::vector<long> depth_hist_;
};

}  // namespace PaddleOCR
