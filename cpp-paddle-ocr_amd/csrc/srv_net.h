// Plan executor of the "server" networks (BASELINE configs[4]; plans/srv_det.plan, plans/srv_rec.plan - hand-written, NOT
// reference artifacts): ResNet50-vd + DBFPN + DB head, SVTR-large + CTC head, on the kernel family of srv_kernels.hip.
// Same role as Net (net.h) for the mobile graphs - one instance per model per handle, the reference's "one predictor per stage
// object" (/root/reference/include/paddle_ocr/ocr_det.h:101) - with the precision the config names:
//   half = true   precision "fp16" (/root/reference/src/ocr_det.cpp:50-57): f16 tensors, f16 matrix instructions, f32 accumulation
//   half = false  the PARITY TWIN: the same launch list on float, bit-identical to the oracle's run of the plan
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <string>
#include <vector>

#include "net.h"  // KernelTiming
#include "pd_format.h"
#include "srv_kernels.h"

namespace ocr {

struct SrvTensor {
  int n = 0, h = 0, w = 0, c = 0, cs = 0;  // cs: stored channels (multiple of 8; f32 outputs: their row pitch)
  bool f32 = false;                        // network input, the detector's probability map, the recognizer's logits
  size_t offset = 0;                       // bytes into the arena
  long pixels() const { return (long)n * h * w; }
  size_t bytes(bool half) const { return (size_t)pixels() * cs * (f32 ? 4 : (half ? 2 : 4)); }
};

class SrvNet {
 public:
  ~SrvNet();
  bool load(const char* plan_text, const WeightMap& weights, bool half, std::string& err);
  bool half() const { return half_; }
  // x: device f32 [N,H,W,3] plain NHWC (already normalised).  Binds the shape if it is new (arena, launch list, tile
  // configuration of every GEMM - timed on the device once per (layer, shape) unless OCR_SRV_TUNE=0) and enqueues the network.
  bool run(const float* x, int N, int H, int W, hipStream_t s, std::string& err);
  int ntensors() const { return ntensors_; }
  int output_tid() const { return out_tid_; }
  const SrvTensor& tensor(int tid) const { return tensors_[tid]; }
  const void* tensor_ptr(int tid) const { return arena_ + tensors_[tid].offset; }
  // parity tap: tensor `tid` in logical NHWC f32 (pad channels dropped)
  bool fetch_logical(int tid, std::vector<float>& host, int dims[4], hipStream_t s, std::string& err);
  void set_keep_all(bool on) { if (on != keep_all_) { keep_all_ = on; bound_n_ = -1; } }
  // CTC mode of the LAST linear (f32 logits), f16 build: it leaves per-row partials in the output tensor's slot instead of the logits
  // (GemmArgs::ctc_part); the caller folds them with srv::launch_ctc_reduce(tensor_ptr(output_tid()), rows, ctc_slots(), ctc_step(), ..)
  void set_ctc_partials(bool on) { if (on != ctc_) { ctc_ = on; bound_n_ = -1; } }
  bool ctc_partials() const { return ctc_ && half_ && ctc_slots_ > 0; }
  int ctc_slots() const { return ctc_slots_; }
  int ctc_step() const { return ctc_step_; }
  void enable_timing(bool on) { timing_ = on; }
  const std::map<std::string, KernelTiming>& timings() const { return timings_; }
  void reset_timings() { timings_.clear(); }
  void collect_timings();
  // algorithmic work of one run of the current binding (sum over launches)
  double flops() const { return flops_; }
  double bytes() const { return bytes_; }

 private:
  struct Stage { std::string kind, a0, a1, a2, a3; float f = 0; int tid = -1, up = 1; };
  struct Op {
    std::string kind;
    std::map<std::string, std::string> kv;
    std::vector<Stage> ep;
    std::vector<int> ins, ups;
    int geti(const char* k, int d = 0) const { auto it = kv.find(k); return it == kv.end() ? d : atoi(it->second.c_str()); }
    float getf(const char* k, float d = 0) const { auto it = kv.find(k); return it == kv.end() ? d : strtof(it->second.c_str(), nullptr); }
    // resolved at load
    int cin_s = 0;                 // stored input channels
    int npad = 0, ncols = 0;
    const void* wimg = nullptr;    // device weight image (T)
    size_t wimg_bytes = 0;
    const float *bias = nullptr, *scale = nullptr, *shift = nullptr, *p0 = nullptr, *p1 = nullptr;
    float fbias = 0;
    int act = 0, res_tid = -1, res_up = 0;
    // fc1 of an MLP behind an `ln` (f16 build): the image of diag(gamma) W1 and the vectors of MlpArgs::ln_* (srv_mlp.h)
    const void* wimg_ln = nullptr;
    size_t wimg_ln_bytes = 0;
    const float *ln_s = nullptr, *ln_c = nullptr;
    int korder = 0;                // GemmArgs::korder of this op's weight image
  };
  struct Launch {
    std::string name;
    double flops = 0, bytes = 0;
    std::function<bool(hipStream_t, std::string&)> fn;
  };
  bool parse(const char* text, std::string& err);
  bool prepare_op(Op& op, const WeightMap& w, std::string& err);
  bool prepare_ln_fold(Op& fc1, const Op& ln, const WeightMap& w, std::string& err);
  bool bind(int N, int H, int W, hipStream_t s, std::string& err);
  int tune(const srv::GemmArgs& a, const std::string& key, hipStream_t s);
  void* upload_bytes(const void* p, size_t n);
  const float* upload_f32(const std::vector<float>& v);

  std::vector<Op> ops_;
  int ntensors_ = 0, out_tid_ = -1;
  bool half_ = true, keep_all_ = false, timing_ = false, ctc_ = false;
  int ctc_slots_ = 0, ctc_step_ = 0;
  std::vector<void*> dev_allocs_;
  std::vector<SrvTensor> tensors_;
  std::vector<Launch> launches_;
  unsigned char* arena_ = nullptr;
  size_t arena_cap_ = 0;
  int bound_n_ = -1, bound_h_ = 0, bound_w_ = 0;
  const float* x_in_ = nullptr;
  std::map<std::string, int> tuned_;  // (op, shape) -> tile configuration
  double flops_ = 0, bytes_ = 0;
  std::map<std::string, KernelTiming> timings_;
  struct EvPair { hipEvent_t a, b; std::string name; double flops, bytes; };
  std::vector<EvPair> ev_pending_;
};

}  // namespace ocr
