// Plan-driven network executor: owns device weights (in kernel-ready layouts), an activation
// arena, and the per-shape launch list.  One instance per model per handle (= per host thread /
// HIP stream), mirroring "one predictor per stage object" of the reference
// (/root/reference/include/paddle_ocr/ocr_det.h:101).
#pragma once
#include <hip/hip_runtime.h>

#include <array>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels_net.h"
#include "pd_format.h"

namespace ocr {

struct PlanStage {
  int kind = 0, act = 0;
  float p0 = 0, p1 = 0;
  std::string n0, n1, n2, n3;
  int tid = -1, up = 1;
};

struct PlanOp {
  enum Kind { CONV, DW, DECONV, LINEAR, SEFC, GAP, POOL, EW, CONCAT, LN, ATTN, SOFTMAX, OUTPUT };
  Kind kind;
  int in = -1, out = -1;
  std::vector<int> ins, ups;  // concat
  int cin = 0, cout = 0, c = 0, cr = 0;
  int kh = 1, kw = 1, sh = 1, sw = 1, ph = 0, pw = 0;
  bool pool_max = false;
  int heads = 0, hd = 0;
  float scale = 0, eps = 0, slope = 0, offset = 0;
  std::string w, w1, b1, w2, b2, g, b;
  std::vector<PlanStage> ep;
};

struct Plan {
  std::string name;
  int ntensors = 0;
  std::vector<PlanOp> ops;
};

bool parse_plan(const char* text, Plan& plan, std::string& err);

struct TensorDesc {
  int n = 0, h = 0, w = 0, c = 0, cs = 0;
  bool plain = false;  // logical channel order, cs == c
  bool f16 = false;    // precision "fp16": a C8I activation tensor stored as f16 (same element indices; its arena slot keeps
                       // the f32 size).  Plain tensors (input, maps, logits) and the per-image vectors (pools, SE gates) stay f32
  size_t offset = 0;   // floats into the arena
  // ragged batch (Net::run_ragged): the tensor's width level (-1: a uniform tensor, e.g. the per-line SE vectors) and
  // its pixel count h * sum of the lines' widths at that level; `w` is then the widest line
  int lvl = -1;
  long pix = 0;
  long pixels() const { return lvl >= 0 ? pix : (long)n * h * w; }
  size_t numel() const { return (size_t)pixels() * cs; }
};

struct KernelTiming {
  double ms = 0;
  long count = 0;
  double flops = 0;  // algorithmic FLOPs summed over the timed launches
  double bytes = 0;  // algorithmic bytes summed over the timed launches
};

class Net {
 public:
  ~Net();
  // half (precision "fp16"): the C8I activation tensors are stored as f16 and the matrix-core products
  // (dense convs, linears, transposed convs, the 1x1 half of the fused depthwise blocks) run as f16 instructions with f32
  // accumulation; every VALU chain (stem, depthwise taps, epilogues, SE, layer norm, attention, softmax, the fused DB head)
  // and every reduction computes in f32 on values converted up, and rounds once when it stores
  bool load(const char* plan_text, const WeightMap& weights, std::string& err, bool half = false);
  bool half() const { return half_; }
  // Binds shapes (re-planning arena + launches if they changed) and enqueues the network.
  // x: device f32 [N,H,W,3] plain NHWC (already normalised).
  bool run(const float* x, int N, int H, int W, hipStream_t s, std::string& err);
  // Ragged batch (the recognizer): N text lines of height H, line n of width widths[n]; x = the lines' [H][w][3]
  // blocks one after the other.  One launch list for all widths (kernels_net.h, RagLevel); every line's results are
  // those of a run of that line alone.  bind_ragged alone prepares the binding (so that the caller can ask for the
  // output widths before it fills x); run_ragged binds if needed.
  bool bind_ragged(int H, const int* widths, int N, std::string& err);
  bool run_ragged(const float* x, int H, const int* widths, int N, hipStream_t s, std::string& err);
  // Ragged batch of IMAGES (the detector on mixed sizes): image n is heights[n] x widths[n] (multiples of 32, as
  // ResizeImgType0 leaves them); x = the images' [h][w][3] blocks one after the other; the output (and every tensor)
  // holds the images one after the other likewise.  Production launch list only (keep_all 0 or 2).
  bool run_ragged_images(const float* x, const int* heights, const int* widths, int N, hipStream_t s, std::string& err);
  // first pixel of image n in the output tensor of the current ragged-images binding is the prefix sum of h*w
  // line widths of tensor `tid` under the current ragged binding (host copy), e.g. the CTC step counts of the output
  const std::vector<int>& ragged_widths(int tid) const;
  // can this many lines / pixels go into one ragged launch? (32-bit pixel indices, attention working set)
  static bool ragged_ok(int H, const int* widths, int N, std::string& why);
  const TensorDesc& tensor(int tid) const { return tensors_[tid]; }
  const float* tensor_ptr(int tid) const { return arena_ + tensors_[tid].offset; }
  int output_tid() const { return out_tid_; }
  int ntensors() const { return plan_.ntensors; }
  // Copies tensor `tid` to host in logical NHWC order (parity taps).
  bool fetch_logical(int tid, std::vector<float>& host, int dims[4], hipStream_t s, std::string& err);

  // --- det tail options: fused u8 threshold bitmap (null = none) ---
  void set_det_bitmap(uint8_t* bitmap, int ithresh) { det_bitmap_ = bitmap; det_ithresh_ = ithresh; invalidate(); }
  // --- rec/cls head options: where the row softmax leaves its results ---
  // The launches read these pointers when they are issued, so new sinks of the same kind keep every binding (a
  // recorded graph holds the pointers it was captured with and is re-recorded when they differ: Net::run); only
  // a change of WHICH outputs are wanted changes the launch list (fused head or not).
  void set_head_outputs(float* probs, int* amax, float* pmax) {
    const bool same_kind = (probs != nullptr) == (head_probs_ != nullptr) && (amax != nullptr) == (head_amax_ != nullptr) &&
                           (pmax != nullptr) == (head_pmax_ != nullptr);
    head_probs_ = probs; head_amax_ = amax; head_pmax_ = pmax;
    if (!same_kind) invalidate();
  }

  // parity debugging: give every tensor its own arena slot so intermediate taps stay valid.
  // 0 = production (liveness-reused arena, fusions on), 1 = every plan tensor materialised (no fusion),
  // 2 = production launch list (gate folding, depthwise->pointwise fusion) but no arena reuse: every tensor that
  //     exists in production can be fetched and compared
  void set_keep_all(int mode) { if (mode != keep_all_) { keep_all_ = mode; invalidate(); } }
  // does tensor `tid` exist in HBM under the current binding? (fused-away tensors do not)
  bool materialised(int tid) const { return cur_ && tid > 0 && tid < (int)cur_->exists.size() && cur_->exists[tid]; }

  // per-kernel-family timing with HIP events on the launch stream (bench / roofline)
  void enable_timing(bool on) { timing_ = on; }
  // events only around launches whose name contains `substr` (empty: every launch)
  void set_timing_filter(const std::string& substr) { timing_filter_ = substr; }
  const std::map<std::string, KernelTiming>& timings() const { return timings_; }
  void reset_timings() { timings_.clear(); }
  void collect_timings();  // after a stream sync: folds event pairs into timings_
  // binding-cache statistics: runs, runs that had to bind a new shape, runs replayed from a recorded hipGraph
  struct Stats { long runs = 0, binds = 0, graph_replays = 0; };
  const Stats& stats() const { return stats_; }

 private:
  struct Launch {
    std::string name;
    double flops = 0, bytes = 0;
    std::function<void(hipStream_t)> fn;
  };
  // One bound input shape: tensor shapes + arena offsets, the launch list, and (once the same input buffer has
  // been seen twice in a row) the hipGraph of that list.  Bindings are cached per (N, H, W): a mixed-size request
  // stream (BASELINE configs[2]) and the rec lanes' odd widths come back to shapes they have seen without
  // re-planning or re-recording.  Launches hold arena pointers: whatever moves the arena drops the cache.
  // device memory of the ragged bindings' line tables: handed back when a binding dies, freed with the network
  struct TablePool {
    std::vector<std::pair<int*, size_t>> free_list;
    ~TablePool() { for (auto& e : free_list) (void)g_free(e.first); }
  };
  struct Binding {
    int n = 0, h = 0, w = 0;
    // ragged batch: the lines' widths per level, the packed tables (per level: w[N], cw[N+1], and the prefix sums of
    // ceil(w / 16), ceil(w / 8), ceil(w / 4), N + 2 entries each) on the host and on the device
    std::vector<int> widths, heights;  // heights: a ragged batch of IMAGES (the detector on mixed sizes)
    std::vector<std::vector<int>> level_w;
    std::vector<int> rag_host;
    int* rag_dev = nullptr;
    size_t rag_cap = 0;
    bool rag_uploaded = false;
    std::shared_ptr<TablePool> pool;
    std::vector<TensorDesc> tensors;
    std::vector<char> exists;        // per plan tensor: written to HBM by this binding's launches
    std::vector<Launch> launches;
    hipGraphExec_t graph_exec = nullptr;
    const float* graph_x = nullptr;
    hipStream_t graph_stream = nullptr;
    const void* graph_head[3] = {nullptr, nullptr, nullptr};
    const float* last_x = nullptr;   // input of the previous run of this binding
    bool graph_failed = false;       // capture was refused once: plain launches from then on
    unsigned long stamp = 0;         // LRU
    ~Binding() {
      if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
      if (rag_dev && pool) pool->free_list.emplace_back(rag_dev, rag_cap);
      else if (rag_dev) (void)g_free(rag_dev);
    }
  };
  std::shared_ptr<TablePool> pool_ = std::make_shared<TablePool>();
  static constexpr size_t kMaxRaggedBindings = 32;
  size_t max_bindings_ = 512;  // a mixed-size stream revisits sizes: BASELINE configs[2] has ~400 distinct det shapes (OCR_NET_BINDINGS, read at load)
  void invalidate() { cache_.clear(); cur_ = nullptr; }
  bool bind(int N, int H, int W, std::string& err, const int* widths = nullptr, const int* heights = nullptr);
  bool run_bound(const float* x, hipStream_t s, std::string& err);
  static std::vector<int> shape_key(int N, int H, int W, const int* widths, const int* heights);
  bool build_epilogue(const PlanOp& op, Epilogue& ep, bool conv_path, std::string& err);
  const float* dev_vec(const std::string& key) const;
  float* upload(const std::string& key, const std::vector<float>& v);

  Plan plan_;
  std::map<std::string, float*> dev_;        // uploaded parameter images by key
  std::map<std::string, float> scalars_;
  WeightMap host_w_;                          // kept only for scalar lookups / shapes
  std::vector<TensorDesc> tensors_;           // shapes/offsets of the current binding (plain flags fixed at load)
  std::map<std::vector<int>, std::unique_ptr<Binding>> cache_;  // key: shape_key
  Binding* cur_ = nullptr;
  unsigned long clock_ = 0;
  float* arena_ = nullptr;
  size_t arena_cap_ = 0;
  float* gap_part_ = nullptr;
  size_t gap_part_cap_ = 0;
  float* head_part_ = nullptr;  // fused softmax head: [rows][groups] max | sum | idx
  size_t head_part_cap_ = 0;
  long fused_head_rows_ = -1;  // rows of the linear that was bound in OUT_HEAD mode (-1: none)
  int fused_head_groups_ = 0;
  const float* bound_x_ = nullptr;
  int out_tid_ = -1;
  uint8_t* det_bitmap_ = nullptr;
  int det_ithresh_ = 0;
  float* head_probs_ = nullptr;
  int* head_amax_ = nullptr;
  float* head_pmax_ = nullptr;
  bool timing_ = false;
  bool half_ = false;
  bool graphs_ = true;          // OCR_GRAPH=0 (read when the network is loaded): plain launches only
  // set by a launch closure whose launcher refused a shape that bind() had accepted (cannot happen by construction;
  // a service process must get an error reply out of it, not an abort): run_bound fails the run with it
  std::string launch_error_;
  Stats stats_;
  std::string timing_filter_;
  int keep_all_ = 0;
  std::map<std::string, KernelTiming> timings_;
  // hipGraph of a binding's launch list: recorded when a run finds the input pointer of the binding's previous
  // run (the first run went out plainly, so one-time per-device setup such as function attributes is behind us),
  // replayed while input, stream and head sinks stay the same (one graph launch instead of 60-75 kernel launches:
  // what a single request's latency is made of).  Not used while per-launch timing events are on.
  struct EvPair { hipEvent_t a, b; std::string name; double flops, bytes; };
  std::vector<EvPair> ev_pending_;
  std::vector<hipEvent_t> ev_pool_;
};

}  // namespace ocr
