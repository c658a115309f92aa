// Fused det -> crop -> [cls -> rotate] -> rec pipeline over device-resident images
// (OCRWorker::processRequest, /root/reference/src/ocr_worker.cpp:213-311) and its C-ABI.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <thread>

#include "capi_common.h"
#include "crop.h"
#include "jpeg_stage.h"
#include "stages.h"

using namespace ocr;

namespace {
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

static int emit(const std::vector<std::vector<ocr_word>>& W, const std::vector<std::vector<int32_t>>& I, const std::vector<int>& order,
                ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids);

// One staged batch: host images copied into pinned memory and sent to the device on the copy stream, laid out size
// group by size group (images of one size are contiguous: one det pass each).  Two slots = double buffering: batch
// k+1 is staged (by another host thread) while batch k runs - SURVEY.md section 8e's "double-buffered pinned staging".
struct StageSlot {
  struct Img { int rows, cols, orig; size_t off, prob_off; };
  struct Group { int rows, cols, first, count; size_t off, prob_off; };
  uint8_t* pinned = nullptr;
  size_t pinned_cap = 0;
  DevBuf<uint8_t> dev;
  DevBuf<float> probs;
  std::vector<Img> imgs;      // layout order
  std::vector<Group> groups;
  size_t bytes = 0, prob_floats = 0;
  bool has_probs = false, staged = false;
  hipEvent_t ready = nullptr;
  ~StageSlot() {
    if (pinned) (void)g_host_free(pinned);
    if (ready) (void)hipEventDestroy(ready);
  }
};

// In-place 180-degree rotations of ROIs in request order (the cls stage: cv::rotate on views that alias their image,
// /root/reference/src/ocr_worker.cpp:255-262).  Only ROIs that intersect constrain each other: the ROIs of a view
// (img pointer + stride identify the image; different views are different images or separate crop buffers and do not
// alias) are grouped into the connected components of "intersects"; a component keeps request order inside one
// workgroup, components run concurrently, ONE launch for the batch.
static int rotate180_in_order(const std::vector<RotDesc>& rd, DevBuf<RotDesc>& scratch, DevBuf<int>& seg_scratch, hipStream_t stream,
                              std::string& err) {
  if (rd.empty()) return OCR_OK;
  const int n = (int)rd.size();
  std::vector<int> parent(n);
  for (int i = 0; i < n; ++i) parent[i] = i;
  auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
  std::map<std::pair<const uint8_t*, size_t>, std::vector<int>> views;  // per view the list is short (the lines of one image)
  for (int k = 0; k < n; ++k) {
    const RotDesc& q = rd[k];
    std::vector<int>& mine = views[{q.img, q.stride}];
    for (int j : mine) {
      const RotDesc& e = rd[j];
      if (e.x < q.x + q.w && q.x < e.x + e.w && e.y < q.y + q.h && q.y < e.y + e.h) {
        const int a = find(j), b = find(k);
        if (a != b) parent[std::max(a, b)] = std::min(a, b);  // the root is the component's first ROI
      }
    }
    mine.push_back(k);
  }
  // components in order of their first ROI, members in request order
  std::vector<std::vector<int>> members(n);
  for (int k = 0; k < n; ++k) members[find(k)].push_back(k);
  std::vector<RotDesc> sorted;
  std::vector<int> seg(1, 0);
  sorted.reserve(n);
  for (int r = 0; r < n; ++r) {
    if (members[r].empty()) continue;
    for (int k : members[r]) sorted.push_back(rd[k]);
    seg.push_back((int)sorted.size());
  }
  if (!scratch.ensure(sorted.size(), err) || !seg_scratch.ensure(seg.size(), err)) return OCR_ERR_DEVICE;
  if (hipMemcpyAsync(scratch.p, sorted.data(), sorted.size() * sizeof(RotDesc), hipMemcpyHostToDevice, stream) != hipSuccess ||
      hipMemcpyAsync(seg_scratch.p, seg.data(), seg.size() * sizeof(int), hipMemcpyHostToDevice, stream) != hipSuccess) {
    err = "rotation list upload failed";
    return OCR_ERR_DEVICE;
  }
  launch_rotate180_groups(scratch.p, seg_scratch.p, (int)seg.size() - 1, stream);
  if (hipGetLastError() != hipSuccess) { err = "rotation launch failed"; return OCR_ERR_DEVICE; }
  return OCR_OK;
}

// One chain det -> crops -> [cls -> rotate] -> rec (OCRWorker::processRequest) with its own stage objects, streams and
// buffers.  A pipeline owns two of them (ocr_pipe below): the chain of a batch is a sequence of dependent phases, some
// of them latency-bound (det post-processing, cls, the tails of the networks), and two chains side by side on two
// halves of the batch fill each other's gaps.
struct PipeWorker {
  DetStage det;
  // Further detector instances (own stream, network, buffers) for batches of MIXED sizes: every distinct size is its own
  // det pass (a few images at most), a latency-bound chain of ~90 small launches that leaves the chip idle - several
  // chains side by side, each driven by its own host thread, fill it.  OCR_DET_LANES (default 8) instances in all.
  std::vector<std::unique_ptr<DetStage>> det_extra;
  DetConfig det_cfg;
  int det_lanes = 8;
  bool det_ragged = true;  // OCR_DET_RAGGED=0: mixed-size batches as one detector pass per distinct size (rounds 1-2)
  bool det_post_ragged = true;  // OCR_DET_RAGGED=net: ragged network, BoxesFromBitmap per size group on the lanes
  RecStage rec;
  std::unique_ptr<ClsStage> cls;
  DevBuf<RotDesc> rot_desc;
  DevBuf<int> rot_seg;
  int crop_mode = 0;  // OCR_CROP_BOUNDING_RECT | OCR_CROP_ROTATE
  DevBuf<uint8_t> crop_arena;
  DevBuf<WarpDesc> warp_desc;
  std::vector<int32_t> boxes;  // [image in layout order][cap][8]
  std::vector<int> nbox;
  static constexpr int kCap = 1000;  // max_candidates bounds the boxes of one image (postprocess_op.cpp:260)

  int create(const DetConfig& d, const RecConfig& r, const ClsConfig* k, int crop, std::string& err) {
    int code = 0;
    if (!det.create(d, err, code)) return code ? code : OCR_ERR_DEVICE;
    det_cfg = d;
    if (!rec.create(r, err, code)) return code ? code : OCR_ERR_DEVICE;
    rec.want_taps = false;
    if (k) {
      cls.reset(new ClsStage());
      if (!cls->create(*k, err, code)) return code ? code : OCR_ERR_DEVICE;
    }
    crop_mode = crop;
    return OCR_OK;
  }

  // a further detector instance; it takes over the timing state of the pipeline (ocr_pipe_timing[_filter]) so that the
  // per-kernel survey sees the chunks that run on it
  bool timing_on = false;
  std::string timing_filter;
  int add_det_extra(std::string& err) {
    std::unique_ptr<DetStage> d(new DetStage());
    int code = 0;
    if (!d->create(det_cfg, err, code)) return code ? code : OCR_ERR_DEVICE;
    d->net().set_timing_filter(timing_filter);
    d->net().enable_timing(timing_on);
    det_extra.push_back(std::move(d));
    return OCR_OK;
  }

  // ---- run: det per size group, then ONE cls pass and ONE rec pass over the crops of every image of the batch
  // (results are batch-invariant, so pooling lines across sizes changes nothing but the launch count)
  int run_images(uint8_t* base, const std::vector<StageSlot::Img>& imgs, const std::vector<StageSlot::Group>& groups, const float* probs,
                 std::vector<std::vector<ocr_word>>& out_words, std::vector<std::vector<int32_t>>& out_ids, double times[3], std::string& err) {
    const int count = (int)imgs.size();
    boxes.resize((size_t)count * kCap * 8);
    nbox.assign(count, 0);
    double t0 = now_ms();
    auto run_group = [&](DetStage& d, const StageSlot::Group& g, std::string& e) {
      const size_t row = (size_t)g.cols * 3, img_bytes = row * g.rows;
      return d.run_device(base + g.off, img_bytes, row, g.rows, g.cols, g.count, boxes.data() + (size_t)g.first * kCap * 8, kCap,
                          nbox.data() + g.first, nullptr, e, probs ? probs + g.prob_off : nullptr);
    };
    const int lanes = (int)std::min<size_t>((size_t)det_lanes, groups.size());
    if (groups.size() > 1 && det_ragged) {
      // MIXED sizes (round 3): ONE detector network pass over all size groups (every image keeps its own size inside a
      // ragged launch: Net::run_ragged_images) instead of a latency-bound pass per distinct size; only BoxesFromBitmap
      // still runs per group, dealt over the detector lanes.  Chunks bound the activation arena (~64 Mpixel of
      // detector input per launch, what a uniform batch of 64 x 960 x 960 needs).
      std::vector<DetStage::MixedGroup> mg(groups.size());
      std::vector<size_t> gpx(groups.size());
      for (size_t gi = 0; gi < groups.size(); ++gi) {
        const auto& g = groups[gi];
        mg[gi] = DetStage::MixedGroup{g.rows, g.cols, g.count, g.off, g.prob_off};
        int rh = 0, rw = 0;
        { float a, b; DetStage::resize_shape(g.rows, g.cols, det.cfg().limit_type, det.cfg().limit_side_len, rh, rw, a, b); }
        gpx[gi] = (size_t)g.count * rh * rw;
      }
      static const size_t budget = [] { const char* e = getenv("OCR_DET_CHUNK_MP"); return (size_t)(e && atoi(e) > 0 ? atoi(e) : 64) << 20; }();
      // chunk k+1's network pass is enqueued (on the other of two detector instances: own arena, own maps) BEFORE the
      // host threads go through chunk k's per-group post-processing, so that the latency-bound post of one chunk runs
      // under the dense kernels of the next
      std::vector<std::pair<size_t, size_t>> chunks;
      for (size_t g0 = 0; g0 < groups.size();) {
        size_t g1 = g0, px = 0;
        while (g1 < groups.size() && (g1 == g0 || px + gpx[g1] <= budget)) px += gpx[g1++];
        chunks.emplace_back(g0, g1);
        g0 = g1;
      }
      // Two network instances whenever there is more than one chunk - whatever OCR_DET_LANES says (the lanes only deal
      // the per-group post-processing of the OCR_DET_RAGGED=net / dilation branch): chunk k+1's network is enqueued
      // before chunk k's maps are post-processed, so it must not run on the instance that holds them.
      const int want_extra = std::max(lanes - 1, chunks.size() > 1 ? 1 : 0);
      while ((int)det_extra.size() < want_extra) {
        int rc = add_det_extra(err);
        if (rc) return rc;
      }
      const int ninst = 1 + (int)det_extra.size();
      auto net_stage = [&](size_t k) -> DetStage& { return (k & 1) && ninst > 1 ? *det_extra[0] : det; };
      auto start = [&](size_t k) {
        return net_stage(k).mixed_net(base, mg.data() + chunks[k].first, (int)(chunks[k].second - chunks[k].first), probs, err);
      };
      int rc = start(0);
      if (rc) return rc;
      for (size_t k = 0; k < chunks.size(); ++k) {
        const size_t g0 = chunks[k].first, g1 = chunks[k].second;
        DetStage& src = net_stage(k);
        const bool more = k + 1 < chunks.size();
        if (more) {
          // the next chunk's stage must have finished ITS previous chunk's post work (it was a lane two chunks ago: joined)
          rc = start(k + 1);
          if (rc) return rc;
        }
        if (!det.cfg().use_dilation && det_post_ragged) {
          // every image of the chunk through ONE pass of the post-processing kernels (each takes its image's own map
          // size), on the stage that ran the chunk's network; it ends with a stream synchronisation
          const auto& gf = groups[g0];
          rc = src.post_mixed(mg.data() + g0, (int)(g1 - g0), boxes.data() + (size_t)gf.first * kCap * 8, kCap, nbox.data() + gf.first, err);
          if (rc) return rc;
          src.collect_timings();
          continue;
        }
        DetStage* busy = more ? &net_stage(k + 1) : nullptr;   // its stream is taken by the next chunk's network
        std::vector<DetStage*> post;
        for (int l = 0; l < ninst; ++l) {
          DetStage* d = l == 0 ? &det : det_extra[l - 1].get();
          if (d != busy) post.push_back(d);
        }
        const int np = (int)post.size();
        std::vector<int> rcs(np, OCR_OK);
        std::vector<std::string> errs(np);
        auto lane_fn = [&](int l) {
          DetStage& d = *post[l];
          for (size_t gi = g0 + l; gi < g1; gi += np) {
            const auto& g = groups[gi];
            rcs[l] = d.post_group(src.mixed_prob((int)(gi - g0)), src.mixed_bitmap((int)(gi - g0)), mg[gi], &d == &src ? nullptr : src.done(),
                                  boxes.data() + (size_t)g.first * kCap * 8, kCap, nbox.data() + g.first, errs[l]);
            if (rcs[l]) return;
          }
        };
        std::vector<std::thread> th;
        for (int l = 1; l < np; ++l) th.emplace_back(lane_fn, l);
        lane_fn(0);
        for (auto& t : th) t.join();
        for (int l = 0; l < np; ++l)
          if (rcs[l]) { err = errs[l]; return rcs[l]; }
        // this chunk's maps are free again once its network stage is idle (its own post lane, if any, has synchronised;
        // when it was not a post lane the network pass itself must be over before the stage takes chunk k+2)
        if (g_stream_sync(src.stream()) != hipSuccess) { err = "det stream sync failed"; return OCR_ERR_DEVICE; }
        src.collect_timings();
      }
    } else if (lanes <= 1) {
      for (const auto& g : groups) {
        int rc = run_group(det, g, err);
        if (rc) return rc;
      }
    } else {
      while ((int)det_extra.size() < lanes - 1) {   // created on first use: single-size callers never pay for them
        int rc = add_det_extra(err);
        if (rc) return rc;
      }
      // the clone above was enqueued on lane 0's stream: the other lanes' streams must not read it earlier
      if (g_stream_sync(det.stream()) != hipSuccess) { err = "det stream sync failed"; return OCR_ERR_DEVICE; }
      std::vector<int> rcs(lanes, OCR_OK);
      std::vector<std::string> errs(lanes);
      auto lane_fn = [&](int l) {
        DetStage& d = l == 0 ? det : *det_extra[l - 1];
        for (size_t gi = l; gi < groups.size(); gi += lanes) {
          rcs[l] = run_group(d, groups[gi], errs[l]);
          if (rcs[l]) return;
        }
      };
      std::vector<std::thread> th;
      for (int l = 1; l < lanes; ++l) th.emplace_back(lane_fn, l);
      lane_fn(0);
      for (auto& t : th) t.join();
      for (int l = 0; l < lanes; ++l)
        if (rcs[l]) { err = errs[l]; return rcs[l]; }
    }
    double t1 = now_ms();
    times[0] += t1 - t0;
    // crop rectangles: cv::boundingRect(points) & image rect (ocr_worker.cpp:245-258)
    std::vector<LineSrc> lines;
    std::vector<int> seg(1, 0);
    std::vector<int> line_box;  // box index (within its image) of every line
    if (crop_mode == OCR_CROP_ROTATE) {
      int rc = rotate_crops(base, imgs, lines, seg, line_box, err);
      if (rc) return rc;
    } else
    for (int i = 0; i < count; ++i) {
      const int rows = imgs[i].rows, cols = imgs[i].cols;
      const size_t row = (size_t)cols * 3;
      for (int j = 0; j < nbox[i]; ++j) {
        const int32_t* b = &boxes[((size_t)i * kCap + j) * 8];
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
          lines.push_back(LineSrc{base + imgs[i].off, row, ix0, iy0, ix1 - ix0, iy1 - iy0});
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
      int rc = cls->run_lines(lines, labels.data(), scores.data(), err);
      if (rc) return rc;
      // rotate in request order, in place on the device copy (cv::rotate on ROI views aliasing the image)
      std::vector<RotDesc> rd;
      for (size_t k = 0; k < lines.size(); ++k)
        if (labels[k] == 1) rd.push_back(RotDesc{const_cast<uint8_t*>(lines[k].img), lines[k].stride, lines[k].x, lines[k].y, lines[k].w, lines[k].h});
      int rrc = rotate180_in_order(rd, rot_desc, rot_seg, cls->stream(), err);
      if (rrc) return rrc;
      if (g_stream_sync(cls->stream()) != hipSuccess) { err = "cls stream sync failed"; return OCR_ERR_DEVICE; }
    }
    double t2 = now_ms();
    times[1] += t2 - t1;
    const int max_len = 256;
    std::vector<int32_t> ids(lines.size() * max_len);
    std::vector<int> lens(lines.size());
    std::vector<float> scores(lines.size());
    int rc = rec.run_lines(lines, seg, ids.data(), max_len, lens.data(), scores.data(), err);
    if (rc) return rc;
    times[2] += now_ms() - t2;
    for (int i = 0; i < count; ++i) {
      for (int k = seg[i]; k < seg[i + 1]; ++k) {
        ocr_word w;
        memcpy(w.box, &boxes[((size_t)i * kCap + line_box[k]) * 8], sizeof(w.box));
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
  int rotate_crops(const uint8_t* base, const std::vector<StageSlot::Img>& imgs, std::vector<LineSrc>& lines, std::vector<int>& seg,
                   std::vector<int>& line_box, std::string& err) {
    std::vector<WarpDesc> wd;
    std::vector<size_t> off;
    size_t total = 0;
    int max_px = 0;
    for (int i = 0; i < (int)imgs.size(); ++i) {
      const int rows = imgs[i].rows, cols = imgs[i].cols;
      const size_t row = (size_t)cols * 3;
      for (int j = 0; j < nbox[i]; ++j) {
        CropPlan p;
        if (!plan_rotate_crop(rows, cols, &boxes[((size_t)i * kCap + j) * 8], p)) continue;
        WarpDesc d;
        d.src = base + imgs[i].off + (size_t)p.top * row + (size_t)p.left * 3;
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
    if (g_stream_sync(det.stream()) != hipSuccess) { err = "crop kernel failed"; return OCR_ERR_DEVICE; }
    return OCR_OK;
  }
};

struct ocr_pipe {
  PipeWorker w0;
  // The second chain (OCR_PIPE_PHASES / ocr_pipe_cfg.phases = 2, the default): a batch of two or more images is cut in
  // two parts that run concurrently, each on its own worker and host thread.  Results are per image and do not depend
  // on what else is in a batch, so nothing changes but the overlap.  phases = 1 keeps one chain (one kernel at a time
  // owns the chip: what the per-kernel roofline figures of bench.py are measured with).
  std::vector<std::unique_ptr<PipeWorker>> extra;  // chains 2 .. phases
  int phases = 2;
  int parts_per_chain = 1;
  int device = 0;
  StageSlot slots[2];
  hipStream_t copy_stream = nullptr;
  JpegScratch jpeg;
  DevBuf<uint8_t> work;  // the requests' clones (OCRRequest copies the Mat, ocr_worker.h:28-29): cls rotates in place on them
  // Two batches in flight (round 5, ocr_pipe_run_device_on / ocr_pipe_run_staged_on): a call that names a chain runs its WHOLE
  // batch on that chain's worker, with a clone buffer of the chain's own; calls on different chains may run concurrently from
  // different host threads - chain 0 on batch k while chain 1 is on batch k+1 - so that consecutive batches overlap instead of
  // the two halves of one (no join per batch: what two free-running handles gained over one, without the second handle)
  DevBuf<uint8_t> work_chain[4];
  PipeWorker* worker(int c) { return c == 0 ? &w0 : (c >= 1 && c <= (int)extra.size() ? extra[c - 1].get() : nullptr); }

  ~ocr_pipe() { if (copy_stream) (void)hipStreamDestroy(copy_stream); }

  // ---- stage: host images -> pinned -> device (asynchronous after the host copies)
  // layout of a batch in a slot: stable order by (rows, cols), images of one size contiguous
  int layout(StageSlot& S, int count, const std::function<void(int, int&, int&)>& size_of, std::string& err) {
    if (!copy_stream && hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess) { err = "hipStreamCreate failed"; return OCR_ERR_DEVICE; }
    if (!S.ready && hipEventCreateWithFlags(&S.ready, hipEventDisableTiming) != hipSuccess) { err = "hipEventCreate failed"; return OCR_ERR_DEVICE; }
    if (S.staged && hipEventSynchronize(S.ready) != hipSuccess) { err = "staging event failed"; return OCR_ERR_DEVICE; }  // pinned buffer free again
    // from here to the end of stage() / stage_jpeg() the slot holds a half-written batch: a failure on the way must not
    // leave it runnable (a later ocr_pipe_run_staged would run the new layout over stale pixels)
    S.staged = false;
    std::vector<int> order(count), rows(count), cols(count);
    for (int i = 0; i < count; ++i) { order[i] = i; size_of(i, rows[i], cols[i]); }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rows[a] != rows[b] ? rows[a] < rows[b] : cols[a] < cols[b]; });
    const std::vector<StageSlot::Img> before = S.imgs;
    S.imgs.clear();
    S.groups.clear();
    size_t off = 0, poff = 0;
    for (int k = 0; k < count; ++k) {
      const int r = rows[order[k]], c = cols[order[k]];
      if (S.groups.empty() || S.groups.back().rows != r || S.groups.back().cols != c) {
        off = (off + 255) & ~(size_t)255;
        S.groups.push_back({r, c, k, 0, off, poff});
      }
      S.groups.back().count++;
      int rh = 0, rw = 0;
      { float a, b; DetStage::resize_shape(r, c, w0.det.cfg().limit_type, w0.det.cfg().limit_side_len, rh, rw, a, b); }
      S.imgs.push_back({r, c, order[k], off, poff});
      off += (size_t)r * c * 3;
      poff += (size_t)rh * rw;
    }
    S.bytes = off;
    S.prob_floats = poff;
    // probability maps attached to the slot (benchmark protocol) stay valid only while the layout is the same
    bool same = before.size() == S.imgs.size();
    for (size_t k = 0; same && k < before.size(); ++k)
      same = before[k].rows == S.imgs[k].rows && before[k].cols == S.imgs[k].cols && before[k].orig == S.imgs[k].orig;
    if (!same) S.has_probs = false;
    if (!S.dev.ensure(off + 256, err)) return OCR_ERR_DEVICE;
    return OCR_OK;
  }

  // ---- stage: host images -> pinned -> device (asynchronous after the host copies)
  int stage(int si, const ocr_img* imgs, int count, std::string& err) {
    StageSlot& S = slots[si];
    int rc = layout(S, count, [&](int i, int& r, int& c) { r = imgs[i].rows; c = imgs[i].cols; }, err);
    if (rc) return rc;
    const size_t off = S.bytes;
    if (off > S.pinned_cap) {
      if (S.pinned) (void)g_host_free(S.pinned);
      S.pinned = nullptr;
      S.pinned_cap = 0;
      if (g_host_malloc((void**)&S.pinned, off, hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; return OCR_ERR_DEVICE; }
      S.pinned_cap = off;
    }
    // host copies on a few threads (one thread moves ~10 GB/s: 64 images of 960x960 would take 18 ms)
    {
      const int nthreads = (int)std::min<size_t>(8, std::max<size_t>(1, off >> 22));
      auto copy_range = [&](int t) {
        for (int k = t; k < count; k += nthreads) {
          const ocr_img& im = imgs[S.imgs[k].orig];
          const size_t row = (size_t)im.cols * 3, stride = im.row_stride ? im.row_stride : row;
          uint8_t* dst = S.pinned + S.imgs[k].off;
          if (stride == row) memcpy(dst, im.data, row * im.rows);
          else for (int y = 0; y < im.rows; ++y) memcpy(dst + row * y, im.data + stride * y, row);
        }
      };
      std::vector<std::thread> th;
      for (int t = 1; t < nthreads; ++t) th.emplace_back(copy_range, t);
      copy_range(0);
      for (auto& t : th) t.join();
    }
    if (hipMemcpyAsync(S.dev.p, S.pinned, off, hipMemcpyHostToDevice, copy_stream) != hipSuccess) { err = "H2D copy failed"; return OCR_ERR_DEVICE; }
    if (hipEventRecord(S.ready, copy_stream) != hipSuccess) { err = "hipEventRecord failed"; return OCR_ERR_DEVICE; }
    S.staged = true;
    return OCR_OK;
  }

  // ---- stage JPEG coefficients: the pixel half of the decoder runs on the copy stream, into the slot
  int stage_jpeg(int si, const ocr_jpeg_img* imgs, int count, std::string& err) {
    StageSlot& S = slots[si];
    int rc = layout(S, count, [&](int i, int& r, int& c) { r = imgs[i].rows; c = imgs[i].cols; }, err);
    if (rc) return rc;
    std::vector<ocr_jpeg_img> ordered(count);
    std::vector<uint8_t*> dst(count);
    for (int k = 0; k < count; ++k) { ordered[k] = imgs[S.imgs[k].orig]; dst[k] = S.dev.p + S.imgs[k].off; }
    rc = jpeg_decode_async(ordered.data(), count, dst.data(), jpeg, copy_stream, err);
    if (rc) return rc;
    if (hipEventRecord(S.ready, copy_stream) != hipSuccess) { err = "hipEventRecord failed"; return OCR_ERR_DEVICE; }
    S.staged = true;
    return OCR_OK;
  }

  // benchmark protocol for synthetic weights: per staged image a probability map of the
  // detector's input resolution that replaces the network's map for thresholding / scoring
  int slot_probs(int si, const float* const* probs, int count, std::string& err) {
    StageSlot& S = slots[si];
    if (!S.staged || count != (int)S.imgs.size()) { err = "stage the slot's images first (same count)"; return OCR_ERR_ARG; }
    if (!S.probs.ensure(S.prob_floats + 64, err)) return OCR_ERR_DEVICE;
    for (int k = 0; k < count; ++k) {
      const size_t n = (k + 1 < count ? S.imgs[k + 1].prob_off : S.prob_floats) - S.imgs[k].prob_off;
      const hipError_t e = hipMemcpyAsync(S.probs.p + S.imgs[k].prob_off, probs[S.imgs[k].orig], n * sizeof(float), hipMemcpyDefault, copy_stream);
      if (e != hipSuccess) {
        err = std::string("probability map upload failed: ") + hipGetErrorString(e);
        return OCR_ERR_DEVICE;
      }
    }
    if (hipEventRecord(S.ready, copy_stream) != hipSuccess) { err = "hipEventRecord failed"; return OCR_ERR_DEVICE; }
    S.has_probs = true;
    return OCR_OK;
  }

  // ---- run: one chain, or several chains on as many parts of the batch
  int run_images(uint8_t* base, const std::vector<StageSlot::Img>& imgs, const std::vector<StageSlot::Group>& groups, const float* probs,
                 std::vector<std::vector<ocr_word>>& out_words, std::vector<std::vector<int32_t>>& out_ids, double times[3], std::string& err) {
    const int count = (int)imgs.size();
    const int nchains = (int)std::min<size_t>(1 + extra.size(), (size_t)count);
    if (nchains < 2) return w0.run_images(base, imgs, groups, probs, out_words, out_ids, times, err);
    // K parts, dealt round-robin to the chains (parts_per_chain > 1: a chain runs several smaller parts one after the
    // other, so that the chains drift out of phase and the tail of the call is a part, not half a batch)
    const int K = (int)std::min<size_t>((size_t)nchains * parts_per_chain, (size_t)count);
    // parts: size groups dealt to the lightest part (pixels), a group cut where that part reaches its share of the batch
    std::vector<std::vector<StageSlot::Img>> pi(K);
    std::vector<std::vector<StageSlot::Group>> pg(K);
    std::vector<std::vector<int>> gidx(K);  // layout index of a part's images
    std::vector<size_t> load(K, 0);
    auto give = [&](int p, const StageSlot::Group& g, int first, int n) {  // images first .. first+n-1 of group g
      int rh = 0, rw = 0;
      { float a, b; DetStage::resize_shape(g.rows, g.cols, w0.det.cfg().limit_type, w0.det.cfg().limit_side_len, rh, rw, a, b); }
      StageSlot::Group ng = g;
      ng.first = (int)pi[p].size();
      ng.count = n;
      ng.off = g.off + (size_t)first * g.rows * g.cols * 3;
      ng.prob_off = g.prob_off + (size_t)first * rh * rw;
      pg[p].push_back(ng);
      for (int k = 0; k < n; ++k) { pi[p].push_back(imgs[g.first + first + k]); gidx[p].push_back(g.first + first + k); }
      load[p] += (size_t)n * g.rows * g.cols;
    };
    size_t total = 0;
    for (const auto& g : groups) total += (size_t)g.count * g.rows * g.cols;
    const size_t share = (total + K - 1) / K;
    for (const auto& g : groups) {
      const size_t px = (size_t)g.rows * g.cols;
      int first = 0;
      while (first < g.count) {
        const int p = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        const int left = g.count - first;
        const size_t room = share > load[p] ? share - load[p] : 0;
        int n = left;
        if ((size_t)left * px > room + px / 2) n = (int)std::min<size_t>(left, std::max<size_t>(1, (room + px / 2) / px));
        give(p, g, first, n);
        first += n;
      }
    }
    // the clone was enqueued on worker 0's detector stream: the other workers' streams must not read it earlier
    if (g_stream_sync(w0.det.stream()) != hipSuccess) { err = "det stream sync failed"; return OCR_ERR_DEVICE; }
    std::vector<std::vector<std::vector<ocr_word>>> W(K);
    std::vector<std::vector<std::vector<int32_t>>> I(K);
    std::vector<std::array<double, 3>> t(K, std::array<double, 3>{0, 0, 0});
    std::vector<int> rc(K, OCR_OK);
    std::vector<std::string> errs(K);
    auto chain = [&](int c) {  // parts c, c + nchains, ... on worker c
      PipeWorker& w = c == 0 ? w0 : *extra[c - 1];
      for (int p = c; p < K; p += nchains) {
        if (pi[p].empty()) continue;
        rc[p] = w.run_images(base, pi[p], pg[p], probs, W[p], I[p], t[p].data(), errs[p]);
        if (rc[p]) return;
      }
    };
    std::vector<std::thread> th;
    // (round 5: starting chain c a few milliseconds late - 3 .. 24 ms - so that one chain's matrix-bound kernels meet the other's
    // HBM-bound ones costs 2-11 %: 1282 -> 1257 / 1218 / 1177 / 1155 / 1140 img/s; what two free-running handles gain comes from
    // batches overlapping batches, not from a phase offset inside one)
    for (int c = 1; c < nchains; ++c) th.emplace_back([&, c]() { (void)rt_set_device(device); chain(c); });
    chain(0);
    for (auto& x : th) x.join();
    for (int p = 0; p < K; ++p)
      if (rc[p]) { err = errs[p]; return rc[p]; }
    out_words.assign(count, {});
    out_ids.assign(count, {});
    for (int p = 0; p < K; ++p)
      for (size_t k = 0; k < gidx[p].size(); ++k) {
        out_words[gidx[p][k]] = std::move(W[p][k]);
        out_ids[gidx[p][k]] = std::move(I[p][k]);
      }
    for (int k = 0; k < 3; ++k) {  // the chains ran side by side: per stage, the busiest chain's time
      double m = 0;
      for (int c = 0; c < nchains; ++c) {
        double sum = 0;
        for (int p = c; p < K; p += nchains) sum += t[p][k];
        m = std::max(m, sum);
      }
      times[k] += m;
    }
    return OCR_OK;
  }

  // run a staged slot: wait for its upload, clone, run, hand the results back in the caller's order
  int run_slot(int si, ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3], int chain = -1) {
    StageSlot& S = slots[si];
    if (!S.staged || S.imgs.empty()) return fail(OCR_ERR_ARG, "nothing staged in this slot");
    std::string err;
    PipeWorker* one = chain >= 0 ? worker(chain) : nullptr;  // the whole batch on one chain (two batches in flight), or split over all
    if (chain >= 0 && !one) return fail(OCR_ERR_ARG, "no such chain (0 <= chain < phases)");
    PipeWorker& first = one ? *one : w0;
    DevBuf<uint8_t>& clone = one ? work_chain[chain] : work;
    if (hipStreamWaitEvent(first.det.stream(), S.ready, 0) != hipSuccess) return fail(OCR_ERR_DEVICE, "hipStreamWaitEvent failed");
    if (!clone.ensure(S.bytes + 256, err)) return fail(OCR_ERR_DEVICE, err);
    if (hipMemcpyAsync(clone.p, S.dev.p, S.bytes, hipMemcpyDeviceToDevice, first.det.stream()) != hipSuccess) return fail(OCR_ERR_DEVICE, "clone failed");
    double t[3] = {0, 0, 0};
    std::vector<std::vector<ocr_word>> W;
    std::vector<std::vector<int32_t>> I;
    const int rc = one ? one->run_images(clone.p, S.imgs, S.groups, S.has_probs ? S.probs.p : nullptr, W, I, t, err)
                       : run_images(clone.p, S.imgs, S.groups, S.has_probs ? S.probs.p : nullptr, W, I, t, err);
    if (rc) return fail(rc, err);
    if (times) { times[0] = t[0]; times[1] = t[1]; times[2] = t[2]; }
    const int count = (int)S.imgs.size();
    std::vector<int> order(count);  // order[original index] = layout index
    for (int k = 0; k < count; ++k) order[S.imgs[k].orig] = k;
    return emit(W, I, order, words, cap_words, word_off, nwords, ids, cap_ids);
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
  if (c->phases < 0 || c->phases > 4) return fail(OCR_ERR_ARG, "phases must be 0 (default) or 1..4");
  h->phases = c->phases ? c->phases : 2;
  if (const char* e = getenv("OCR_PIPE_PHASES")) h->phases = std::min(4, std::max(1, atoi(e)));
  std::string err;
  DetConfig d;
  d.model_dir = c->det.model_dir; d.device = c->det.device_id;
  if (c->det.limit_type) d.limit_type = c->det.limit_type;
  d.limit_side_len = c->det.limit_side_len; d.thresh = c->det.det_db_thresh; d.box_thresh = c->det.det_db_box_thresh;
  d.unclip_ratio = c->det.det_db_unclip_ratio;
  if (c->det.det_db_score_mode) d.score_mode = c->det.det_db_score_mode;
  d.use_dilation = c->det.use_dilation;
  if (c->det.precision) d.precision = c->det.precision;
  d.max_batch = c->det.max_batch > 0 ? c->det.max_batch : 1;
  d.cv_compat = resolve_cv_compat(c->det.cv_compat);
  RecConfig r;
  r.model_dir = c->rec.model_dir; r.label_path = c->rec.label_path; r.device = c->det.device_id;
  r.batch_num = c->rec.rec_batch_num; r.img_h = c->rec.rec_img_h; r.img_w = c->rec.rec_img_w; r.sort_mode = c->rec.sort_mode;
  if (c->rec.precision) r.precision = c->rec.precision;
  ClsConfig k;
  if (c->enable_cls) {
    k.model_dir = c->cls.model_dir; k.device = c->det.device_id; k.thresh = c->cls.cls_thresh;
    k.batch_num = c->cls.cls_batch_num > 0 ? c->cls.cls_batch_num : 1;
    if (c->cls.precision) k.precision = c->cls.precision;
  }
  int lanes = 8;
  if (const char* e = getenv("OCR_DET_LANES")) lanes = std::min(32, std::max(1, atoi(e)));
  int rc = h->w0.create(d, r, c->enable_cls ? &k : nullptr, c->crop_mode, err);
  if (rc) return fail(rc, err);
  h->w0.det_lanes = lanes;
  for (int p = 1; p < h->phases; ++p) {
    h->extra.emplace_back(new PipeWorker());
    rc = h->extra.back()->create(d, r, c->enable_cls ? &k : nullptr, c->crop_mode, err);
    if (rc) return fail(rc, err);
  }
  // the detector lanes (mixed-size batches) are dealt over the chains
  h->w0.det_lanes = std::max(1, (lanes + h->phases - 1) / h->phases);
  for (auto& w : h->extra) w->det_lanes = h->w0.det_lanes;
  if (const char* e = getenv("OCR_DET_RAGGED")) {
    h->w0.det_ragged = e[0] != '0';
    h->w0.det_post_ragged = e[0] != 'n';
    for (auto& w : h->extra) { w->det_ragged = h->w0.det_ragged; w->det_post_ragged = h->w0.det_post_ragged; }
  }
  // two idle high-priority streams per device (once per process), created after the first pipeline's stage streams and
  // before the detector lanes' (which come into being at the first mixed-size batch): the configuration in which the
  // lanes measured fastest (capi_net.hip)
  priority_anchor(h->device, 2);
  *out = h.release();
  return OCR_OK;
}
void ocr_pipe_destroy(ocr_pipe* h) { delete h; }

static int pipe_run_device(ocr_pipe* h, int chain, const void* dev_bgr, int rows, int cols, int count, const float* dev_prob, ocr_word* words,
                           int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3]) {
  if (!h || !dev_bgr || rows <= 0 || cols <= 0 || count < 1 || !words || !word_off || !nwords || !ids)
    return fail(OCR_ERR_ARG, "bad argument");
  CAPI_HIP(rt_set_device(h->device));
  PipeWorker* one = chain >= 0 ? h->worker(chain) : nullptr;
  if (chain >= 0 && !one) return fail(OCR_ERR_ARG, "no such chain (0 <= chain < phases)");
  PipeWorker& first = one ? *one : h->w0;
  DevBuf<uint8_t>& clone = one ? h->work_chain[chain] : h->work;
  double t[3] = {0, 0, 0};
  std::vector<std::vector<ocr_word>> W;
  std::vector<std::vector<int32_t>> I;
  std::string err;
  // the request's clone (OCRRequest copies the Mat, ocr_worker.h:28-29): cls rotation is in place on it
  const size_t img_bytes = (size_t)rows * cols * 3, bytes = img_bytes * count;
  if (!clone.ensure(bytes + 256, err)) return fail(OCR_ERR_DEVICE, err);
  CAPI_HIP(hipMemcpyAsync(clone.p, dev_bgr, bytes, hipMemcpyDeviceToDevice, first.det.stream()));
  std::vector<StageSlot::Img> imgs(count);
  for (int i = 0; i < count; ++i) imgs[i] = {rows, cols, i, img_bytes * i, 0};
  const std::vector<StageSlot::Group> groups = {{rows, cols, 0, count, 0, 0}};
  const int rc = one ? one->run_images(clone.p, imgs, groups, dev_prob, W, I, t, err) : h->run_images(clone.p, imgs, groups, dev_prob, W, I, t, err);
  if (rc) return fail(rc, err);
  if (times) { times[0] = t[0]; times[1] = t[1]; times[2] = t[2]; }
  std::vector<int> order(count);
  for (int i = 0; i < count; ++i) order[i] = i;
  return emit(W, I, order, words, cap_words, word_off, nwords, ids, cap_ids);
}
int ocr_pipe_run_device(ocr_pipe* h, const void* dev_bgr, int rows, int cols, int count, const float* dev_prob, ocr_word* words,
                        int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3]) {
  return pipe_run_device(h, -1, dev_bgr, rows, cols, count, dev_prob, words, cap_words, word_off, nwords, ids, cap_ids, times);
}
int ocr_pipe_run_device_on(ocr_pipe* h, int chain, const void* dev_bgr, int rows, int cols, int count, const float* dev_prob,
                           ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3]) {
  if (chain < 0) return fail(OCR_ERR_ARG, "no such chain (0 <= chain < phases)");
  return pipe_run_device(h, chain, dev_bgr, rows, cols, count, dev_prob, words, cap_words, word_off, nwords, ids, cap_ids, times);
}

int ocr_pipe_stage(ocr_pipe* h, int slot, const ocr_img* imgs, int count) {
  if (!h || !imgs || count < 1 || slot < 0 || slot > 1) return fail(OCR_ERR_ARG, "bad argument");
  for (int i = 0; i < count; ++i)
    if (!imgs[i].data || imgs[i].rows <= 0 || imgs[i].cols <= 0) return fail(OCR_ERR_ARG, "Empty image data provided");
  CAPI_HIP(rt_set_device(h->device));
  std::string err;
  const int rc = h->stage(slot, imgs, count, err);
  return rc ? fail(rc, err) : OCR_OK;
}

int ocr_pipe_stage_jpeg(ocr_pipe* h, int slot, const ocr_jpeg_img* imgs, int count) {
  if (!h || !imgs || count < 1 || slot < 0 || slot > 1) return fail(OCR_ERR_ARG, "bad argument");
  for (int i = 0; i < count; ++i)
    if (!jpeg_img_valid(imgs[i])) return fail(OCR_ERR_ARG, "bad JPEG coefficient descriptor");
  CAPI_HIP(rt_set_device(h->device));
  std::string err;
  const int rc = h->stage_jpeg(slot, imgs, count, err);
  return rc ? fail(rc, err) : OCR_OK;
}

int ocr_pipe_slot_probs(ocr_pipe* h, int slot, const float* const* probs, int count) {
  if (!h || !probs || count < 1 || slot < 0 || slot > 1) return fail(OCR_ERR_ARG, "bad argument");
  for (int i = 0; i < count; ++i) if (!probs[i]) return fail(OCR_ERR_ARG, "null probability map");
  CAPI_HIP(rt_set_device(h->device));
  std::string err;
  const int rc = h->slot_probs(slot, probs, count, err);
  return rc ? fail(rc, err) : OCR_OK;
}

int ocr_pipe_run_staged(ocr_pipe* h, int slot, ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids,
                        int cap_ids, double times[3]) {
  if (!h || slot < 0 || slot > 1 || !words || !word_off || !nwords || !ids) return fail(OCR_ERR_ARG, "bad argument");
  CAPI_HIP(rt_set_device(h->device));
  return h->run_slot(slot, words, cap_words, word_off, nwords, ids, cap_ids, times);
}

int ocr_pipe_run_staged_on(ocr_pipe* h, int chain, int slot, ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids,
                           int cap_ids, double times[3]) {
  if (!h || chain < 0 || slot < 0 || slot > 1 || !words || !word_off || !nwords || !ids) return fail(OCR_ERR_ARG, "bad argument");
  CAPI_HIP(rt_set_device(h->device));
  return h->run_slot(slot, words, cap_words, word_off, nwords, ids, cap_ids, times, chain);
}

int ocr_pipe_run(ocr_pipe* h, const ocr_img* imgs, int count, ocr_word* words, int cap_words, int* word_off, int* nwords,
                 int32_t* ids, int cap_ids, double times[3]) {
  if (!h || !imgs || count < 1 || !words || !word_off || !nwords || !ids) return fail(OCR_ERR_ARG, "bad argument");
  const int rc = ocr_pipe_stage(h, 0, imgs, count);
  if (rc) return rc;
  return h->run_slot(0, words, cap_words, word_off, nwords, ids, cap_ids, times);
}

const char* ocr_pipe_label(ocr_pipe* h, int id) {
  if (!h || id < 0 || id >= (int)h->w0.rec.labels().size()) return nullptr;
  return h->w0.rec.labels()[id].c_str();
}

int ocr_pipe_det_shape(ocr_pipe* h, int rows, int cols, int* net_rows, int* net_cols) {
  if (!h || !net_rows || !net_cols) return fail(OCR_ERR_ARG, "null argument");
  float a, b;
  DetStage::resize_shape(rows, cols, h->w0.det.cfg().limit_type, h->w0.det.cfg().limit_side_len, *net_rows, *net_cols, a, b);
  return OCR_OK;
}

static std::vector<Net*> pipe_nets(ocr_pipe* h) {
  std::vector<Net*> v;
  std::vector<PipeWorker*> ws{&h->w0};
  for (auto& w : h->extra) ws.push_back(w.get());
  for (PipeWorker* w : ws) {
    v.push_back(&w->det.net());
    for (auto& d : w->det_extra) v.push_back(&d->net());  // odd chunks of a mixed-size batch run here
    if (w->cls) v.push_back(&w->cls->net());
    v.push_back(&w->rec.net());
  }
  return v;
}
static std::vector<SrvNet*> pipe_srv_nets(ocr_pipe* h) {  // the server networks of a configs[4] pipeline
  std::vector<SrvNet*> v;
  std::vector<PipeWorker*> ws{&h->w0};
  for (auto& w : h->extra) ws.push_back(w.get());
  for (PipeWorker* w : ws) {
    if (w->det.srv()) v.push_back(w->det.srv());
    if (w->rec.srv()) v.push_back(w->rec.srv());
  }
  return v;
}
int ocr_pipe_timing(ocr_pipe* h, int enable) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  for (Net* net : pipe_nets(h)) {
    net->enable_timing(enable != 0);
    net->reset_timings();
  }
  for (SrvNet* net : pipe_srv_nets(h)) {
    net->enable_timing(enable != 0);
    net->reset_timings();
  }
  h->w0.timing_on = enable != 0;
  for (auto& w : h->extra) w->timing_on = enable != 0;
  return OCR_OK;
}
int ocr_pipe_stats(ocr_pipe* h, long long out[3]) {
  if (!h || !out) return fail(OCR_ERR_ARG, "null argument");
  out[0] = out[1] = out[2] = 0;
  auto add = [&](Net& n) { out[0] += n.stats().runs; out[1] += n.stats().binds; out[2] += n.stats().graph_replays; };
  std::vector<PipeWorker*> ws{&h->w0};
  for (auto& w : h->extra) ws.push_back(w.get());
  for (PipeWorker* w : ws) {
    add(w->det.net());
    for (auto& d : w->det_extra) add(d->net());
    if (w->cls) add(w->cls->net());
    add(w->rec.net());
  }
  return OCR_OK;
}
int ocr_pipe_timing_filter(ocr_pipe* h, const char* substr) {
  if (!h) return fail(OCR_ERR_ARG, "null handle");
  const std::string f = substr ? substr : "";
  for (Net* net : pipe_nets(h)) net->set_timing_filter(f);
  h->w0.timing_filter = f;
  for (auto& w : h->extra) w->timing_filter = f;
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
  {
    std::vector<PipeWorker*> ws{&h->w0};
    for (auto& w : h->extra) ws.push_back(w.get());
    for (PipeWorker* w : ws)
      for (int k = 0; k < 2; ++k) {
        SrvNet* net = k ? w->rec.srv() : w->det.srv();
        if (!net) continue;
        for (auto& kv : net->timings()) {  // (the mobile networks' names start with "det." / "rec.": the same convention)
          KernelTiming& t = all[std::string(k ? "rec." : "det.") + kv.first];
          t.ms += kv.second.ms; t.count += kv.second.count; t.flops += kv.second.flops; t.bytes += kv.second.bytes;
        }
      }
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
  CAPI_HIP(g_memcpy2d(dimg.p, row, bgr, stride, row, rows, hipMemcpyHostToDevice));
  CAPI_HIP(g_memcpy(ddesc.p, wd.data(), n * sizeof(WarpDesc), hipMemcpyHostToDevice));
  launch_warp_crops(ddesc.p, n, max_px, 0);
  CAPI_HIP(hipGetLastError());
  CAPI_HIP(g_memcpy(out, dout.p, total, hipMemcpyDeviceToHost));
  return OCR_OK;
}

int ocr_rotate180_rois(uint8_t* bgr, int rows, int cols, size_t row_stride, const int32_t* rects, int n) {
  if (!bgr || rows <= 0 || cols <= 0 || !rects || n < 1) return fail(OCR_ERR_ARG, "bad argument");
  const size_t row = (size_t)cols * 3, stride = row_stride ? row_stride : row;
  std::string err;
  DevBuf<uint8_t> dimg;
  DevBuf<RotDesc> ddesc;
  DevBuf<int> dseg;
  if (!dimg.ensure(row * rows, err)) return fail(OCR_ERR_DEVICE, err);
  std::vector<RotDesc> rd(n);
  for (int k = 0; k < n; ++k) {
    const int32_t* r = rects + 4 * k;
    if (r[0] < 0 || r[1] < 0 || r[2] < 1 || r[3] < 1 || r[0] + r[2] > cols || r[1] + r[3] > rows) return fail(OCR_ERR_ARG, "rectangle outside the image");
    rd[k] = RotDesc{dimg.p, row, r[0], r[1], r[2], r[3]};
  }
  CAPI_HIP(g_memcpy2d(dimg.p, row, bgr, stride, row, rows, hipMemcpyHostToDevice));
  const int rc = rotate180_in_order(rd, ddesc, dseg, 0, err);
  if (rc) return fail(rc, err);
  CAPI_HIP(g_memcpy2d(bgr, stride, dimg.p, row, row, rows, hipMemcpyDeviceToHost));
  return OCR_OK;
}

int ocr_dev_alloc(void** p, size_t bytes) {
  if (!p) return fail(OCR_ERR_ARG, "null argument");
  CAPI_HIP(g_malloc(p, bytes));
  return OCR_OK;
}
int ocr_dev_free(void* p) { CAPI_HIP(g_free(p)); return OCR_OK; }
int ocr_dev_upload(void* dst, const void* src, size_t bytes) { CAPI_HIP(g_memcpy(dst, src, bytes, hipMemcpyHostToDevice)); return OCR_OK; }
int ocr_dev_download(void* dst, const void* src, size_t bytes) { CAPI_HIP(g_memcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return OCR_OK; }
int ocr_dev_sync(void) { CAPI_HIP(hipDeviceSynchronize()); return OCR_OK; }

}  // extern "C"
