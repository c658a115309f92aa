// GPUWorkerPool under a BASELINE configs[3]-shaped stream (round 6): `total` requests of mixed sizes
// (H, W uniform in 640..1280, the cfg3 generator's range; seeded pixel content with dark stroke rectangles on a light
// background) through a pool of `workers` workers in a closed loop of `clients` outstanding requests - worker i on logical
// device i mod n (OCR_DEVICE_MAP=0,0,0,0,0,0,0,0 gives a one-GPU lease the eight devices of a node: own attribute memos,
// arenas and streams per logical id, /root/reference/src/gpu_worker_pool.cpp:12-16,46-59).
//   pool_load <model_root> <workers> <total> <clients> <distinct images> [least]
// Prints one JSON object: throughput, requests per worker, the submit-depth histogram, and - when workers > 1 - whether every
// reply's words equal the ONE-worker pool's reply for the same image (a reply's worker_id / processing_time_ms are its own).
#include <chrono>
#include <cstdio>
#include <deque>
#include <random>

#include "paddle_ocr_hip.h"

using namespace PaddleOCR;

static std::string words_of(const std::string& js) {
  const size_t p = js.find("\"words\":");
  if (p == std::string::npos) return js;
  const size_t e = js.rfind(",\"worker_id\":");  // (the keys a reply owns: worker_id behind the words, processing_time_ms in front)
  return js.substr(p, e == std::string::npos || e < p ? std::string::npos : e - p);
}

struct Img { std::vector<uint8_t> px; int rows, cols; };

static Img make_image(int seed) {
  std::mt19937 rng(7000 + seed);
  auto U = [&](int lo, int hi) { return lo + (int)(rng() % (unsigned)(hi - lo + 1)); };
  Img im;
  im.rows = U(640, 1280);
  im.cols = U(640, 1280);
  im.px.resize((size_t)im.rows * im.cols * 3);
  for (auto& b : im.px) b = (uint8_t)(222 + (rng() & 15));
  const int k = U(4, 24);
  for (int t = 0; t < k; ++t) {
    const int w = U(80, 400), h = U(16, 48), x0 = U(0, im.cols - w - 1 > 0 ? im.cols - w - 1 : 0), y0 = U(0, im.rows - h - 1);
    const int period = U(6, 14);
    for (int y = y0; y < y0 + h && y < im.rows; ++y)
      for (int x = x0; x < x0 + w && x < im.cols; ++x)
        if (((x - x0) % period) < period / 2)
          for (int c = 0; c < 3; ++c) im.px[((size_t)y * im.cols + x) * 3 + c] = (uint8_t)(30 + (rng() & 31));
  }
  return im;
}

struct RunOut { double seconds; std::vector<std::string> words; std::vector<long> per_worker, hist; };

static RunOut run(const std::string& root, int workers, int total, int clients, const std::vector<Img>& imgs, bool least) {
  GPUWorkerPool pool(root, workers);
  if (least) pool.setDispatch(GPUWorkerPool::LeastQueued);
  pool.start();
  RunOut o;
  o.words.assign(imgs.size(), "");
  // warm every worker on every distinct size class once (bindings, graphs): not timed
  {
    std::vector<std::future<std::string>> f;
    for (int w = 0; w < workers * 2; ++w) {
      const Img& im = imgs[w % imgs.size()];
      f.push_back(pool.submitRequest(std::make_shared<OCRRequest>(-1 - w, ImageView{im.px.data(), im.rows, im.cols, (size_t)im.cols * 3})));
    }
    for (auto& x : f) x.get();
  }
  const std::vector<long> served0 = pool.requestsPerWorker(), hist0 = pool.submitDepthHistogram();
  const auto t0 = std::chrono::steady_clock::now();
  std::deque<std::pair<int, std::future<std::string>>> inflight;
  int next = 0, bad = 0;
  auto submit = [&] {
    const int k = next % (int)imgs.size();
    const Img& im = imgs[k];
    inflight.emplace_back(k, pool.submitRequest(std::make_shared<OCRRequest>(next, ImageView{im.px.data(), im.rows, im.cols, (size_t)im.cols * 3})));
    ++next;
  };
  while (next < total && (int)inflight.size() < clients) submit();
  while (!inflight.empty()) {
    auto fr = std::move(inflight.front());
    inflight.pop_front();
    const std::string js = fr.second.get();
    if (js.find("\"success\":true") == std::string::npos) { if (!bad++) fprintf(stderr, "failed request: %.400s\n", js.c_str()); }
    const std::string w = words_of(js);
    if (o.words[fr.first].empty()) o.words[fr.first] = w;
    else if (o.words[fr.first] != w) {  // the same image must get the same words every time, whoever serves it
      if (!bad++) {
        const std::string& a = o.words[fr.first];
        size_t d = 0;
        while (d < a.size() && d < w.size() && a[d] == w[d]) ++d;
        const size_t from = d > 120 ? d - 120 : 0;
        fprintf(stderr, "image %d (%dx%d): two replies differ at byte %zu of %zu / %zu\n  %.300s\n  %.300s\n", fr.first, imgs[fr.first].rows,
                imgs[fr.first].cols, d, a.size(), w.size(), a.c_str() + from, w.c_str() + from);
      }
    }
    if (next < total) submit();
  }
  o.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  o.per_worker = pool.requestsPerWorker();
  o.hist = pool.submitDepthHistogram();
  for (size_t i = 0; i < o.per_worker.size(); ++i) o.per_worker[i] -= served0[i];
  for (size_t i = 0; i < o.hist.size(); ++i) o.hist[i] -= hist0[i];
  pool.stop();
  if (bad) o.words.clear();
  return o;
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: pool_load <model_root> <workers> <total> <clients> <distinct> [least]\n"); return 2; }
  const std::string root = argv[1];
  const int workers = atoi(argv[2]), total = atoi(argv[3]), clients = atoi(argv[4]), distinct = atoi(argv[5]);
  const bool least = argc > 6 && std::string(argv[6]) == "least";
  std::vector<Img> imgs;
  for (int i = 0; i < distinct; ++i) imgs.push_back(make_image(i));
  RunOut many = run(root, workers, total, clients, imgs, least);
  if (many.words.empty()) { printf("{\"error\":\"a request failed or an image got two different replies\"}\n"); return 1; }
  // the one-worker pool on the same images: distinct requests only (its rate is measured over those)
  RunOut one = run(root, 1, distinct * 2 > total ? total : distinct * 2, clients, imgs, false);
  int differ = 0;
  for (size_t i = 0; i < imgs.size(); ++i)
    if (!one.words.empty() && !many.words[i].empty() && !one.words[i].empty() && one.words[i] != many.words[i]) ++differ;
  printf("{\"workers\":%d,\"dispatch\":\"%s\",\"requests\":%d,\"clients\":%d,\"distinct_images\":%d,\"seconds\":%.3f,\"requests_per_s\":%.2f,", workers,
         least ? "least_queued" : "idle_first_round_robin", total, clients, distinct, many.seconds, total / many.seconds);
  printf("\"one_worker_requests_per_s\":%.2f,\"replies_differing_from_one_worker\":%d,\"requests_per_worker\":[", (double)(distinct * 2 > total ? total : distinct * 2) / one.seconds, one.words.empty() ? -1 : differ);
  for (size_t i = 0; i < many.per_worker.size(); ++i) printf("%s%ld", i ? "," : "", many.per_worker[i]);
  printf("],\"submit_depth_histogram\":[");
  for (size_t i = 0; i < many.hist.size(); ++i) printf("%s%ld", i ? "," : "", many.hist[i]);
  printf("]}\n");
  return differ || one.words.empty() ? 1 : 0;
}
