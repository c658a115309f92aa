// Host-layer test in the shape of the reference's own integration test
// (/root/reference/tests/test_ocr_worker.cpp: ctor/idle :107-117, start/stop idempotence :119-150, real image
// twice :182-233, empty image -> success=false + error :235-260, three queued requests keep their ids :262-296,
// idle after completion :298-318, cls on/off :328-382), plus a two-worker pool.  Prints one line per check and
// the JSON of the first request (compared with the oracle by tests/test_host_layer.py).
#include <cstdio>
#include <fstream>
#include <iostream>

#include "paddle_ocr_hip.h"

using namespace PaddleOCR;

static int failures = 0;
#define CHECK(cond)                                              \
  do {                                                           \
    if (!(cond)) { ++failures; printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); } \
    else printf("ok   %s\n", #cond);                           \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: test_worker <model_root> <raw_bgr_file> <rows>x<cols>\n"); return 2; }
  const std::string model_root = argv[1];
  int rows = 0, cols = 0;
  sscanf(argv[3], "%dx%d", &rows, &cols);
  std::ifstream f(argv[2], std::ios::binary);
  std::vector<uint8_t> px((size_t)rows * cols * 3);
  f.read((char*)px.data(), px.size());
  ImageView img{px.data(), rows, cols, (size_t)cols * 3};

  if (argc > 4 && std::string(argv[4]) == "multi") {
    // two workers of one pool on two DIFFERENT devices from one process (gpu_worker_pool.cpp:12-16 maps every worker to
    // GPU 0; here worker i -> GPU i mod n): every per-device resource (kernel attributes, streams, arenas) must exist on
    // both.  Prints each reply's JSON; tests/test_host_layer.py compares all of them with the oracle.
    const int ngpu = ocr_rt_device_count();
    if (ngpu < 2) { printf("SKIP one device\n"); return 0; }
    GPUWorkerPool pool(model_root, 2);
    CHECK(pool.workerDevice(0) == 0 && pool.workerDevice(1) == 1);
    pool.start();
    std::vector<std::future<std::string>> fut;
    for (int id = 0; id < 8; ++id) fut.push_back(pool.submitRequest(std::make_shared<OCRRequest>(id, img)));
    for (int id = 0; id < 8; ++id) printf("POOLJSON %s\n", fut[id].get().c_str());
    pool.stop();
    printf("%s\n", failures ? "FAILED" : "ALL OK");
    return failures ? 1 : 0;
  }

  {
    OCRWorker w(7, model_root, true, 0);
    CHECK(w.isIdle());
    CHECK(w.getWorkerId() == 7);
    w.start();
    w.start();  // idempotent
    std::vector<std::future<std::string>> fut;
    for (int id = 100; id < 103; ++id) {
      auto req = std::make_shared<OCRRequest>(id, img);
      fut.push_back(req->result_promise.get_future());
      w.addRequest(req);
    }
    std::string first;
    for (int k = 0; k < 3; ++k) {
      const std::string js = fut[k].get();
      if (k == 0) first = js;
      char key[64];
      snprintf(key, sizeof key, "\"request_id\":%d", 100 + k);
      CHECK(js.find(key) != std::string::npos);
      CHECK(js.find("\"success\":true") != std::string::npos);
      CHECK(js.find("\"worker_id\":7") != std::string::npos);
      CHECK(js.find("\"words\":[") != std::string::npos);
    }
    printf("JSON %s\n", first.c_str());
    auto empty = std::make_shared<OCRRequest>(5, ImageView{});
    auto fe = empty->result_promise.get_future();
    w.addRequest(empty);
    const std::string je = fe.get();
    CHECK(je.find("\"success\":false") != std::string::npos);
    CHECK(je.find("\"error\":\"Empty image data provided\"") != std::string::npos);
    for (int spin = 0; spin < 1000 && !w.isIdle(); ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    CHECK(w.isIdle());
    w.stop();
    w.stop();
  }
  {
    OCRWorker wc(1, model_root, true, 0, true);  // cls on
    OCRRequest r(1, img);
    OCRResult res = wc.processRequest(r);
    CHECK(res.success && res.width == cols && res.height == rows && res.processing_time_ms > 0);
    printf("CLSWORDS %zu\n", res.words.size());
    OCRWorker wr(2, model_root, true, 0, true, true);  // cls on, perspective crops
    OCRResult rr = wr.processRequest(r);
    CHECK(rr.success && rr.words.size() == res.words.size());
  }
  {
    // Utility::GetRotateCropImage: an axis-aligned box is the plain crop; a tall one comes back turned
    Image im(img);
    Image c = Utility::GetRotateCropImage(img, {{5, 7}, {90, 7}, {90, 40}, {5, 40}});
    CHECK(c.rows == 33 && c.cols == 85);
    bool same = true;
    for (int y = 0; y < c.rows && same; ++y)
      same = memcmp(&c.pixels[(size_t)y * c.cols * 3], &im.pixels[((size_t)(y + 7) * cols + 5) * 3], (size_t)c.cols * 3) == 0;
    CHECK(same);
    Image t = Utility::GetRotateCropImage(img, {{5, 7}, {30, 7}, {30, 140}, {5, 140}});
    CHECK(t.rows == 25 && t.cols == 133);
    bool threw = false;
    try { Utility::GetRotateCropImage(img, {{5, 5}, {5, 5}, {5, 5}, {5, 5}}); } catch (const std::exception&) { threw = true; }
    CHECK(threw);
  }
  {
    bool threw = false;
    try { OCRWorker bad(0, "/nonexistent/models", true, 0); } catch (const std::exception& e) { threw = true; printf("ok   invalid model dir: %s\n", e.what()); }
    CHECK(threw);
  }
  {
    GPUWorkerPool pool(model_root, 2);
    pool.start();
    std::vector<std::future<std::string>> fut;
    for (int id = 0; id < 6; ++id) fut.push_back(pool.submitRequest(std::make_shared<OCRRequest>(id, img)));
    int okc = 0;
    for (int id = 0; id < 6; ++id) {
      const std::string js = fut[id].get();
      char key[64];
      snprintf(key, sizeof key, "\"request_id\":%d,", id);
      okc += js.find(key) != std::string::npos && js.find("\"success\":true") != std::string::npos;
    }
    CHECK(okc == 6);
    pool.stop();
  }
  printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}
