// Host side of the device JPEG back-end + its C-ABI test entry point.
#include <algorithm>
#include <cstring>
#include <thread>

#include "capi_common.h"
#include "jpeg_stage.h"

namespace ocr {

JpegScratch::~JpegScratch() {
  if (pinned) (void)g_host_free(pinned);
  if (copied) (void)hipEventDestroy(copied);
}

bool jpeg_img_valid(const ocr_jpeg_img& im) {
  if (im.rows <= 0 || im.cols <= 0 || (long)im.rows * im.cols > (64L << 20) || (im.ncomp != 1 && im.ncomp != 3)) return false;
  const bool s444 = im.hmax == 1 && im.vmax == 1, s422 = im.hmax == 2 && im.vmax == 1, s420 = im.hmax == 2 && im.vmax == 2;
  if (!(s444 || s422 || s420) || (im.ncomp == 1 && !s444)) return false;
  for (int c = 0; c < im.ncomp; ++c) {
    const ocr_jpeg_comp& k = im.comp[c];
    const int h = c == 0 ? im.hmax : 1, v = c == 0 ? im.vmax : 1;
    if (!k.coef || k.bw <= 0 || k.bh <= 0) return false;
    if (k.dw != (im.cols * h + im.hmax - 1) / im.hmax || k.dh != (im.rows * v + im.vmax - 1) / im.vmax) return false;
    if (k.bw * 8 < k.dw || k.bh * 8 < k.dh || k.bw > 16384 || k.bh > 16384) return false;  // the planes cover the component
  }
  return true;
}

int jpeg_decode_async(const ocr_jpeg_img* imgs, int count, uint8_t* const* dst, JpegScratch& sc, hipStream_t s, std::string& err) {
  std::vector<JpegPlaneDesc> pd;
  std::vector<JpegImageDesc> id((size_t)count);
  std::vector<size_t> coef_off, plane_off;
  size_t ncoef = 0, nplane = 0;
  long nblocks = 0, max_px = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_jpeg_img& im = imgs[i];
    if (!jpeg_img_valid(im)) { err = "bad JPEG coefficient descriptor"; return OCR_ERR_ARG; }
    for (int c = 0; c < im.ncomp; ++c) {
      const ocr_jpeg_comp& k = im.comp[c];
      JpegPlaneDesc d{};
      memcpy(d.quant, k.quant, sizeof d.quant);
      d.bw = k.bw; d.bh = k.bh; d.first_block = nblocks;
      pd.push_back(d);
      coef_off.push_back(ncoef);
      plane_off.push_back(nplane);
      ncoef += (size_t)k.bw * k.bh * 64;
      nplane += ((size_t)k.bw * 8 * k.bh * 8 + 255) & ~(size_t)255;
      nblocks += (long)k.bw * k.bh;
    }
    max_px = std::max(max_px, (long)im.rows * im.cols);
  }
  if (!sc.coef.ensure(ncoef + 64, err) || !sc.planes.ensure(nplane + 256, err) || !sc.pd.ensure(pd.size(), err) || !sc.id.ensure(count, err))
    return OCR_ERR_DEVICE;
  if (!sc.copied && hipEventCreateWithFlags(&sc.copied, hipEventDisableTiming) != hipSuccess) { err = "hipEventCreate failed"; return OCR_ERR_DEVICE; }
  if (sc.pinned && hipEventSynchronize(sc.copied) != hipSuccess) { err = "staging event failed"; return OCR_ERR_DEVICE; }
  if (ncoef > sc.pinned_cap) {
    if (sc.pinned) (void)g_host_free(sc.pinned);
    sc.pinned = nullptr;
    sc.pinned_cap = 0;
    if (g_host_malloc((void**)&sc.pinned, ncoef * sizeof(int16_t), hipHostMallocDefault) != hipSuccess) { err = "hipHostMalloc failed"; return OCR_ERR_DEVICE; }
    sc.pinned_cap = ncoef;
  }
  {  // coefficient arrays -> pinned memory, a few host threads
    struct Piece { const int16_t* src; size_t off, n; };
    std::vector<Piece> pieces;
    size_t p = 0;
    for (int i = 0; i < count; ++i)
      for (int c = 0; c < imgs[i].ncomp; ++c, ++p) pieces.push_back({imgs[i].comp[c].coef, coef_off[p], (size_t)imgs[i].comp[c].bw * imgs[i].comp[c].bh * 64});
    const int nthreads = (int)std::min<size_t>(8, std::max<size_t>(1, (ncoef * 2) >> 22));
    auto run = [&](int t) { for (size_t k = t; k < pieces.size(); k += nthreads) memcpy(sc.pinned + pieces[k].off, pieces[k].src, pieces[k].n * sizeof(int16_t)); };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(run, t);
    run(0);
    for (auto& t : th) t.join();
  }
  size_t p = 0;
  for (int i = 0; i < count; ++i) {
    const ocr_jpeg_img& im = imgs[i];
    JpegImageDesc& d = id[i];
    d = JpegImageDesc{};
    d.rows = im.rows; d.cols = im.cols; d.ncomp = im.ncomp; d.hmax = im.hmax; d.vmax = im.vmax; d.bgr = dst[i];
    for (int c = 0; c < im.ncomp; ++c, ++p) {
      pd[p].coef = sc.coef.p + coef_off[p];
      pd[p].plane = sc.planes.p + plane_off[p];
      d.plane[c] = pd[p].plane;
      d.stride[c] = im.comp[c].bw * 8;
      d.dw[c] = im.comp[c].dw;
      d.dh[c] = im.comp[c].dh;
    }
  }
  if (hipMemcpyAsync(sc.coef.p, sc.pinned, ncoef * sizeof(int16_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipEventRecord(sc.copied, s) != hipSuccess ||
      hipMemcpyAsync(sc.pd.p, pd.data(), pd.size() * sizeof(JpegPlaneDesc), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(sc.id.p, id.data(), id.size() * sizeof(JpegImageDesc), hipMemcpyHostToDevice, s) != hipSuccess) {
    err = "JPEG coefficient upload failed";
    return OCR_ERR_DEVICE;
  }
  launch_jpeg_idct(sc.pd.p, (int)pd.size(), nblocks, s);
  launch_jpeg_output(sc.id.p, count, max_px, s);
  if (hipGetLastError() != hipSuccess) { err = "JPEG kernels failed to launch"; return OCR_ERR_DEVICE; }
  return OCR_OK;
}

}  // namespace ocr

using namespace ocr;

extern "C" int ocr_jpeg_decode(const ocr_jpeg_img* img, int device_id, uint8_t* bgr, size_t cap) {
  if (!img || !bgr) return fail(OCR_ERR_ARG, "null argument");
  int rc = ocr_rt_init(device_id);
  if (rc) return rc;
  if (!jpeg_img_valid(*img)) return fail(OCR_ERR_ARG, "bad JPEG coefficient descriptor");
  const size_t bytes = (size_t)img->rows * img->cols * 3;
  if (bytes > cap) return fail(OCR_ERR_CAPACITY, "output buffer too small");
  JpegScratch sc;
  DevBuf<uint8_t> out;
  std::string err;
  if (!out.ensure(bytes, err)) return fail(OCR_ERR_DEVICE, err);
  uint8_t* dst = out.p;
  rc = jpeg_decode_async(img, 1, &dst, sc, nullptr, err);
  if (rc) return fail(rc, err);
  CAPI_HIP(g_memcpy(bgr, out.p, bytes, hipMemcpyDeviceToHost));
  return OCR_OK;
}
