// TEST INFRASTRUCTURE — CPU oracle, not part of the product path.
//
// Restatement of the image-side arithmetic of the reference's pre-processing:
//   ResizeImgType0::Run   /root/reference/src/preprocess_op.cpp:57-93
//   CrnnResizeImg::Run    /root/reference/src/preprocess_op.cpp:95-118
//   ClsResizeImg::Run     /root/reference/src/preprocess_op.cpp:120-137
//   Normalize::Run        /root/reference/src/preprocess_op.cpp:40-55
//   Permute/PermuteBatch  /root/reference/src/preprocess_op.cpp:19-38   (layout only; we emit NHWC)
// The pixel arithmetic lives in OpenCV 4.x (vcpkg, version unpinned in the reference README,
// absent from /root/reference): cv::resize INTER_LINEAR 8UC3 (fixed point, 11-bit coefficients),
// Mat::convertTo, cv::copyMakeBorder.  Their published algorithms are restated here
// (SURVEY.md Appendix B.1, B.2, B.9).  PARITY UNPINNED: the reference holds no golden image
// vectors and OpenCV cannot run in this container.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline int cv_round_half_even(float v) { return (int)lrintf(v); }  // cvRound under the default rounding mode
inline short sat_short(float v) {
  int r = cv_round_half_even(v);
  return (short)std::min(32767, std::max(-32768, r));
}

// cv::resize(src, dst, Size(dw, dh)) for CV_8UC3, default INTER_LINEAR.
void resize_linear_u8c3(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw) {
  const int cn = 3;
  if (sh == dh && sw == dw) {
    for (int y = 0; y < sh; ++y) memcpy(dst + (size_t)y * dw * cn, src + y * sstride, (size_t)sw * cn);
    return;
  }
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);  // saturate_cast<int>(double)
  const bool is_area_fast = std::abs(scale_x - iscale_x) < 2.220446049250313e-16 && std::abs(scale_y - iscale_y) < 2.220446049250313e-16;
  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    // INTER_LINEAR is silently replaced by INTER_AREA for an exact 2x2 decimation
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x)
        for (int c = 0; c < cn; ++c) {
          const uint8_t* s0 = src + (size_t)(2 * y) * sstride + (2 * x) * cn + c;
          const uint8_t* s1 = s0 + sstride;
          dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((s0[0] + s0[cn] + s1[0] + s1[cn] + 2) >> 2);
        }
    return;
  }
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  int xmax = dw;
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    ialpha[dx * 2] = sat_short((1.f - fx) * 2048.f);
    ialpha[dx * 2 + 1] = sat_short(fx * 2048.f);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    yofs[dy] = sy;
    ibeta[dy * 2] = sat_short((1.f - fy) * 2048.f);
    ibeta[dy * 2 + 1] = sat_short(fy * 2048.f);
  }
  auto clip = [](int v, int lo, int hi) { return v >= lo ? (v < hi ? v : hi - 1) : lo; };
  std::vector<int> row0((size_t)dw * cn), row1((size_t)dw * cn);
  auto hresize = [&](const uint8_t* s, std::vector<int>& d) {
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx];
      for (int c = 0; c < cn; ++c) {
        if (dx < xmax) d[dx * cn + c] = s[sx * cn + c] * ialpha[dx * 2] + s[(sx + 1) * cn + c] * ialpha[dx * 2 + 1];
        else d[dx * cn + c] = s[sx * cn + c] * 2048;
      }
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    const int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(src + (size_t)sy0 * sstride, row0);
    hresize(src + (size_t)sy1 * sstride, row1);
    const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* d = dst + (size_t)dy * dw * cn;
    for (int i = 0; i < dw * cn; ++i) {
      int v = (((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2;
      d[i] = (uint8_t)std::min(255, std::max(0, v));
    }
  }
}

// Normalize::Run per value: convertTo(CV_32FC3, 1/255) then per channel convertTo(alpha=scale, beta=-mean*scale).
inline float normalize_px(uint8_t v, float mean, float scale) {
  const float e = (float)(1.0 / 255.0);
  const float f = (float)v * e;
  const float a = (float)(1.0 * (double)scale);
  const float b = (float)((0.0 - (double)mean) * (double)scale);
  return fmaf(f, a, b);  // OpenCV's SIMD cvt uses v_fma (B.2); the scalar tail differs by <= 1 ulp
}

}  // namespace

extern "C" {

void oracle_resize_u8c3(const uint8_t* src, int sh, int sw, size_t sstride, uint8_t* dst, int dh, int dw) {
  resize_linear_u8c3(src, sh, sw, sstride, dst, dh, dw);
}

// ResizeImgType0 size rule. returns resize_h/resize_w and ratios.
void oracle_det_resize_shape(int h, int w, const char* limit_type, int limit_side_len, int* rh, int* rw, float* ratio_h,
                             float* ratio_w) {
  float ratio = 1.f;
  if (!strcmp(limit_type, "min")) {
    int min_wh = std::min(h, w);
    if (min_wh < limit_side_len) ratio = h < w ? float(limit_side_len) / float(h) : float(limit_side_len) / float(w);
  } else {
    int max_wh = std::max(h, w);
    if (max_wh > limit_side_len) ratio = h > w ? float(limit_side_len) / float(h) : float(limit_side_len) / float(w);
  }
  int resize_h = int(float(h) * ratio);
  int resize_w = int(float(w) * ratio);
  resize_h = std::max(int(round(float(resize_h) / 32) * 32), 32);
  resize_w = std::max(int(round(float(resize_w) / 32) * 32), 32);
  *rh = resize_h;
  *rw = resize_w;
  *ratio_h = float(resize_h) / float(h);
  *ratio_w = float(resize_w) / float(w);
}

// det pre-processing: resize -> normalize (ImageNet mean/scale in BGR memory order) -> NHWC f32.
// out must hold rh*rw*3 floats; resized (optional) receives the u8 image.
void oracle_det_preprocess(const uint8_t* bgr, int h, int w, size_t stride, int rh, int rw, float* out, uint8_t* resized) {
  std::vector<uint8_t> tmp((size_t)rh * rw * 3);
  resize_linear_u8c3(bgr, h, w, stride, tmp.data(), rh, rw);
  const float mean[3] = {0.485f, 0.456f, 0.406f};
  const float scale[3] = {1 / 0.229f, 1 / 0.224f, 1 / 0.225f};
  for (size_t i = 0; i < (size_t)rh * rw; ++i)
    for (int c = 0; c < 3; ++c) out[i * 3 + c] = normalize_px(tmp[i * 3 + c], mean[c], scale[c]);
  if (resized) memcpy(resized, tmp.data(), tmp.size());
}

// CrnnResizeImg + Normalize for one line: output NHWC f32 [imgH][imgW][3], imgW = int(imgH*max_wh_ratio).
// returns imgW.  (pad is u8 zero BEFORE normalisation -> -1.0)
int oracle_rec_width(int imgH, float max_wh_ratio) { return int(imgH * max_wh_ratio); }
int oracle_rec_resize_w(int rows, int cols, int imgH, int imgW) {
  float ratio = float(cols) / float(rows);
  return ceilf(imgH * ratio) > imgW ? imgW : int(ceilf(imgH * ratio));
}
void oracle_rec_preprocess(const uint8_t* bgr, int rows, int cols, size_t stride, int imgH, int imgW, float* out) {
  const int resize_w = oracle_rec_resize_w(rows, cols, imgH, imgW);
  std::vector<uint8_t> tmp((size_t)imgH * resize_w * 3);
  resize_linear_u8c3(bgr, rows, cols, stride, tmp.data(), imgH, resize_w);
  for (int y = 0; y < imgH; ++y)
    for (int x = 0; x < imgW; ++x)
      for (int c = 0; c < 3; ++c) {
        const uint8_t v = x < resize_w ? tmp[((size_t)y * resize_w + x) * 3 + c] : 0;
        out[((size_t)y * imgW + x) * 3 + c] = normalize_px(v, 0.5f, 1 / 0.5f);
      }
}

// ClsResizeImg + Normalize + right pad with 0.0f (AFTER normalisation) to 192.
void oracle_cls_preprocess(const uint8_t* bgr, int rows, int cols, size_t stride, float* out /*48x192x3*/) {
  const int imgH = 48, imgW = 192;
  const int resize_w = oracle_rec_resize_w(rows, cols, imgH, imgW);
  std::vector<uint8_t> tmp((size_t)imgH * resize_w * 3);
  resize_linear_u8c3(bgr, rows, cols, stride, tmp.data(), imgH, resize_w);
  for (int y = 0; y < imgH; ++y)
    for (int x = 0; x < imgW; ++x)
      for (int c = 0; c < 3; ++c)
        out[((size_t)y * imgW + x) * 3 + c] = x < resize_w ? normalize_px(tmp[((size_t)y * resize_w + x) * 3 + c], 0.5f, 1 / 0.5f) : 0.0f;
}

// cv::rotate(roi, roi, ROTATE_180) in place on a (possibly strided) ROI.
void oracle_rotate180_inplace(uint8_t* bgr, int rows, int cols, size_t stride) {
  const long total = (long)rows * cols;
  for (long i = 0; i < total / 2; ++i) {
    const long j = total - 1 - i;
    uint8_t* a = bgr + (i / cols) * stride + (i % cols) * 3;
    uint8_t* b = bgr + (j / cols) * stride + (j % cols) * 3;
    for (int c = 0; c < 3; ++c) std::swap(a[c], b[c]);
  }
}

}  // extern "C"
