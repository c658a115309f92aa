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
#include <climits>
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

// ------------------------------------------------------------------ Utility::GetRotateCropImage
// /root/reference/src/utility.cpp:137-190.  OpenCV pieces restated (OpenCV 4.x, classic code path):
//   getPerspectiveTransform : 8x8 system in double, cv::solve(DECOMP_LU) = LU with partial pivoting
//   warpPerspective         : the reference passes cv::BORDER_REPLICATE (=1) in the FLAGS slot
//                             (utility.cpp:178-180), so flags = INTER_LINEAR and the border mode stays
//                             BORDER_CONSTANT with value 0; M is inverted with the 3x3 closed form;
//                             coordinates in double per (bw0 x bh0) block, rounded to 1/32 px,
//                             bilinear taps with 15-bit fixed-point weights
//   transpose + flip(0)     : when rows >= 1.5 * cols
namespace {
// hal::LU64f (modules/core/src/matrix_decomp.cpp LUImpl), one right-hand side
int lu_solve(double* A, int m, double* b) {
  const double eps = 2.220446049250313e-16 * 100;
  for (int i = 0; i < m; ++i) {
    int k = i;
    for (int j = i + 1; j < m; ++j)
      if (std::fabs(A[j * m + i]) > std::fabs(A[k * m + i])) k = j;
    if (std::fabs(A[k * m + i]) < eps) return 0;
    if (k != i) {
      for (int j = i; j < m; ++j) std::swap(A[i * m + j], A[k * m + j]);
      std::swap(b[i], b[k]);
    }
    const double d = -1 / A[i * m + i];
    for (int j = i + 1; j < m; ++j) {
      const double alpha = A[j * m + i] * d;
      for (int c = i + 1; c < m; ++c) A[j * m + c] += alpha * A[i * m + c];
      b[j] += alpha * b[i];
    }
  }
  for (int i = m - 1; i >= 0; --i) {
    double s = b[i];
    for (int c = i + 1; c < m; ++c) s -= A[i * m + c] * b[c];
    b[i] = s / A[i * m + i];
  }
  return 1;
}
// cv::getPerspectiveTransform(src[4], dst[4]) -> M[9]
void perspective_transform(const float sx[4], const float sy[4], const float dx[4], const float dy[4], double M[9]) {
  double a[64], b[8];
  for (int i = 0; i < 4; ++i) {
    double* r0 = a + i * 8;
    double* r1 = a + (i + 4) * 8;
    r0[0] = r1[3] = sx[i];
    r0[1] = r1[4] = sy[i];
    r0[2] = r1[5] = 1;
    r0[3] = r0[4] = r0[5] = r1[0] = r1[1] = r1[2] = 0;
    r0[6] = -(double)sx[i] * dx[i];
    r0[7] = -(double)sy[i] * dx[i];
    r1[6] = -(double)sx[i] * dy[i];
    r1[7] = -(double)sy[i] * dy[i];
    b[i] = dx[i];
    b[i + 4] = dy[i];
  }
  if (!lu_solve(a, 8, b))
    for (int i = 0; i < 8; ++i) b[i] = 0;  // cv::solve on a singular system leaves X = 0
  for (int i = 0; i < 8; ++i) M[i] = b[i];
  M[8] = 1.;
}
// cv::invert of a 3x3 CV_64F (closed form, matrix_decomp / lapack.cpp); singular -> all zeros
void invert3(const double S[9], double T[9]) {
  const double d0 = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
  if (d0 == 0.) {
    for (int i = 0; i < 9; ++i) T[i] = 0;
    return;
  }
  const double d = 1. / d0;
  T[0] = (S[4] * S[8] - S[5] * S[7]) * d;
  T[1] = (S[2] * S[7] - S[1] * S[8]) * d;
  T[2] = (S[1] * S[5] - S[2] * S[4]) * d;
  T[3] = (S[5] * S[6] - S[3] * S[8]) * d;
  T[4] = (S[0] * S[8] - S[2] * S[6]) * d;
  T[5] = (S[2] * S[3] - S[0] * S[5]) * d;
  T[6] = (S[3] * S[7] - S[4] * S[6]) * d;
  T[7] = (S[1] * S[6] - S[0] * S[7]) * d;
  T[8] = (S[0] * S[4] - S[1] * S[3]) * d;
}
inline int round_sat_int(double v) {  // saturate_cast<int>(double) = cvRound: nearest-even
  return (int)std::nearbyint(v);
}
inline short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
// cv::warpPerspective(src, dst(dh x dw), M, INTER_LINEAR, BORDER_CONSTANT, 0) for 8UC3
void warp_perspective_u8c3(const uint8_t* src, int sh, int sw, size_t sstride, const double M0[9], uint8_t* dst, int dh, int dw) {
  double M[9];
  invert3(M0, M);
  const int BLOCK_SZ = 32;
  int bh0 = std::min(BLOCK_SZ / 2, dh);
  const int bw0 = std::min(BLOCK_SZ * BLOCK_SZ / bh0, dw);
  bh0 = std::min(BLOCK_SZ * BLOCK_SZ / bw0, dh);
  (void)bh0;  // rows of a block share nothing: only the column split enters the arithmetic
  for (int y = 0; y < dh; ++y)
    for (int xb = 0; xb < dw; xb += bw0) {
      const int bw = std::min(bw0, dw - xb);
      const double X0 = M[0] * xb + M[1] * y + M[2];
      const double Y0 = M[3] * xb + M[4] * y + M[5];
      const double W0 = M[6] * xb + M[7] * y + M[8];
      for (int x1 = 0; x1 < bw; ++x1) {
        double W = W0 + M[6] * x1;
        W = W ? 32. / W : 0;
        const double fX = std::max((double)INT_MIN, std::min((double)INT_MAX, (X0 + M[0] * x1) * W));
        const double fY = std::max((double)INT_MIN, std::min((double)INT_MAX, (Y0 + M[3] * x1) * W));
        const int X = round_sat_int(fX), Y = round_sat_int(fY);
        const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
        const int fx = X & 31, fy = Y & 31;
        // 15-bit weights of the (fy, fx) table entry; entry (0,0) is {32767,0,0,1} in OpenCV after its
        // sum fix-up, which rounds to the same byte as {32768,0,0,0} for every input
        const int w[4] = {(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32};
        uint8_t* D = dst + ((size_t)y * dw + xb + x1) * 3;
        if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) {
          D[0] = D[1] = D[2] = 0;
          continue;
        }
        for (int k = 0; k < 3; ++k) {
          auto at = [&](int yy, int xx) -> int {
            return (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * sstride + (size_t)xx * 3 + k] : 0;
          };
          const int v = at(sy, sx) * w[0] + at(sy, sx + 1) * w[1] + at(sy + 1, sx) * w[2] + at(sy + 1, sx + 1) * w[3];
          const int r = (v + (1 << 14)) >> 15;
          D[k] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
      }
    }
}
}  // namespace

extern "C" {
// out dims of GetRotateCropImage for a box (utility.cpp:143-164, 182-189); returns 0 when the
// bounding-box crop is empty or leaves the image (cv::Mat::operator() would throw inside noexcept)
int oracle_rotate_crop_shape(int rows, int cols, const int* box8, int* orows, int* ocols) {
  int left = box8[0], right = box8[0], top = box8[1], bottom = box8[1];
  for (int i = 1; i < 4; ++i) {
    left = std::min(left, box8[2 * i]); right = std::max(right, box8[2 * i]);
    top = std::min(top, box8[2 * i + 1]); bottom = std::max(bottom, box8[2 * i + 1]);
  }
  if (left < 0 || top < 0 || right > cols || bottom > rows || right - left <= 0 || bottom - top <= 0) return 0;
  const int cw = int(std::sqrt(std::pow(box8[0] - box8[2], 2) + std::pow(box8[1] - box8[3], 2)));
  const int ch = int(std::sqrt(std::pow(box8[0] - box8[6], 2) + std::pow(box8[1] - box8[7], 2)));
  int dw = cw, dh = ch;
  if (cw <= 0 || ch <= 0) { dw = right - left; dh = bottom - top; }  // dsize.empty() -> src.size()
  if (float(dh) >= float(dw) * 1.5) { *orows = dw; *ocols = dh; }
  else { *orows = dh; *ocols = dw; }
  return 1;
}
// --- pieces, exposed for unit tests (tests/test_oracle_independent.py) ---
// cv::getPerspectiveTransform: src / dst = 4 (x, y) pairs each
void oracle_perspective_transform(const float* src_xy, const float* dst_xy, double* M9) {
  float sx[4], sy[4], dx[4], dy[4];
  for (int i = 0; i < 4; ++i) { sx[i] = src_xy[2 * i]; sy[i] = src_xy[2 * i + 1]; dx[i] = dst_xy[2 * i]; dy[i] = dst_xy[2 * i + 1]; }
  perspective_transform(sx, sy, dx, dy, M9);
}
// cv::warpPerspective(src, dst(dh x dw), M, INTER_LINEAR, BORDER_CONSTANT 0) for 8UC3 with a given forward matrix
void oracle_warp_perspective(const uint8_t* src, int sh, int sw, size_t sstride, const double* M9, uint8_t* dst, int dh, int dw) {
  warp_perspective_u8c3(src, sh, sw, sstride, M9, dst, dh, dw);
}
int oracle_rotate_crop(const uint8_t* bgr, int rows, int cols, size_t stride, const int* box8, uint8_t* out, int* orows,
                       int* ocols) {
  if (!oracle_rotate_crop_shape(rows, cols, box8, orows, ocols)) return 0;
  int left = box8[0], right = box8[0], top = box8[1], bottom = box8[1];
  for (int i = 1; i < 4; ++i) {
    left = std::min(left, box8[2 * i]); right = std::max(right, box8[2 * i]);
    top = std::min(top, box8[2 * i + 1]); bottom = std::max(bottom, box8[2 * i + 1]);
  }
  const int cw = int(std::sqrt(std::pow(box8[0] - box8[2], 2) + std::pow(box8[1] - box8[3], 2)));
  const int ch = int(std::sqrt(std::pow(box8[0] - box8[6], 2) + std::pow(box8[1] - box8[7], 2)));
  const float px[4] = {(float)(box8[0] - left), (float)(box8[2] - left), (float)(box8[4] - left), (float)(box8[6] - left)};
  const float py[4] = {(float)(box8[1] - top), (float)(box8[3] - top), (float)(box8[5] - top), (float)(box8[7] - top)};
  const float qx[4] = {0.f, (float)cw, (float)cw, 0.f}, qy[4] = {0.f, 0.f, (float)ch, (float)ch};
  double M[9];
  perspective_transform(px, py, qx, qy, M);
  int dw = cw, dh = ch;
  if (cw <= 0 || ch <= 0) { dw = right - left; dh = bottom - top; }
  std::vector<uint8_t> warped((size_t)dh * dw * 3);
  warp_perspective_u8c3(bgr + (size_t)top * stride + (size_t)left * 3, bottom - top, right - left, stride, M, warped.data(), dh, dw);
  if (float(dh) >= float(dw) * 1.5) {
    // transpose, then flip around the x axis: out(r, c) = warped(c, dw-1-r), out is dw rows x dh cols
    for (int r = 0; r < dw; ++r)
      for (int c = 0; c < dh; ++c) memcpy(out + ((size_t)r * dh + c) * 3, &warped[((size_t)c * dw + (dw - 1 - r)) * 3], 3);
  } else {
    memcpy(out, warped.data(), warped.size());
  }
  return 1;
}
}
