// Host-side geometry of the perspective crop: Utility::GetRotateCropImage
// (/root/reference/src/utility.cpp:137-190).  The 8x8 homography solve and the 3x3 inverse are a few
// hundred double operations per text box, so they stay on the host (the boxes are on the host anyway
// between det and rec); the per-pixel work runs in warp_crop_kernel (kernels_pre.hip).
// Arithmetic: plain IEEE double, evaluation order of cv::getPerspectiveTransform (LU with partial
// pivoting) and of cv::invert's 3x3 closed form; this file is compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <utility>

#include "kernels_pre.h"

namespace ocr {

struct CropPlan {
  int left, top, sw, sh;  // bounding-box crop inside the source image (the warp's source)
  int dw, dh;             // warp output size
  int rot;                // dh >= 1.5 dw: the result is transposed and flipped (90 degrees)
  int orows, ocols;       // size of the returned image
  int bw0;                // column block of cv::WarpPerspectiveInvoker
  double minv[9];         // inverse homography (destination -> source)
};

namespace cropmath {
// Gaussian elimination with row pivoting on an 8x8 system, one right-hand side; false if singular
inline bool solve8(double a[8][8], double b[8]) {
  const double tiny = 2.220446049250313e-16 * 100;
  for (int col = 0; col < 8; ++col) {
    int piv = col;
    for (int r = col + 1; r < 8; ++r)
      if (std::fabs(a[r][col]) > std::fabs(a[piv][col])) piv = r;
    if (std::fabs(a[piv][col]) < tiny) return false;
    if (piv != col) {
      for (int c = col; c < 8; ++c) std::swap(a[col][c], a[piv][c]);
      std::swap(b[col], b[piv]);
    }
    const double ninv = -1 / a[col][col];
    for (int r = col + 1; r < 8; ++r) {
      const double f = a[r][col] * ninv;
      for (int c = col + 1; c < 8; ++c) a[r][c] += f * a[col][c];
      b[r] += f * b[col];
    }
  }
  for (int r = 7; r >= 0; --r) {
    double acc = b[r];
    for (int c = r + 1; c < 8; ++c) acc -= a[r][c] * b[c];
    b[r] = acc / a[r][r];
  }
  return true;
}
inline void inverse3(const double m[9], double o[9]) {
  const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[3] * m[8] - m[5] * m[6], c02 = m[3] * m[7] - m[4] * m[6];
  const double det = m[0] * c00 - m[1] * c01 + m[2] * c02;
  if (det == 0.) {
    for (int i = 0; i < 9; ++i) o[i] = 0;
    return;
  }
  const double s = 1. / det;
  o[0] = c00 * s;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * s;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * s;
  o[3] = (m[5] * m[6] - m[3] * m[8]) * s;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * s;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * s;
  o[6] = c02 * s;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * s;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * s;
}
}  // namespace cropmath

// false: the bounding-box crop is empty or not inside the image (cv::Mat::operator()(Rect) throws there)
inline bool plan_rotate_crop(int rows, int cols, const int32_t* b, CropPlan& p) {
  int x0 = b[0], x1 = b[0], y0 = b[1], y1 = b[1];
  for (int k = 1; k < 4; ++k) {
    x0 = std::min(x0, (int)b[2 * k]); x1 = std::max(x1, (int)b[2 * k]);
    y0 = std::min(y0, (int)b[2 * k + 1]); y1 = std::max(y1, (int)b[2 * k + 1]);
  }
  if (x0 < 0 || y0 < 0 || x1 > cols || y1 > rows || x1 <= x0 || y1 <= y0) return false;
  p.left = x0; p.top = y0; p.sw = x1 - x0; p.sh = y1 - y0;
  const int cw = int(std::sqrt(std::pow(b[0] - b[2], 2) + std::pow(b[1] - b[3], 2)));
  const int ch = int(std::sqrt(std::pow(b[0] - b[6], 2) + std::pow(b[1] - b[7], 2)));
  // an empty dsize makes cv::warpPerspective fall back to the source size
  p.dw = (cw > 0 && ch > 0) ? cw : p.sw;
  p.dh = (cw > 0 && ch > 0) ? ch : p.sh;
  p.rot = float(p.dh) >= float(p.dw) * 1.5;
  p.orows = p.rot ? p.dw : p.dh;
  p.ocols = p.rot ? p.dh : p.dw;
  const int bh0 = std::min(16, p.dh);
  p.bw0 = std::min(1024 / bh0, p.dw);
  double a[8][8], rhs[8];
  const float tx[4] = {0.f, (float)cw, (float)cw, 0.f}, ty[4] = {0.f, 0.f, (float)ch, (float)ch};
  for (int i = 0; i < 4; ++i) {
    const float sx = (float)(b[2 * i] - x0), sy = (float)(b[2 * i + 1] - y0);
    for (int c = 0; c < 8; ++c) a[i][c] = a[i + 4][c] = 0;
    a[i][0] = a[i + 4][3] = sx;
    a[i][1] = a[i + 4][4] = sy;
    a[i][2] = a[i + 4][5] = 1;
    a[i][6] = -(double)sx * tx[i];
    a[i][7] = -(double)sy * tx[i];
    a[i + 4][6] = -(double)sx * ty[i];
    a[i + 4][7] = -(double)sy * ty[i];
    rhs[i] = tx[i];
    rhs[i + 4] = ty[i];
  }
  double fwd[9];
  if (!cropmath::solve8(a, rhs))
    for (int i = 0; i < 8; ++i) rhs[i] = 0;
  for (int i = 0; i < 8; ++i) fwd[i] = rhs[i];
  fwd[8] = 1.;
  cropmath::inverse3(fwd, p.minv);
  return true;
}

}  // namespace ocr
