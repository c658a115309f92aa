// Host-visible launch interface of kernels_pre.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocr {

struct DetPreArgs {
  const uint8_t* src;       // N images, each src_image_bytes apart, rows src_stride apart (BGR u8)
  size_t src_image_bytes, src_stride;
  int N, sh, sw, dh, dw;
  const float* lut;         // [3][256] normalised value per channel/byte
  float* out;               // [N][dh][dw][3] f32
  uint8_t* resized;         // [N][dh][dw][3] u8 or null
};
void launch_det_pre(const DetPreArgs& a, hipStream_t s);

// One text line: an ROI (x,y,w,h) of a device-resident BGR image, resized to (imgH x resize_w),
// written (right-padded to imgW) into slot `slot` of the batch tensor.
struct LineDesc {
  const uint8_t* img;
  size_t stride;
  int x, y, w, h;
  int resize_w;
  int slot;
  // cv::resize's scale factors of this line, 1 / ((double)dst / src) per axis, computed once on the host (the same two
  // IEEE double operations the kernel used to repeat in every thread: four f64 divisions were 40 % of its instructions)
  double scale_x, scale_y;
  // ragged batch (launch_line_pre_ragged / launch_ctc_ragged): the line's own tensor width, its first pixel in the
  // batch tensor (= imgH * the widths of the lines before it), and where its CTC steps sit in the step arrays
  int tensor_w = 0, pix0 = 0, step0 = 0, steps = 0;
  static LineDesc make(const uint8_t* img, size_t stride, int x, int y, int w, int h, int resize_w, int slot, int imgH) {
    LineDesc d{img, stride, x, y, w, h, resize_w, slot, 0.0, 0.0, 0, 0, 0, 0};
    const double inv_x = (double)resize_w / w, inv_y = (double)imgH / h;
    d.scale_x = 1. / inv_x;
    d.scale_y = 1. / inv_y;
    return d;
  }
};
void launch_line_pre(const LineDesc* lines, int nlines, int imgH, int imgW, const float* lut, bool pad_after_norm,
                     float* out, hipStream_t s);
void launch_rotate180(uint8_t* img, size_t stride, int x0, int y0, int w, int h, hipStream_t s);
struct RotDesc {
  uint8_t* img;
  size_t stride;
  int x, y, w, h;
};
// d: ROIs grouped into components of mutually reachable intersecting ROIs, each group in request order; seg: ngroups+1
// offsets into d.  One workgroup per group applies its rotations one after the other, groups run concurrently.
void launch_rotate180_groups(const RotDesc* d, const int* seg, int ngroups, hipStream_t s);
// One perspective crop (Utility::GetRotateCropImage): destination pixel -> source position through
// the inverse homography, bilinear taps in 15-bit fixed point, constant-0 border, optional 90-degree turn.
struct WarpDesc {
  const uint8_t* src;   // first pixel of the bounding-box crop inside the source image
  size_t sstride;
  int sw, sh;           // crop size (taps outside read 0)
  uint8_t* dst;         // packed result, (rot ? dw x dh : dh x dw) x 3
  int dw, dh, rot, bw0;
  double m[9];
};
void launch_warp_crops(const WarpDesc* d, int ncrops, int max_pixels, hipStream_t s);
// ragged batch: line i = [imgH][lines[i].tensor_w][3] at pixel lines[i].pix0 of `out` (lines ordered by pix0)
void launch_line_pre_ragged(const LineDesc* lines, int nlines, long total_pixels, int imgH, const float* lut, float* out,
                            hipStream_t s);
void launch_ctc(const int* amax, const float* pmax, int nlines, int T, int max_len, int* ids, int* lens, float* scores,
                hipStream_t s);
// ragged batch: line i has lines[i].steps steps starting at lines[i].step0 of amax / pmax
void launch_ctc_ragged(const int* amax, const float* pmax, const LineDesc* lines, int nlines, int max_len, int* ids, int* lens,
                       float* scores, hipStream_t s);

}  // namespace ocr
