// Host-visible launch interface of kernels_jpeg.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocr {

struct JpegPlaneDesc {       // one component of one image
  const int16_t* coef;       // device: bw*bh blocks x 64 quantised coefficients, natural order
  uint16_t quant[64];        // its quantisation table, natural order
  uint8_t* plane;            // device: (bh*8) x (bw*8) samples out
  int bw, bh;
  long first_block;          // prefix sum of blocks over the launch's planes
};
struct JpegImageDesc {
  const uint8_t* plane[3];
  int stride[3], dw[3], dh[3];
  uint8_t* bgr;              // device: rows x cols packed BGR out
  int rows, cols, ncomp, hmax, vmax;
};
void launch_jpeg_idct(const JpegPlaneDesc* descs, int ndesc, long total_blocks, hipStream_t s);
void launch_jpeg_output(const JpegImageDesc* imgs, int nimg, long max_pixels, hipStream_t s);

}  // namespace ocr
