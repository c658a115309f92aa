// JPEG coefficients -> device pixels for a batch of images (host side of kernels_jpeg.hip).
#pragma once
#include <string>
#include <vector>

#include "kernels_jpeg.h"
#include "stages.h"

namespace ocr {

struct JpegScratch {
  DevBuf<int16_t> coef;
  DevBuf<uint8_t> planes;
  DevBuf<JpegPlaneDesc> pd;
  DevBuf<JpegImageDesc> id;
  int16_t* pinned = nullptr;   // coefficient staging (hipHostMalloc)
  size_t pinned_cap = 0;       // in int16
  hipEvent_t copied = nullptr; // the pinned buffer may be refilled once this has passed
  ~JpegScratch();
};
// Validates the descriptors, stages the coefficients through pinned memory, and enqueues upload + IDCT + upsampling /
// colour conversion on `s`; image i is written as packed BGR to dst[i] (device).  Returns an OCR_* code.
int jpeg_decode_async(const ocr_jpeg_img* imgs, int count, uint8_t* const* dst, JpegScratch& sc, hipStream_t s, std::string& err);
bool jpeg_img_valid(const ocr_jpeg_img& im);

}  // namespace ocr
