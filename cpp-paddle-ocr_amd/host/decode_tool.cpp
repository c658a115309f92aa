// decode_tool [--device] <image file> <out.ppm>: runs the IPC service's image decoders (test aid).  --device: a JPEG's
// pixel half (IDCT, upsampling, colour conversion) runs on the GPU through ocr_jpeg_decode instead of on the host.
#include <cstdio>
#include <cstring>

#include "ocr_ipc_service.h"

int main(int argc, char** argv) {
  const bool device = argc == 4 && !strcmp(argv[1], "--device");
  if (argc != 3 && !device) { fprintf(stderr, "usage: decode_tool [--device] <in> <out.ppm>\n"); return 2; }
  const char* in = argv[device ? 2 : 1];
  const char* outp = argv[device ? 3 : 2];
  std::vector<uint8_t> bytes;
  PaddleOCR::Image im;
  if (!PaddleOCR::ipc::read_file(in, bytes) || !PaddleOCR::ipc::decode_image(bytes, im, device) || im.empty()) { fprintf(stderr, "decode failed\n"); return 1; }
  if (im.device_decodable()) {
    const ocr_jpeg_img d = im.jpeg_desc();
    im.pixels.resize((size_t)d.rows * d.cols * 3);
    if (ocr_jpeg_decode(&d, 0, im.pixels.data(), im.pixels.size()) != OCR_OK) { fprintf(stderr, "device decode failed: %s\n", ocr_last_error()); return 1; }
  }
  FILE* f = fopen(outp, "wb");
  if (!f) return 1;
  fprintf(f, "P6\n%d %d\n255\n", im.cols, im.rows);
  for (size_t p = 0; p < (size_t)im.rows * im.cols; ++p) { const uint8_t rgb[3] = {im.pixels[3 * p + 2], im.pixels[3 * p + 1], im.pixels[3 * p]}; fwrite(rgb, 1, 3, f); }
  fclose(f);
  return 0;
}
