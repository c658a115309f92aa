// decode_tool <image file> <out.ppm>: runs the IPC service's image decoders (test aid, no GPU needed).
#include <cstdio>

#include "ocr_ipc_service.h"

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: decode_tool <in> <out.ppm>\n"); return 2; }
  std::vector<uint8_t> bytes;
  PaddleOCR::Image im;
  if (!PaddleOCR::ipc::read_file(argv[1], bytes) || !PaddleOCR::ipc::decode_image(bytes, im)) { fprintf(stderr, "decode failed\n"); return 1; }
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 1;
  fprintf(f, "P6\n%d %d\n255\n", im.cols, im.rows);
  for (size_t p = 0; p < (size_t)im.rows * im.cols; ++p) { const uint8_t rgb[3] = {im.pixels[3 * p + 2], im.pixels[3 * p + 1], im.pixels[3 * p]}; fwrite(rgb, 1, 3, f); }
  fclose(f);
  return 0;
}
