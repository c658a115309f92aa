// Device half of JPEG decoding (image decode moved to the GPU; the reference decodes with
// cv::imdecode / cv::imread on the host, /root/reference/src/ocr_ipc_service.cpp:42,336).
//
// The bit-serial part of a JPEG - Huffman decoding of the scans, sequential or progressive - stays on the host
// (host/jpeg_decode.h) and ends in quantised DCT coefficients.  Everything after that is data parallel and runs here,
// straight into the pipeline's device image, stage by stage what libjpeg(-turbo) computes with default settings:
//   * dequantisation + the "islow" integer IDCT (jidctint.c: 13-bit constants, two passes, PASS1_BITS = 2)
//   * "fancy" (triangle) chroma upsampling h2v1 / h2v2 with replicated edges (jdsample.c), evaluated per output pixel
//   * YCbCr -> RGB with the 16-bit fixed-point constants of jdcolor.c, written as packed BGR
// Integer arithmetic throughout: results equal host/jpeg_decode.h (itself pinned to libjpeg-turbo through PIL) bit for
// bit - tests/test_ipc_service.py::test_device_jpeg_decode_equals_host.
#include <hip/hip_runtime.h>

#include "kernels_jpeg.h"

namespace ocr {

namespace {

__device__ __forceinline__ int jdescale(long long x, int n) { return (int)((x + (1LL << (n - 1))) >> n); }
__device__ __forceinline__ uint8_t jclamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// one 8-point pass of jpeg_idct_islow on s[0..7] (64-bit intermediates, as the host restatement's `long`)
__device__ __forceinline__ void idct8(const long long (&s)[8], long long (&o)[8]) {
  const int CB = 13;
  const long long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
                  F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
  long long z2 = s[2], z3 = s[6];
  long long z1 = (z2 + z3) * F0541;
  long long tmp2 = z1 + z3 * (-F1847);
  long long tmp3 = z1 + z2 * F0765;
  z2 = s[0]; z3 = s[4];
  long long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
  const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = s[7]; tmp1 = s[5]; tmp2 = s[3]; tmp3 = s[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  long long z4 = tmp1 + tmp3;
  const long long z5 = (z3 + z4) * F1175;
  tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
  z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  o[0] = tmp10 + tmp3; o[1] = tmp11 + tmp2; o[2] = tmp12 + tmp1; o[3] = tmp13 + tmp0;
  o[4] = tmp13 - tmp0; o[5] = tmp12 - tmp1; o[6] = tmp11 - tmp2; o[7] = tmp10 - tmp3;
}

}  // namespace

// 64 threads = 8 blocks x 8 lanes: pass 1 works on columns, pass 2 on rows, through an LDS workspace per block.
__global__ void __launch_bounds__(256) jpeg_idct_kernel(const JpegPlaneDesc* __restrict__ descs, int ndesc) {
  __shared__ int ws[32][64 + 8];
  // which plane does this workgroup belong to?  (block ranges are given as a prefix sum over the planes)
  const long gblock = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
  int pi = 0, hi = ndesc - 1;   // last plane whose first_block <= gblock
  while (pi < hi) {
    const int mid = (pi + hi + 1) >> 1;
    if (descs[mid].first_block <= gblock) pi = mid; else hi = mid - 1;
  }
  const JpegPlaneDesc d = descs[pi];
  const long b = gblock - d.first_block;
  const bool live = b < (long)d.bw * d.bh;
  const int lb = threadIdx.x >> 3, i = threadIdx.x & 7;
  const long bb = live ? b : 0;
  const int16_t* coef = d.coef + bb * 64;
  long long s[8], o[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = (long long)((int)coef[8 * k + i] * (int)d.quant[8 * k + i]);  // column i, dequantised
  idct8(s, o);
#pragma unroll
  for (int k = 0; k < 8; ++k) ws[lb][8 * k + i] = jdescale(o[k], 13 - 2);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = ws[lb][8 * i + k];  // row i
  idct8(s, o);
  if (!live) return;
  const int by = (int)(b / d.bw), bx = (int)(b - (long)by * d.bw);
  uint8_t* out = d.plane + ((size_t)by * 8 + i) * ((size_t)d.bw * 8) + (size_t)bx * 8;
  uint8_t px[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) px[k] = jclamp8(jdescale(o[k], 13 + 2 + 3) + 128);
  *(uint2*)out = *(const uint2*)px;  // rows of a plane are 8-byte aligned (bw * 8 wide, planes 256-byte aligned)
}

namespace {

// jdsample.c fancy upsampling of one chroma sample position (x, y) of the full-resolution image
__device__ __forceinline__ int chroma_at(const uint8_t* plane, int stride, int dw, int dh, int hs, int vs, int x, int y) {
  if (hs == 1 && vs == 1) return plane[(size_t)y * stride + x];
  if (vs == 1) {  // h2v1
    const uint8_t* in = plane + (size_t)y * stride;
    if (dw == 1) return in[0];
    const int xi = x >> 1;
    if (x == 0) return in[0];
    if (x == 2 * dw - 1) return in[dw - 1];
    if (x & 1) return (in[xi] * 3 + in[xi + 1] + 2) >> 2;
    return (in[xi] * 3 + in[xi - 1] + 1) >> 2;
  }
  // h2v2: rows 2r, 2r+1 from row r and its upper / lower neighbour (edges replicated)
  const int r = y >> 1;
  const int rn = (y & 1) ? (r + 1 < dh ? r + 1 : dh - 1) : (r > 0 ? r - 1 : 0);
  const uint8_t* in0 = plane + (size_t)r * stride;
  const uint8_t* in1 = plane + (size_t)rn * stride;
  const int xi = x >> 1;
  const int c = in0[xi] * 3 + in1[xi];
  if (dw == 1) return (x & 1) ? (c * 4 + 7) >> 4 : (c * 4 + 8) >> 4;
  if (x == 0) return (c * 4 + 8) >> 4;
  if (x == 2 * dw - 1) return (c * 4 + 7) >> 4;
  if (x & 1) { const int cn = in0[xi + 1] * 3 + in1[xi + 1]; return (c * 3 + cn + 7) >> 4; }
  const int cp = in0[xi - 1] * 3 + in1[xi - 1];
  return (c * 3 + cp + 8) >> 4;
}

}  // namespace

// one thread per output pixel of one image
__global__ void __launch_bounds__(256) jpeg_output_kernel(const JpegImageDesc* __restrict__ imgs, int nimg) {
  const int ii = blockIdx.y;
  const JpegImageDesc im = imgs[ii];
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)im.rows * im.cols) return;
  const int y = (int)(t / im.cols), x = (int)(t - (long)y * im.cols);
  const int Y = im.plane[0][(size_t)y * im.stride[0] + x];
  uint8_t* o = im.bgr + ((size_t)y * im.cols + x) * 3;
  if (im.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
  const int cb = chroma_at(im.plane[1], im.stride[1], im.dw[1], im.dh[1], im.hmax, im.vmax, x, y);
  const int cr = chroma_at(im.plane[2], im.stride[2], im.dw[2], im.dh[2], im.hmax, im.vmax, x, y);
  const long long xb = cb - 128, xr = cr - 128;
  const int crr = (int)((91881LL * xr + 32768) >> 16);
  const int cbb = (int)((116130LL * xb + 32768) >> 16);
  const long long crg = -46802LL * xr, cbg = -22554LL * xb + 32768;
  o[2] = jclamp8(Y + crr);
  o[1] = jclamp8(Y + (int)((cbg + crg) >> 16));
  o[0] = jclamp8(Y + cbb);
}

void launch_jpeg_idct(const JpegPlaneDesc* descs, int ndesc, long total_blocks, hipStream_t s) {
  if (total_blocks <= 0) return;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((total_blocks + 31) / 32)), dim3(256), 0, s, descs, ndesc);
}
void launch_jpeg_output(const JpegImageDesc* imgs, int nimg, long max_pixels, hipStream_t s) {
  if (nimg <= 0 || max_pixels <= 0) return;
  hipLaunchKernelGGL(jpeg_output_kernel, dim3((unsigned)((max_pixels + 255) / 256), (unsigned)nimg), dim3(256), 0, s, imgs, nimg);
}

}  // namespace ocr
