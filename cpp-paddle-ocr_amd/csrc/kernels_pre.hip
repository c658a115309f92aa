// Image-side kernels: OpenCV-semantics bilinear resize (8-bit fixed point), normalisation via a
// 3x256 LUT, ROI crops, in-place 180-degree rotation, greedy CTC collapse.
// Replaces /root/reference/src/preprocess_op.cpp:40-137 (+ cv::resize / convertTo / copyMakeBorder),
// the ROI views and cv::rotate of /root/reference/src/ocr_worker.cpp:245-281 and the CTC loop of
// /root/reference/src/ocr_rec.cpp:97-128.  Pure HBM-bound byte work: one thread per output pixel,
// coalesced along x, no LDS.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels_pre.h"

namespace ocr {

// cv::resize INTER_LINEAR coefficient for destination index d: source index, a0, a1.
__device__ __forceinline__ void lin_coef_x(int d, int ssize, double scale, int& s, int& a0, int& a1, bool& edge) {
  float f = (float)((d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= s;
  edge = false;
  if (s < 0) { f = 0; s = 0; }
  if (s + 1 >= ssize) {
    edge = true;  // dx >= xmax: HResize takes S[sx]*ONE
    if (s >= ssize - 1) { f = 0; s = ssize - 1; }
  }
  a0 = __float2int_rn((1.f - f) * 2048.f);
  a1 = __float2int_rn(f * 2048.f);
  a0 = min(32767, max(-32768, a0));
  a1 = min(32767, max(-32768, a1));
}
__device__ __forceinline__ void lin_coef_y(int d, double scale, int& s, int& b0, int& b1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= s;
  b0 = __float2int_rn((1.f - f) * 2048.f);
  b1 = __float2int_rn(f * 2048.f);
  b0 = min(32767, max(-32768, b0));
  b1 = min(32767, max(-32768, b1));
}
__device__ __forceinline__ int clipi(int v, int lo, int hi) { return v >= lo ? (v < hi ? v : hi - 1) : lo; }

// one resized BGR pixel of an (sh x sw) source at (dy, dx) of a (dh x dw) destination; scale_x/y = 1 / ((double)d / s)
__device__ __forceinline__ void resize_px_scaled(const uint8_t* __restrict__ src, size_t stride, int sh, int sw, int dh, int dw,
                                                 double scale_x, double scale_y, int dy, int dx, uint8_t out[3]) {
  if (sh == dh && sw == dw) {
    const uint8_t* p = src + (size_t)dy * stride + dx * 3;
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    return;
  }
  if (sw == 2 * dw && sh == 2 * dh) {  // exact 2x2 decimation: INTER_LINEAR silently becomes INTER_AREA
    const uint8_t* s0 = src + (size_t)(2 * dy) * stride + (2 * dx) * 3;
    const uint8_t* s1 = s0 + stride;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (uint8_t)((s0[c] + s0[3 + c] + s1[c] + s1[3 + c] + 2) >> 2);
    return;
  }
  int sx, a0, a1, sy, b0, b1;
  bool edge;
  lin_coef_x(dx, sw, scale_x, sx, a0, a1, edge);
  lin_coef_y(dy, scale_y, sy, b0, b1);
  const int y0 = clipi(sy, 0, sh), y1 = clipi(sy + 1, 0, sh);
  const uint8_t* r0 = src + (size_t)y0 * stride + sx * 3;
  const uint8_t* r1 = src + (size_t)y1 * stride + sx * 3;
  // the two source pixels of a row = six consecutive bytes: ONE unaligned 8-byte load when those eight bytes lie inside
  // the source row (sx <= sw - 3), byte loads at the row's end.  The texture-address unit takes as long for a wave's
  // byte load as for its 8-byte load, and twelve byte loads per pixel were what bounded the kernel.
  unsigned long long q0, q1;
  if (sx + 3 <= sw) {
    __builtin_memcpy(&q0, r0, 8);
    __builtin_memcpy(&q1, r1, 8);
  } else {
    q0 = (unsigned long long)r0[0] | ((unsigned long long)r0[1] << 8) | ((unsigned long long)r0[2] << 16);
    q1 = (unsigned long long)r1[0] | ((unsigned long long)r1[1] << 8) | ((unsigned long long)r1[2] << 16);
    if (!edge) {
      q0 |= ((unsigned long long)r0[3] << 24) | ((unsigned long long)r0[4] << 32) | ((unsigned long long)r0[5] << 40);
      q1 |= ((unsigned long long)r1[3] << 24) | ((unsigned long long)r1[4] << 32) | ((unsigned long long)r1[5] << 40);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int p00 = (int)((q0 >> (8 * c)) & 0xffu), p01 = (int)((q0 >> (8 * (3 + c))) & 0xffu);
    const int p10 = (int)((q1 >> (8 * c)) & 0xffu), p11 = (int)((q1 >> (8 * (3 + c))) & 0xffu);
    int S0, S1;
    if (!edge) {
      S0 = p00 * a0 + p01 * a1;
      S1 = p10 * a0 + p11 * a1;
    } else {
      S0 = p00 * 2048;
      S1 = p10 * 2048;
    }
    const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    out[c] = (uint8_t)min(255, max(0, v));
  }
}

__device__ __forceinline__ void resize_px(const uint8_t* __restrict__ src, size_t stride, int sh, int sw, int dh, int dw,
                                          int dy, int dx, uint8_t out[3]) {
  const double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
  resize_px_scaled(src, stride, sh, sw, dh, dw, 1. / inv_x, 1. / inv_y, dy, dx, out);
}

// det: N same-size images -> resized u8 (tap) + normalised f32 NHWC
// PX adjacent output pixels per thread (PX = 4 when the row length allows): the kernel is bound by a thread's chain
// source bytes -> table -> store times the number of thread waves it takes, so all PX pixels' source bytes are fetched
// before the first store (no store between them: byte pointers alias everything), the normalisation table sits in LDS
// instead of behind a second L2 round trip, and a thread's PX * 12 output bytes leave as 16-byte stores.
template <int PX>
__global__ void __launch_bounds__(256) det_pre_kernel(const DetPreArgs a) {
  __shared__ float s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = a.lut[i];
  __syncthreads();
  const long t = ((long)blockIdx.x * 256 + threadIdx.x) * PX;
  const long per = (long)a.dh * a.dw;
  if (t >= per * a.N) return;
  const int n = (int)(t / per);
  const int r = (int)(t - (long)n * per);
  const int dy = r / a.dw, dx = r - dy * a.dw;  // (dw % PX == 0: the PX pixels are in one row)
  uint8_t px[PX][3];
  const uint8_t* img = a.src + (size_t)n * a.src_image_bytes;
  bool done = false;
  if constexpr (PX == 4) {
    // same-size images (a 960x960 card at limit_side_len 960): the four pixels are twelve consecutive source bytes -
    // three dword loads when the address is 4-byte aligned (it is for packed rows whose length is a multiple of 4)
    const uint8_t* p = img + (size_t)dy * a.src_stride + (size_t)dx * 3;
    if (a.sh == a.dh && a.sw == a.dw && (((size_t)p) & 3) == 0) {
      unsigned w3[3];
      __builtin_memcpy(w3, __builtin_assume_aligned(p, 4), 12);
#pragma unroll
      for (int b = 0; b < 12; ++b) px[b / 3][b % 3] = (uint8_t)(w3[b >> 2] >> (8 * (b & 3)));
      done = true;
    }
  }
  if (!done) {
#pragma unroll
    for (int u = 0; u < PX; ++u) resize_px(img, a.src_stride, a.sh, a.sw, a.dh, a.dw, dy, dx + u, px[u]);
  }
  float v[PX * 3];
#pragma unroll
  for (int u = 0; u < PX; ++u) {
    v[3 * u] = s_lut[px[u][0]];
    v[3 * u + 1] = s_lut[256 + px[u][1]];
    v[3 * u + 2] = s_lut[512 + px[u][2]];
  }
  float* o = a.out + t * 3;
  if constexpr (PX == 4) {
#pragma unroll
    for (int q = 0; q < 3; ++q) *(float4*)(o + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  } else {
#pragma unroll
    for (int q = 0; q < PX * 3; ++q) o[q] = v[q];
  }
  if (a.resized) {
    uint8_t* u8 = a.resized + t * 3;
    if constexpr (PX == 4) {  // (t % 4 == 0: the twelve bytes start 4-byte aligned when the tap buffer is)
      if ((((size_t)u8) & 3) == 0) {
        unsigned w3[3] = {0u, 0u, 0u};
#pragma unroll
        for (int b = 0; b < 12; ++b) w3[b >> 2] |= (unsigned)px[b / 3][b % 3] << (8 * (b & 3));
        __builtin_memcpy(__builtin_assume_aligned(u8, 4), w3, 12);
        return;
      }
    }
#pragma unroll
    for (int u = 0; u < PX; ++u) { u8[3 * u] = px[u][0]; u8[3 * u + 1] = px[u][1]; u8[3 * u + 2] = px[u][2]; }
  }
}
void launch_det_pre(const DetPreArgs& a, hipStream_t s) {
  const long total = (long)a.N * a.dh * a.dw;
  if (a.dw % 4 == 0) hipLaunchKernelGGL(det_pre_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(det_pre_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// rec / cls: one text line = ROI of a device image -> [imgH][imgW][3] f32 slot of the batch tensor
template <int PX>
__global__ void __launch_bounds__(256) line_pre_kernel(const LineDesc* __restrict__ lines, int nlines, int imgH, int imgW,
                                                       const float* __restrict__ lut, float pad_value,
                                                       float* __restrict__ out) {
  // (as det_pre_kernel: PX adjacent pixels per thread, their source bytes fetched before the first store, table in LDS)
  __shared__ float s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  const long t = ((long)blockIdx.x * 256 + threadIdx.x) * PX;
  const long per = (long)imgH * imgW;
  if (t >= per * nlines) return;
  const int li = (int)(t / per);
  const int r = (int)(t - (long)li * per);
  const int dy = r / imgW, dx = r - dy * imgW;  // (imgW % PX == 0: the PX pixels are in one row)
  const LineDesc L = lines[li];
  float* o = out + ((size_t)L.slot * per + r) * 3;
  uint8_t px[PX][3];
#pragma unroll
  for (int u = 0; u < PX; ++u) {
    const int x = dx + u < L.resize_w ? dx + u : L.resize_w - 1;  // pad columns: a valid pixel is read and dropped
    resize_px_scaled(L.img + (size_t)L.y * L.stride + (size_t)L.x * 3, L.stride, L.h, L.w, imgH, L.resize_w, L.scale_x, L.scale_y, dy, x,
                     px[u]);
  }
  float v[PX * 3];
#pragma unroll
  for (int u = 0; u < PX; ++u) {
    const bool pad = dx + u >= L.resize_w;
    // rec: u8 zero pad BEFORE normalise (lut[0]); cls: 0.0f AFTER normalise -> pad_value selects
    const bool zero = pad && pad_value == 0.0f;
    const float v0 = s_lut[pad ? 0 : px[u][0]], v1 = s_lut[256 + (pad ? 0 : px[u][1])], v2 = s_lut[512 + (pad ? 0 : px[u][2])];
    v[3 * u] = zero ? 0.f : v0;
    v[3 * u + 1] = zero ? 0.f : v1;
    v[3 * u + 2] = zero ? 0.f : v2;
  }
  if constexpr (PX == 4) {
#pragma unroll
    for (int q = 0; q < 3; ++q) *(float4*)(o + 4 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  } else {
#pragma unroll
    for (int q = 0; q < PX * 3; ++q) o[q] = v[q];
  }
}
void launch_line_pre(const LineDesc* lines, int nlines, int imgH, int imgW, const float* lut, bool pad_after_norm,
                     float* out, hipStream_t s) {
  const long total = (long)nlines * imgH * imgW;
  if (total == 0) return;
  if (imgW % 4 == 0)
    hipLaunchKernelGGL(line_pre_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, lines, nlines, imgH, imgW, lut,
                       pad_after_norm ? 0.0f : 1.0f, out);
  else
    hipLaunchKernelGGL(line_pre_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, lines, nlines, imgH, imgW, lut,
                       pad_after_norm ? 0.0f : 1.0f, out);
}

// Ragged batch (rec): every line has its own tensor width (kernels_net.h, RagLevel); one thread per output pixel, the
// line found from the pixel index (lines[i].pix0 ascending).  Same per-pixel arithmetic as line_pre_kernel; the pad is
// the recognizer's (u8 zero BEFORE normalisation, /root/reference/src/preprocess_op.cpp:115-117).
__global__ void __launch_bounds__(256) line_pre_ragged_kernel(const LineDesc* __restrict__ lines, int nlines, long total, int imgH,
                                                              const float* __restrict__ lut, float* __restrict__ out) {
  __shared__ float s_lut[768];
  for (int i = threadIdx.x; i < 768; i += 256) s_lut[i] = lut[i];
  __syncthreads();
  const long t0 = (long)blockIdx.x * 256, t = t0 + threadIdx.x;
  if (t >= total) return;
  int lo = 0, hi = nlines;  // the workgroup's first pixel (uniform), then forward: a workgroup spans a line or two
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((long)lines[mid].pix0 <= t0) lo = mid;
    else hi = mid;
  }
  while (lo + 1 < nlines && (long)lines[lo + 1].pix0 <= t) ++lo;
  const LineDesc L = lines[lo];
  const int r = (int)(t - L.pix0);
  const int dy = r / L.tensor_w, dx = r - dy * L.tensor_w;
  const bool pad = dx >= L.resize_w;
  uint8_t px[3];
  resize_px_scaled(L.img + (size_t)L.y * L.stride + (size_t)L.x * 3, L.stride, L.h, L.w, imgH, L.resize_w, L.scale_x, L.scale_y, dy,
                   pad ? L.resize_w - 1 : dx, px);  // pad columns: a valid pixel is read and dropped
  float* o = out + t * 3;
  o[0] = s_lut[pad ? 0 : px[0]];
  o[1] = s_lut[256 + (pad ? 0 : px[1])];
  o[2] = s_lut[512 + (pad ? 0 : px[2])];
}
void launch_line_pre_ragged(const LineDesc* lines, int nlines, long total_pixels, int imgH, const float* lut, float* out,
                            hipStream_t s) {
  if (total_pixels <= 0) return;
  hipLaunchKernelGGL(line_pre_ragged_kernel, dim3((unsigned)((total_pixels + 255) / 256)), dim3(256), 0, s, lines, nlines, total_pixels,
                     imgH, lut, out);
}

// cv::rotate(roi, roi, ROTATE_180) in place: pixel i <-> total-1-i
__global__ void __launch_bounds__(256) rotate180_kernel(uint8_t* img, size_t stride, int x0, int y0, int w, int h) {
  const long total = (long)w * h;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total / 2) return;
  const long j = total - 1 - i;
  uint8_t* a = img + (size_t)(y0 + i / w) * stride + (size_t)(x0 + i % w) * 3;
  uint8_t* b = img + (size_t)(y0 + j / w) * stride + (size_t)(x0 + j % w) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) { const uint8_t t = a[c]; a[c] = b[c]; b[c] = t; }
}
void launch_rotate180(uint8_t* img, size_t stride, int x0, int y0, int w, int h, hipStream_t s) {
  const long half = ((long)w * h) / 2;
  if (half <= 0) return;
  hipLaunchKernelGGL(rotate180_kernel, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, s, img, stride, x0, y0, w, h);
}

// The same for a whole batch.  ROIs of one image may overlap, so their order matters - but only between ROIs that do
// overlap: the host groups the batch into the connected components of the "intersects" relation (pipe.hip), one
// workgroup applies the rotations of ONE component in request order, components run concurrently.  Pixel (y, x)
// swaps with (h-1-y, w-1-x) for the first total/2 pixels in raster order (cv::rotate's flip).
// one BGR pixel as the low 24 bits of a word.  A pixel that is not the last of its ROI row is fetched with ONE unaligned
// 4-byte load (the fourth byte is the next pixel's first, inside the ROI); the TA processes a wave's byte loads no
// faster than its dword loads, and the kernel is bound by their number.
__device__ __forceinline__ unsigned load_px(const uint8_t* p, bool wide) {
  if (wide) {
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v & 0xffffffu;
  }
  return (unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16);
}
__device__ __forceinline__ void store_px(uint8_t* p, unsigned v) {
  const unsigned short lo = (unsigned short)(v & 0xffffu);
  __builtin_memcpy(p, &lo, 2);
  p[2] = (uint8_t)(v >> 16);
}
__global__ void __launch_bounds__(1024) rotate180_groups_kernel(const RotDesc* __restrict__ d, const int* __restrict__ seg) {
  // 16 rows x 64 columns of threads; a thread's P pixels of a row (x = tx + 64 u) are all loaded, with their mirror
  // pixels, before the first store: no divisions, one memory round trip per 16 x 512 pixel block
  constexpr int P = 8;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = seg[blockIdx.x]; r < seg[blockIdx.x + 1]; ++r) {
    const RotDesc q = d[r];
    const int half_rows = q.h >> 1, rows = half_rows + (q.h & 1);
    uint8_t* base = q.img + (size_t)q.y * q.stride + (size_t)q.x * 3;
    for (int y = ty; y < rows; y += 16) {
      const int xe = y < half_rows ? q.w : q.w >> 1;  // the middle row of an odd height: its first half
      uint8_t* ra = base + (size_t)y * q.stride;
      uint8_t* rb = base + (size_t)(q.h - 1 - y) * q.stride;
      for (int x0 = tx; x0 < xe; x0 += 64 * P) {
        unsigned va[P], vb[P];
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int x = x0 + 64 * u, xc = x < xe ? x : 0;  // past the end: pixel 0 is read, nothing is written
          va[u] = load_px(ra + (size_t)xc * 3, xc + 1 < q.w);
          vb[u] = load_px(rb + (size_t)(q.w - 1 - xc) * 3, xc > 0);
        }
#pragma unroll
        for (int u = 0; u < P; ++u) {
          const int x = x0 + 64 * u;
          if (x < xe) {
            store_px(ra + (size_t)x * 3, vb[u]);
            store_px(rb + (size_t)(q.w - 1 - x) * 3, va[u]);
          }
        }
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}
void launch_rotate180_groups(const RotDesc* d, const int* seg, int ngroups, hipStream_t s) {
  if (ngroups <= 0) return;
  hipLaunchKernelGGL(rotate180_groups_kernel, dim3(ngroups), dim3(1024), 0, s, d, seg);
}

// cv::warpPerspective(crop, M, INTER_LINEAR, BORDER_CONSTANT 0) restated per pixel
// (/root/reference/src/utility.cpp:175-180; the reference's BORDER_REPLICATE lands in the flags slot
// and therefore selects INTER_LINEAR).  The source coordinate is evaluated in double exactly as
// WarpPerspectiveInvoker does: base terms at the start of the pixel's bw0-column block, the in-block
// offset added afterwards, scaled by 32/W, rounded to nearest-even, split into integer and 1/32 parts.
__global__ void __launch_bounds__(256) warp_crop_kernel(const WarpDesc* __restrict__ descs) {
  const WarpDesc& q = descs[blockIdx.y];
  const int dw = q.dw, dh = q.dh;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)dw * dh) return;
  const int y = (int)(t / dw), x = (int)(t - (long)y * dw);
  const int xb = (x / q.bw0) * q.bw0, x1 = x - xb;
  const double X0 = q.m[0] * xb + q.m[1] * y + q.m[2];
  const double Y0 = q.m[3] * xb + q.m[4] * y + q.m[5];
  const double W0 = q.m[6] * xb + q.m[7] * y + q.m[8];
  double W = W0 + q.m[6] * x1;
  W = W != 0. ? 32. / W : 0.;
  double fX = (X0 + q.m[0] * x1) * W, fY = (Y0 + q.m[3] * x1) * W;
  fX = fX < 2147483647.0 ? fX : 2147483647.0;   // std::min(INT_MAX, v) then std::max(INT_MIN, .)
  fX = -2147483648.0 < fX ? fX : -2147483648.0;
  fY = fY < 2147483647.0 ? fY : 2147483647.0;
  fY = -2147483648.0 < fY ? fY : -2147483648.0;
  const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
  const int sx = min(32767, max(-32768, X >> 5)), sy = min(32767, max(-32768, Y >> 5));
  const int fx = X & 31, fy = Y & 31;
  const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
  uint8_t* D = q.dst + (q.rot ? ((size_t)(dw - 1 - x) * dh + y) * 3 : ((size_t)y * dw + x) * 3);
  if (sx >= q.sw || sx + 1 < 0 || sy >= q.sh || sy + 1 < 0) {
    D[0] = 0; D[1] = 0; D[2] = 0;
    return;
  }
  const bool xin0 = sx >= 0, xin1 = sx + 1 < q.sw, yin0 = sy >= 0, yin1 = sy + 1 < q.sh;
  const uint8_t* r0 = q.src + (long)sy * (long)q.sstride + (long)sx * 3;
  const uint8_t* r1 = r0 + q.sstride;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v00 = (yin0 && xin0) ? r0[c] : 0, v01 = (yin0 && xin1) ? r0[3 + c] : 0;
    const int v10 = (yin1 && xin0) ? r1[c] : 0, v11 = (yin1 && xin1) ? r1[3 + c] : 0;
    const int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    D[c] = (uint8_t)min(255, max(0, v));
  }
}
void launch_warp_crops(const WarpDesc* d, int ncrops, int max_pixels, hipStream_t s) {
  if (max_pixels <= 0) return;
  for (int at = 0; at < ncrops; at += 65535)  // grid.y limit
    hipLaunchKernelGGL(warp_crop_kernel, dim3((unsigned)((max_pixels + 255) / 256), (unsigned)std::min(65535, ncrops - at)), dim3(256),
                       0, s, d + at);
}

// greedy CTC collapse, one thread per line (sequential over T: the score sum order of the reference)
__global__ void __launch_bounds__(64) ctc_kernel(const int* __restrict__ amax, const float* __restrict__ pmax, int nlines,
                                                 int T, int max_len, int* __restrict__ ids, int* __restrict__ lens,
                                                 float* __restrict__ scores, const LineDesc* __restrict__ lines) {
  const int li = blockIdx.x * 64 + threadIdx.x;
  if (li >= nlines) return;
  long first = (long)li * T;
  if (lines) { first = lines[li].step0; T = lines[li].steps; }  // ragged batch: the line's own step count
  const int* am = amax + first;
  const float* pm = pmax + first;
  int last = 0, count = 0;
  float s = 0.f;
  for (int n = 0; n < T; ++n) {
    const int idx = am[n];
    if (idx > 0 && !(n > 0 && idx == last)) {
      s += pm[n];
      if (count < max_len) ids[(long)li * max_len + count] = idx;
      count += 1;
    }
    last = idx;
  }
  lens[li] = count;
  scores[li] = count > 0 ? s / (float)count : 0.f;
}
void launch_ctc(const int* amax, const float* pmax, int nlines, int T, int max_len, int* ids, int* lens, float* scores,
                hipStream_t s) {
  if (nlines <= 0) return;
  hipLaunchKernelGGL(ctc_kernel, dim3((nlines + 63) / 64), dim3(64), 0, s, amax, pmax, nlines, T, max_len, ids, lens, scores,
                     (const LineDesc*)nullptr);
}
void launch_ctc_ragged(const int* amax, const float* pmax, const LineDesc* lines, int nlines, int max_len, int* ids, int* lens,
                       float* scores, hipStream_t s) {
  if (nlines <= 0) return;
  hipLaunchKernelGGL(ctc_kernel, dim3((nlines + 63) / 64), dim3(64), 0, s, amax, pmax, nlines, 0, max_len, ids, lens, scores, lines);
}

}  // namespace ocr
