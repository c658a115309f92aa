// Shared definitions for the MI355X OCR runtime (host + device).
//
// Activation layout in HBM ("C8I"): NHWC, channel count padded to a multiple of 8, and inside each
// group of 8 the even logical channels come first: physical = 8*(c/8) + 4*(c%2) + (c%8)/2.
// Why: `v_mfma_f32_32x32x2_f32` takes k = lane>>5 from each lane.  With C8I a lane's single
// 16-byte load (physical channels 4h..4h+3, h = lane>>5) feeds four consecutive MFMA steps whose
// (k=0,k=1) pairs are the logical channels (2s, 2s+1): the hardware's k-ordered fmaf chain is then
// the ascending-k chain of the arithmetic contract with no LDS transpose.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#if defined(__HIPCC__)
#define OCR_HD __host__ __device__ __forceinline__
#else
#define OCR_HD inline
#endif

OCR_HD int c8i_phys(int c) { return (c & ~7) | ((c & 1) << 2) | ((c & 7) >> 1); }
OCR_HD int c8i_logical(int p) { return (p & ~7) | (((p & 3) << 1) | ((p >> 2) & 1)); }
OCR_HD int c8i_stride(int c) { return (c + 7) & ~7; }

// exp with a fixed sequence of f32 roundings (same sequence as the CPU oracle; never libm).
OCR_HD float ocr_expf(float x) {
  x = fminf(fmaxf(x, -87.0f), 88.0f);
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  float r2 = r * r;
  float y = fmaf(p, r2, r);
  y = y + 1.0f;
  int ni = (int)n;
  uint32_t bits = (uint32_t)(ni + 127) << 23;
  float sc;
#if defined(__HIP_DEVICE_COMPILE__)
  sc = __uint_as_float(bits);
#else
  memcpy(&sc, &bits, 4);
#endif
  return y * sc;
}

// ---- epilogue description (kernel argument, by value) ----
enum : int { EP_BIAS = 0, EP_SMUL, EP_SADD, EP_BN, EP_ACT, EP_MULC, EP_ADDT, EP_ADDUP };
enum : int { ACT_RELU = 0, ACT_HSWISH, ACT_HSIG, ACT_SWISH, ACT_SIGMOID };

struct EpStage {
  int kind;
  int act;
  float p0, p1;
  const float* v0;  // BIAS: b[Cs] / BN: s[Cs] / MULC: gate [N][Cs] / ADDT, ADDUP: tensor
  const float* v1;  // BN: t[Cs]
  int a0, a1, a2;   // ADDUP: a0 = scale, a1 = source W, a2 = source H
};
#define OCR_MAX_EP 8
struct Epilogue {
  int n;
  EpStage st[OCR_MAX_EP];
};

OCR_HD float ocr_act(int act, float p0, float p1, float y) {
  switch (act) {
    case ACT_RELU: return fmaxf(y, 0.0f);
    case ACT_HSWISH: { float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f); float u = y * t; return u / 6.0f; }
    case ACT_HSIG: { float t = y * p0; t = t + p1; return fminf(fmaxf(t, 0.0f), 1.0f); }
    case ACT_SWISH: { float e = ocr_expf(-y); float d = 1.0f + e; return y / d; }
    default: { float e = ocr_expf(-y); float d = 1.0f + e; return 1.0f / d; }
  }
}
