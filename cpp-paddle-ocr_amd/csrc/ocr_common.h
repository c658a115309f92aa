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
enum : int { EP_BIAS = 0, EP_SMUL, EP_SADD, EP_BN, EP_ACT, EP_MULC, EP_ADDT, EP_ADDUP, EP_GATERES, EP_SFMA };  // GATERES: v = v * gate[n][c] + v (an `ew mulc | addt self` folded into its producer); SFMA: v = fmaf(v, p0, p1) (the LAB fold, net.hip fold_lab)
// HSW6: u = y * clamp(y + 3, 0, 6), the hard-swish's product - its 1/6 travels with the scale that follows (fold_lab)
enum : int { ACT_RELU = 0, ACT_HSWISH, ACT_HSIG, ACT_SWISH, ACT_SIGMOID, ACT_HSW6 };

struct EpStage {
  int kind;
  int act;
  float p0, p1;
  const float* v0;  // BIAS: b[Cs] / BN: s[Cs] / MULC, GATERES: gate [N][Cs] / ADDT, ADDUP: tensor
  const float* v1;  // BN: t[Cs]
  int a0, a1, a2;   // ADDUP: a0 = scale, a1 = source W, a2 = source H
};
#define OCR_MAX_EP 8
struct Epilogue {
  int n;
  EpStage st[OCR_MAX_EP];
};

#if defined(__HIPCC__)
// ---- hard-swish without the IEEE division sequence ---------------------------------------------------
// u / 6.0f costs 13 VALU slots per value (two v_div_scale, a quarter-rate v_rcp, five fma, v_div_fmas,
// v_div_fixup), and VALU work is not hidden behind another wave's MFMAs on this chip
// (tools/micro/mfma_valu.hip: the two add up), so epilogue instructions cost matrix time one for one.
// With r = RN(1/6): q0 = RN(u*r), e = fma(-6, q0, u) is the exact residual and RN(q0 + e*r) is the
// correctly rounded quotient whenever q0 is a normal number (Markstein); copysign restores -0.  Checked
// against u / 6.0f over all 2^32 bit patterns (tools/check_div6.c, run by tests/test_identities.py): the only inputs that differ are those
// whose q0 is denormal or not finite (exact ties exist only among denormal quotients).  For
// u = y * clamp(y + 3, 0, 6) that cannot happen while 2^-119 <= |y| < 2^125 (then u = +-0 or
// 2^-118 <= |u| < 2^128): a sweep first takes min/max of |y| over its values (one v_min3/v_max3 pair per
// two values) and falls back to the division for anything outside, zero included.  NaN needs no guard: it
// comes out as NaN either way.
typedef float ocr_f2 __attribute__((ext_vector_type(2)));
#define OCR_HSW_LO 0x1p-119f
#define OCR_HSW_HI 0x1p+125f
__device__ __forceinline__ void ocr_absrange(float& mn, float& mx, float a, float b) {
  asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(mn) : "v"(a), "v"(b));
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(a), "v"(b));
}
__device__ __forceinline__ bool ocr_hsw_fast_ok(float mn, float mx) { return mn >= OCR_HSW_LO && mx < OCR_HSW_HI; }
__device__ __forceinline__ ocr_f2 ocr_hswish2_fast(ocr_f2 y) {
  const ocr_f2 three = {3.0f, 3.0f};
  const ocr_f2 r = {0x1.555556p-3f, 0x1.555556p-3f};  // RN(1/6) = 0x3e2aaaab
  const ocr_f2 m6 = {-6.0f, -6.0f};
  ocr_f2 t = y + three;
  t.x = fminf(fmaxf(t.x, 0.0f), 6.0f);
  t.y = fminf(fmaxf(t.y, 0.0f), 6.0f);
  const ocr_f2 u = y * t;
  const ocr_f2 q0 = u * r;
  const ocr_f2 e = __builtin_elementwise_fma(m6, q0, u);
  ocr_f2 q = __builtin_elementwise_fma(e, r, q0);
  q.x = __builtin_copysignf(q.x, u.x);
  q.y = __builtin_copysignf(q.y, u.y);
  return q;
}
// u = y * clamp(y + 3, 0, 6) on a packed pair: v_pk_add, two v_med3, v_pk_mul
__device__ __forceinline__ ocr_f2 ocr_hsw6_2(ocr_f2 y) {
  const ocr_f2 three = {3.0f, 3.0f};
  ocr_f2 t = y + three;
  t.x = __builtin_amdgcn_fmed3f(t.x, 0.0f, 6.0f);
  t.y = __builtin_amdgcn_fmed3f(t.y, 0.0f, 6.0f);
  return y * t;
}
// four values whose |y| range has been checked with ocr_hsw_fast_ok
__device__ __forceinline__ void ocr_hswish4_fast(float& a, float& b, float& c, float& d) {
  const ocr_f2 q0 = ocr_hswish2_fast(ocr_f2{a, b}), q1 = ocr_hswish2_fast(ocr_f2{c, d});
  a = q0.x; b = q0.y; c = q1.x; d = q1.y;
}
__device__ __forceinline__ float ocr_hswish_div(float y) {
  float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f);
  float u = y * t;
  return u / 6.0f;
}
// self-contained form for code that handles one quad at a time
__device__ __forceinline__ void ocr_hswish4(float& a, float& b, float& c, float& d) {
  float mn = INFINITY, mx = 0.0f;
  ocr_absrange(mn, mx, a, b);
  ocr_absrange(mn, mx, c, d);
  if (ocr_hsw_fast_ok(mn, mx)) ocr_hswish4_fast(a, b, c, d);
  else { a = ocr_hswish_div(a); b = ocr_hswish_div(b); c = ocr_hswish_div(c); d = ocr_hswish_div(d); }
}
#endif

OCR_HD float ocr_act(int act, float p0, float p1, float y) {
  switch (act) {
    case ACT_RELU: return fmaxf(y, 0.0f);
    case ACT_HSWISH: { float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f); float u = y * t; return u / 6.0f; }
    case ACT_HSIG: { float t = y * p0; t = t + p1; return fminf(fmaxf(t, 0.0f), 1.0f); }
    case ACT_SWISH: { float e = ocr_expf(-y); float d = 1.0f + e; return y / d; }
    case ACT_HSW6: { float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f); return y * t; }
    default: { float e = ocr_expf(-y); float d = 1.0f + e; return 1.0f / d; }
  }
}
