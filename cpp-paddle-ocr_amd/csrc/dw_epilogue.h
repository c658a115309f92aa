// The depthwise kernels' epilogue: the plan's generic stage list on a register patch of R x TO output pixels x 4 physical
// channels (kernels_net.hip dw_conv_kernel, kernels_dwlds.hip dw_lds_kernel).  Stage loop outside, the patch inside: one
// copy of each stage's code.  Every stage is its own rounding, as ocr_common.h's scalar forms (v_pk_* = two IEEE lanes).
#pragma once
#include "conv_device.h"

namespace ocr {

struct DwF4 { ocr_f2 lo, hi; };

// pc: the thread's first physical channel; n: its image (per-image gate vectors); obase / orow: element index of the patch's
// first output and the output row stride (the residual operand of an ADDT stage has the output's shape; H16: stored as f16);
// outputs at (y0 + r, x0 + o) beyond (OHn, OW) do not exist (their residual is not read)
template <int R, int TO, bool H16>
__device__ __forceinline__ void dw_patch_epilogue(DwF4 (&acc)[R][TO], const Epilogue& ep, int pc, int n, int Cs, long obase, long orow,
                                                  int y0, int x0, int OHn, int OW) {
  using F4 = DwF4;
  for (int s = 0; s < ep.n; ++s) {
    const EpStage& st = ep.st[s];
#define OCR_DW_SWEEP(BODY)                                       \
  _Pragma("unroll") for (int r = 0; r < R; ++r)                   \
    _Pragma("unroll") for (int o = 0; o < TO; ++o) {              \
      F4& v = acc[r][o];                                          \
      BODY                                                        \
    }
    switch (st.kind) {
      case EP_BIAS: {
        const float4 b = *(const float4*)(st.v0 + pc);
        const ocr_f2 blo = {b.x, b.y}, bhi = {b.z, b.w};
        OCR_DW_SWEEP({ v.lo = v.lo + blo; v.hi = v.hi + bhi; })
      } break;
      case EP_SMUL: {
        const ocr_f2 k = {st.p0, st.p0};
        OCR_DW_SWEEP({ v.lo = k * v.lo; v.hi = k * v.hi; })
      } break;
      case EP_SADD: {
        const ocr_f2 k = {st.p0, st.p0};
        OCR_DW_SWEEP({ v.lo = v.lo + k; v.hi = v.hi + k; })
      } break;
      case EP_SFMA: {
        const ocr_f2 k = {st.p0, st.p0}, d = {st.p1, st.p1};
        OCR_DW_SWEEP({ v.lo = __builtin_elementwise_fma(v.lo, k, d); v.hi = __builtin_elementwise_fma(v.hi, k, d); })
      } break;
      case EP_BN: {
        const float4 sc = *(const float4*)(st.v0 + pc), sh = *(const float4*)(st.v1 + pc);
        const ocr_f2 clo = {sc.x, sc.y}, chi = {sc.z, sc.w}, hlo = {sh.x, sh.y}, hhi = {sh.z, sh.w};
        OCR_DW_SWEEP({
          ocr_f2 u;
          u = v.lo * clo; v.lo = u + hlo;
          u = v.hi * chi; v.hi = u + hhi;
        })
      } break;
      case EP_ACT: {
        const float p0 = st.p0, p1 = st.p1;
#define OCR_DW_ACT(KIND) OCR_DW_SWEEP({ v.lo.x = ocr_act(KIND, p0, p1, v.lo.x); v.lo.y = ocr_act(KIND, p0, p1, v.lo.y); v.hi.x = ocr_act(KIND, p0, p1, v.hi.x); v.hi.y = ocr_act(KIND, p0, p1, v.hi.y); })
        switch (st.act) {
          case ACT_RELU: OCR_DW_ACT(ACT_RELU) break;
          case ACT_HSWISH: {
            // range pass, then the division-free sweep (ocr_common.h); anything out of range takes the division
            float mn = INFINITY, mx = 0.0f;
            OCR_DW_SWEEP({ ocr_absrange(mn, mx, v.lo.x, v.lo.y); ocr_absrange(mn, mx, v.hi.x, v.hi.y); })
            if (ocr_hsw_fast_ok(mn, mx)) { OCR_DW_SWEEP({ v.lo = ocr_hswish2_fast(v.lo); v.hi = ocr_hswish2_fast(v.hi); }) }
            else { OCR_DW_ACT(ACT_HSWISH) }
          } break;
          case ACT_HSW6: OCR_DW_SWEEP({ v.lo = ocr_hsw6_2(v.lo); v.hi = ocr_hsw6_2(v.hi); }) break;
          case ACT_HSIG: OCR_DW_ACT(ACT_HSIG) break;
          case ACT_SWISH: OCR_DW_ACT(ACT_SWISH) break;
          default: OCR_DW_ACT(ACT_SIGMOID) break;
        }
#undef OCR_DW_ACT
      } break;
      case EP_MULC: {
        const float4 g = *(const float4*)(st.v0 + (long)n * Cs + pc);
        const ocr_f2 glo = {g.x, g.y}, ghi = {g.z, g.w};
        OCR_DW_SWEEP({ v.lo = v.lo * glo; v.hi = v.hi * ghi; })
      } break;
      case EP_ADDT:
        OCR_DW_SWEEP({
          if (y0 + r < OHn && x0 + o < OW) {
            const float4 g = ld4<H16>(st.v0, obase + r * orow + (long)o * Cs);
            ocr_f2 glo;
            ocr_f2 ghi;
            glo.x = g.x; glo.y = g.y; ghi.x = g.z; ghi.y = g.w;
            v.lo = v.lo + glo; v.hi = v.hi + ghi;
          }
        })
        break;
      default: break;  // ADDUP never follows a depthwise conv on this path (host checks)
    }
#undef OCR_DW_SWEEP
  }
}

}  // namespace ocr
