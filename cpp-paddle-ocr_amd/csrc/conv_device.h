// Device-side pieces shared by the network kernels (kernels_net.hip, kernels_dwpw.hip): XCD-aware block order,
// pixel decode, the scalar / 4-wide epilogue, and the epilogue + parameter staging of the MFMA conv tiles.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_net.h"

namespace ocr {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ---- precision = "fp16" (DESIGN.md section 9): the matrix-core products in f16 with f32 accumulation.  Activations stay
// f32 C8I in HBM; a lane's 16-byte pixel operand (physical channels 4h..4h+3 of an octet) is rounded to four halfs
// (v_cvt_pk_f16_f32, round to nearest even) and ONE v_mfma_f32_32x32x8_f16 consumes the octet (k = 4h + s on both
// operands) where the f32 path issues four v_mfma_f32_32x32x2_f32.  Weights come from an f16 fragment image with the
// f32 image's indexing, 8 bytes per lane instead of 16.
typedef _Float16 ocr_h4 __attribute__((ext_vector_type(4)));
typedef float ocr_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ocr_h4 ocr_to_h4(const float4& v) { return __builtin_convertvector(ocr_f4v{v.x, v.y, v.z, v.w}, ocr_h4); }
__device__ __forceinline__ ocr_h4 ocr_as_h4(const uint2& v) { return __builtin_bit_cast(ocr_h4, v); }
template <bool HALF> struct WFrag { using T = float4; };
template <> struct WFrag<true> { using T = uint2; };

// ---- f16 STORAGE of the C8I activation tensors (precision "fp16", second half of the mode): the arena pointers stay
// `float*` in every argument block; a kernel instantiated with H16 reads / writes the tensor as _Float16 with the SAME
// element indices (pixel * Cs + physical channel).  Four consecutive channels are 8 bytes.  Arithmetic stays f32: a load
// converts up (exact), a store rounds to nearest even.  Per-image vectors (pool results, SE gates), the network input, the
// detector's probability map and the recognizer's logits stay f32.
template <bool H16>
__device__ __forceinline__ float4 ld4(const float* base, long idx) {
  if constexpr (H16) {
    const ocr_h4 h = __builtin_bit_cast(ocr_h4, *(const uint2*)((const _Float16*)base + idx));
    const ocr_f4v f = __builtin_convertvector(h, ocr_f4v);
    return make_float4(f.x, f.y, f.z, f.w);
  } else {
    return *(const float4*)(base + idx);
  }
}
// (a stored activation saturates at the largest f16 instead of overflowing to infinity: an inf would turn into NaN in the next
// layer's 0 * inf or inf - inf; 4 v_med3 per 8-byte store)
__device__ __forceinline__ float ocr_sat_h(float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); }
template <bool H16>
__device__ __forceinline__ void st4(float* base, long idx, const float4& v) {
  if constexpr (H16) *(uint2*)((_Float16*)base + idx) = __builtin_bit_cast(uint2, ocr_to_h4(make_float4(ocr_sat_h(v.x), ocr_sat_h(v.y), ocr_sat_h(v.z), ocr_sat_h(v.w))));
  else *(float4*)(base + idx) = v;
}
template <bool H16>
__device__ __forceinline__ float ld1(const float* base, long idx) {
  if constexpr (H16) return (float)((const _Float16*)base)[idx];
  else return base[idx];
}
template <bool H16>
__device__ __forceinline__ void st1(float* base, long idx, float v) {
  if constexpr (H16) ((_Float16*)base)[idx] = (_Float16)ocr_sat_h(v);
  else base[idx] = v;
}
// the same in two steps, for software-pipelined loops: the raw 4-channel piece as it travels (a conversion next to the load
// would wait for the data where the load was meant to stay in flight), converted where it is consumed
template <bool H16>
__device__ __forceinline__ typename WFrag<H16>::T ld4_raw(const float* base, long idx) {
  if constexpr (H16) return *(const uint2*)((const _Float16*)base + idx);
  else return *(const float4*)(base + idx);
}
__device__ __forceinline__ float4 up4(const float4& v) { return v; }
__device__ __forceinline__ float4 up4(const uint2& v) {
  const ocr_f4v f = __builtin_convertvector(__builtin_bit_cast(ocr_h4, v), ocr_f4v);
  return make_float4(f.x, f.y, f.z, f.w);
}
// the matrix-core operand of a lane straight from an f16 tensor (no conversion), or rounded from an f32 one
template <bool H16>
__device__ __forceinline__ ocr_h4 ld_h4(const float* base, long idx) {
  if constexpr (H16) return __builtin_bit_cast(ocr_h4, *(const uint2*)((const _Float16*)base + idx));
  else return ocr_to_h4(*(const float4*)(base + idx));
}

// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2).  Neighbouring
// tiles of these kernels re-read each other's input rows (3x3/5x5 taps, N-groups of one M-tile), so
// give each XCD a CONTIGUOUS range of logical tiles: re-reads then hit that XCD's L2 instead of
// HBM (rocprof r1e: 7x over-fetch on the det 3x3 conv, 5x on dw5x5 before this).  Bijective for any
// block count; affects speed only.
// development probe (tools/micro/conv_probe.hip defines OCR_CONV_PROBE): per-wave phase timestamps
#ifdef OCR_CONV_PROBE
__device__ long long* g_conv_probe;
#define CONV_PROBE(slot)                                                                               \
  if (g_conv_probe && (threadIdx.x & 63) == 0)                                                         \
    g_conv_probe[((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (slot)] = (long long)clock64()
#else
#define CONV_PROBE(slot)
#endif

__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return x * q + (x < r ? x : r) + i;
}

__device__ __forceinline__ void decompose(long m, int hw, int w, int& n, int& y, int& x) {
  n = (int)(m / hw);
  int r = (int)(m - (long)n * hw);
  y = r / w;
  x = r - y * w;
}

// ---- ragged batches (kernels_net.h, RagLevel) ----
// Line of flat index t when line n starts at mul * cw[n]: the largest n in [0, N) with mul * cw[n] <= t.
__device__ __forceinline__ int rag_line(const int* __restrict__ cw, int N, long t, long mul) {
  int lo = 0, hi = N;  // mul * cw[lo] <= t < mul * cw[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((long)cw[mid] * mul <= t) lo = mid;
    else hi = mid;
  }
  return lo;
}
// The same for a thread of a workgroup whose FIRST flat index is t0 (wave-uniform: the search runs on scalar values),
// walking forward from that line: a workgroup's indices span a line or two.
__device__ __forceinline__ int rag_line_near(const int* __restrict__ cw, int N, long t, long mul, long t0) {
  int n = rag_line(cw, N, t0, mul);
  while (n + 1 < N && (long)cw[n + 1] * mul <= t) ++n;
  return n;
}
// ---- geometry of sample n of a ragged tensor (lines: one height Hu for all; images: own height, level = input >> shift)
__device__ __forceinline__ int rag_w(const RagLevel& r, int n) { return r.h ? r.w[n] >> r.shift : r.w[n]; }
__device__ __forceinline__ int rag_h(const RagLevel& r, int n, int Hu) { return r.h ? r.h[n] >> r.shift : Hu; }
__device__ __forceinline__ long rag_pix0(const RagLevel& r, int n, int Hu) { return r.h ? (long)(r.cw[n] >> (2 * r.shift)) : (long)r.cw[n] * Hu; }
__device__ __forceinline__ long rag_row0(const RagLevel& r, int n, int Hu) { return r.h ? (long)(r.ch[n] >> r.shift) : (long)n * Hu; }
// sample of pixel m (m0 = the workgroup's first pixel, wave-uniform)
__device__ __forceinline__ int rag_sample_of_pixel(const RagLevel& r, int N, int Hu, long m, long m0) {
  if (r.h) return rag_line_near(r.cw, N, m << (2 * r.shift), 1, m0 << (2 * r.shift));  // (cw[n] are multiples of 4^shift)
  return rag_line_near(r.cw, N, m, Hu, m0);
}
// sample of row `row` of the [rows][Cs] view (SE pools)
__device__ __forceinline__ int rag_sample_of_row(const RagLevel& r, int N, int Hu, long row, long row0) {
  if (r.h) return rag_line_near(r.ch, N, row << r.shift, 1, row0 << r.shift);
  return (int)(row / Hu);
}
// pixel m of a ragged tensor -> (sample, y, x) and the sample's width / height; m0 = the workgroup's first pixel (uniform)
__device__ __forceinline__ void rag_decompose(const RagLevel& r, int N, int Hu, long m, long m0, int& n, int& y, int& x, int& w) {
  n = rag_sample_of_pixel(r, N, Hu, m, m0);
  w = rag_w(r, n);
  const int rem = (int)(m - rag_pix0(r, n, Hu));
  y = rem / w;
  x = rem - y * w;
}

// ---- scalar epilogue for one value at (n, y, x, physical channel pc); oidx = its NHWC offset ----
__device__ __forceinline__ float apply_epilogue(const Epilogue& ep, float v, int pc, int n, int y, int x, long oidx,
                                                int cs) {
  for (int s = 0; s < ep.n; ++s) {
    const EpStage& st = ep.st[s];
    switch (st.kind) {
      case EP_BIAS: v = v + st.v0[pc]; break;
      case EP_SMUL: v = st.p0 * v; break;
      case EP_SADD: v = v + st.p0; break;
      case EP_SFMA: v = fmaf(v, st.p0, st.p1); break;
      case EP_BN: { float t = v * st.v0[pc]; v = t + st.v1[pc]; } break;
      case EP_ACT: v = ocr_act(st.act, st.p0, st.p1, v); break;
      case EP_MULC: v = v * st.v0[(long)n * cs + pc]; break;
      case EP_ADDT: v = v + st.v0[oidx]; break;
      case EP_ADDUP: {
        int sy = y / st.a0, sx = x / st.a0;
        v = v + st.v0[(((long)n * st.a2 + sy) * st.a1 + sx) * cs + pc];
      } break;
    }
  }
  return v;
}

// 4 consecutive physical channels
// up_base >= 0 (ragged batch of images): first pixel and width of sample n in the upsampled operand's (coarser) tensor
template <bool H16 = false>
__device__ __forceinline__ float4 apply_epilogue4(const Epilogue& ep, float4 v, int pc, int n, int y, int x, long oidx,
                                                  int cs, long up_base = -1, int up_w = 0) {
  for (int s = 0; s < ep.n; ++s) {
    const EpStage& st = ep.st[s];
    switch (st.kind) {
      case EP_BIAS: { float4 b = *(const float4*)(st.v0 + pc); v.x = v.x + b.x; v.y = v.y + b.y; v.z = v.z + b.z; v.w = v.w + b.w; } break;
      case EP_SMUL: v.x = st.p0 * v.x; v.y = st.p0 * v.y; v.z = st.p0 * v.z; v.w = st.p0 * v.w; break;
      case EP_SADD: v.x = v.x + st.p0; v.y = v.y + st.p0; v.z = v.z + st.p0; v.w = v.w + st.p0; break;
      case EP_SFMA: v.x = fmaf(v.x, st.p0, st.p1); v.y = fmaf(v.y, st.p0, st.p1); v.z = fmaf(v.z, st.p0, st.p1); v.w = fmaf(v.w, st.p0, st.p1); break;
      case EP_BN: {
        float4 sc = *(const float4*)(st.v0 + pc), sh = *(const float4*)(st.v1 + pc);
        float t;
        t = v.x * sc.x; v.x = t + sh.x;
        t = v.y * sc.y; v.y = t + sh.y;
        t = v.z * sc.z; v.z = t + sh.z;
        t = v.w * sc.w; v.w = t + sh.w;
      } break;
      case EP_ACT:
        if (st.act == ACT_HSWISH) { ocr_hswish4(v.x, v.y, v.z, v.w); break; }
        v.x = ocr_act(st.act, st.p0, st.p1, v.x); v.y = ocr_act(st.act, st.p0, st.p1, v.y);
        v.z = ocr_act(st.act, st.p0, st.p1, v.z); v.w = ocr_act(st.act, st.p0, st.p1, v.w);
        break;
      case EP_MULC: { float4 g = *(const float4*)(st.v0 + (long)n * cs + pc); v.x = v.x * g.x; v.y = v.y * g.y; v.z = v.z * g.z; v.w = v.w * g.w; } break;
      case EP_GATERES: {
        float4 g = *(const float4*)(st.v0 + (long)n * cs + pc);
        float t;
        t = v.x * g.x; v.x = t + v.x;
        t = v.y * g.y; v.y = t + v.y;
        t = v.z * g.z; v.z = t + v.z;
        t = v.w * g.w; v.w = t + v.w;
      } break;
      case EP_ADDT: { float4 g = ld4<H16>(st.v0, oidx); v.x = v.x + g.x; v.y = v.y + g.y; v.z = v.z + g.z; v.w = v.w + g.w; } break;
      case EP_ADDUP: {
        int sy = y / st.a0, sx = x / st.a0;
        const long spix = up_base >= 0 ? up_base + (long)sy * up_w + sx : ((long)n * st.a2 + sy) * st.a1 + sx;
        float4 g = ld4<H16>(st.v0, spix * cs + pc);
        v.x = v.x + g.x; v.y = v.y + g.y; v.z = v.z + g.z; v.w = v.w + g.w;
      } break;
    }
  }
  return v;
}

// Epilogue of the NT 32x32 accumulator tiles of a wave.  The GEMM is issued as D = W * X^T (weights
// are the MFMA A operand, pixels the B operand), so a lane owns ONE pixel (column = lane & 31) and, per
// tile, 16 output channels: registers 4g..4g+3 are the four consecutive PHYSICAL channels
// tile*32 + 8g + 4*(lane>>5).  => per-channel parameters arrive as float4, results leave as 16-byte
// stores, per-pixel stages (SE gate, residual, upsample-add) need the pixel's (n, y, x) once per lane.
//
// Shape of the code (probe, tools/micro/conv_probe.hip: at K = 240 the old tile-by-tile epilogue lasted
// as long as the K loop, almost all of it waiting): the STAGE loop is outermost and every stage sweeps
// all tiles, so a stage's parameter loads for the whole wave are issued back to back and waited for
// once, the stage descriptor is decoded once, and no load is ever issued behind a store (loads and
// stores retire through one in-order counter).  PLAIN (logical channel order, unaligned rows: the two
// heads) is a template parameter so the packed path keeps its 16-byte stores.
// (n, y, x) of output pixel m for the per-image epilogue stages: a uniform grid, or the ragged batch's tables (the wave's
// first pixel - uniform - starts the search)
__device__ __forceinline__ void conv_finish_nyx(const ConvArgs& a, long m, int& n, int& y, int& x) {
  if (a.rout.w) {
    int w;
    rag_decompose(a.rout, a.N, a.OH, m, m - (threadIdx.x & 31), n, y, x, w);
  } else {
    decompose(m, a.OH * a.OW, a.OW, n, y, x);
  }
}

// xp (OUT_C8I only): a per-wave LDS tile of OCR_XP_FLOATS floats.  A lane owns a PIXEL, so its stores are 16-byte pieces of 32
// different lines per instruction (a pixel's row is Cs_out * 4 bytes); with the tile the wave's 32 x 32 block of column tile t
// goes through LDS and leaves as whole 128-byte lines - lane = (pixel 8 i + lane / 8, quad lane % 8), eight pixels per
// instruction.  The wave's 32 pixels must be consecutive (m = first + lane % 32) and every lane must arrive here.  Thin layers
// are bound by exactly this (tools/micro/conv_time.hip, round 5: 12 -> 96 at 3.7 M pixels 0.454 -> 0.355 ms with the line
// pattern, 16 -> 32 0.426 -> 0.297, 48 -> 96 0.681 -> 0.565; 480 -> 480 unchanged).
#define OCR_XP_STRIDE 36
#define OCR_XP_FLOATS (32 * OCR_XP_STRIDE)
template <int NT, int MODE, bool H16 = false>
__device__ __forceinline__ void conv_finish(const ConvArgs& a, const Epilogue& ep, floatx16 (&acc)[NT], int nt0, long m, int hb,
                                            const float* spar, const int sstride = NT * 32, float* xp = nullptr) {
  const long m_lane = m;
  if (m >= a.M) {
    if (MODE != OUT_C8I || !xp) return;
    m = a.M - 1;  // (past the end: the last pixel again - the lane takes part in the transposed stores below)
  }
  // the pixel's (n, y, x) is only needed by the deconv scatter and by the per-image / upsampled
  // stages: decoded there (two integer divisions), not in front of every K loop
  int n = 0, y = 0, x = 0;
  if constexpr (MODE == OUT_DECONV) decompose(m, a.OH * a.OW, a.OW, n, y, x);
  // Row of quad (t, g): R = r0 + rel, rel = 32t + 8g a compile-time constant, r0 = nt0*32 + 4hb.
  // Everything addressed per quad is "one per-lane base + a uniform offset": nothing per-quad is kept
  // in registers across the stage loop (16 channel indices + 16 64-bit offsets hoisted out of it cost
  // 50 VGPRs and spilled).  Packed modes: ColsStore and CoutPadded are multiples of 8, so whether a
  // quad exists and which deconv quadrant it falls in is uniform; PLAIN (ColsStore = Cout) tests rows.
  const int c0 = nt0 * 32, r0 = c0 + 4 * hb;
  auto exists = [&](int rel) { return c0 + rel < a.ColsStore; };  // uniform
  // offset of the quad's first float relative to the lane's pixel in an output-shaped tensor
  auto ooff = [&](int rel) -> long {
    if constexpr (MODE == OUT_DECONV) {
      const int dq = (c0 + rel) / a.CoutPadded;  // uniform
      return ((long)(dq >> 1) * (2 * a.OW) + (dq & 1)) * a.Cs_out + (rel - dq * a.CoutPadded);
    }
    return rel;
  };
  // channel offset of the quad relative to r0 in a per-channel vector
  auto coff = [&](int rel) {
    if constexpr (MODE == OUT_DECONV) return rel - ((c0 + rel) / a.CoutPadded) * a.CoutPadded;
    return rel;
  };
  const long opix = MODE == OUT_DECONV ? (((long)n * (2 * a.OH) + 2 * y) * (2 * a.OW) + 2 * x) * a.Cs_out + r0
                                       : m * a.Cs_out + r0;
#define OCR_EP_SWEEP(BODY)                                   \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) {           \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) {          \
      constexpr_rel(t, g);                                   \
      if (!exists(rel)) continue;                            \
      float wx = acc[t][4 * g], wy = acc[t][4 * g + 1];      \
      float wz = acc[t][4 * g + 2], ww = acc[t][4 * g + 3];  \
      BODY                                                   \
      acc[t][4 * g] = wx; acc[t][4 * g + 1] = wy;            \
      acc[t][4 * g + 2] = wz; acc[t][4 * g + 3] = ww;        \
    }                                                        \
  }
#define constexpr_rel(t, g) const int rel = 32 * (t) + 8 * (g)
  int vs = 0;  // slot of the next per-channel vector in spar (conv_stage_params packs them in stage order)
  for (int s = 0; s < ep.n; ++s) {
    const EpStage& st = ep.st[s];
    switch (st.kind) {
      case EP_BIAS: {
        const float* q0 = spar + vs * sstride + 4 * hb;
        vs += 1;
        OCR_EP_SWEEP({ const float4 b = *(const float4*)(q0 + rel); wx = wx + b.x; wy = wy + b.y; wz = wz + b.z; ww = ww + b.w; })
      } break;
      case EP_SMUL: {
        const float k = st.p0;
        OCR_EP_SWEEP({ wx = k * wx; wy = k * wy; wz = k * wz; ww = k * ww; })
      } break;
      case EP_SADD: {
        const float k = st.p0;
        OCR_EP_SWEEP({ wx = wx + k; wy = wy + k; wz = wz + k; ww = ww + k; })
      } break;
      case EP_SFMA: {  // one rounding per value (v_pk_fma_f32 with the scalar pair)
        const ocr_f2 k = {st.p0, st.p0};
        const ocr_f2 d = {st.p1, st.p1};
        OCR_EP_SWEEP({
          ocr_f2 lo; ocr_f2 hi;
          lo.x = wx; lo.y = wy; hi.x = wz; hi.y = ww;
          lo = __builtin_elementwise_fma(lo, k, d); hi = __builtin_elementwise_fma(hi, k, d);
          wx = lo.x; wy = lo.y; wz = hi.x; ww = hi.y;
        })
      } break;
      case EP_BN: {
        const float* q0 = spar + vs * sstride + 4 * hb;
        vs += 2;
        OCR_EP_SWEEP({
          const float4 sc = *(const float4*)(q0 + rel);
          const float4 sh = *(const float4*)(q0 + sstride + rel);
          float u;
          u = wx * sc.x; wx = u + sh.x;
          u = wy * sc.y; wy = u + sh.y;
          u = wz * sc.z; wz = u + sh.z;
          u = ww * sc.w; ww = u + sh.w;
        })
      } break;
      case EP_ACT: {
        // the activation kind is decoded once per stage, not once per value
        const float p0 = st.p0, p1 = st.p1;
#define OCR_ACT_SWEEP(KIND) OCR_EP_SWEEP({ wx = ocr_act(KIND, p0, p1, wx); wy = ocr_act(KIND, p0, p1, wy); wz = ocr_act(KIND, p0, p1, wz); ww = ocr_act(KIND, p0, p1, ww); })
        switch (st.act) {
          case ACT_RELU: OCR_ACT_SWEEP(ACT_RELU) break;
          case ACT_HSWISH: {
            // range pass, then the division-free sweep (ocr_common.h); anything out of range takes the division
            float mn = INFINITY, mx = 0.0f;
            OCR_EP_SWEEP({ ocr_absrange(mn, mx, wx, wy); ocr_absrange(mn, mx, wz, ww); })
            if (ocr_hsw_fast_ok(mn, mx)) { OCR_EP_SWEEP({ ocr_hswish4_fast(wx, wy, wz, ww); }) }
            else { OCR_ACT_SWEEP(ACT_HSWISH) }
          } break;
          case ACT_HSW6:
            OCR_EP_SWEEP({
              ocr_f2 lo; ocr_f2 hi;
              lo.x = wx; lo.y = wy; hi.x = wz; hi.y = ww;
              lo = ocr_hsw6_2(lo); hi = ocr_hsw6_2(hi);
              wx = lo.x; wy = lo.y; wz = hi.x; ww = hi.y;
            })
            break;
          case ACT_HSIG: OCR_ACT_SWEEP(ACT_HSIG) break;
          case ACT_SWISH: OCR_ACT_SWEEP(ACT_SWISH) break;
          default: OCR_ACT_SWEEP(ACT_SIGMOID) break;
        }
#undef OCR_ACT_SWEEP
      } break;
      case EP_MULC: {  // per-image channel gate [N][Cs_out]
        if constexpr (MODE != OUT_DECONV) conv_finish_nyx(a, m, n, y, x);
        const float* gate = st.v0 + (long)n * a.Cs_out + r0;
        OCR_EP_SWEEP({ const float4 r = *(const float4*)(gate + coff(rel)); wx = wx * r.x; wy = wy * r.y; wz = wz * r.z; ww = ww * r.w; })
      } break;
      case EP_GATERES: {  // x * gate + x: two roundings, as the ew pass (mulc, then addt of x itself) it replaces
        if constexpr (MODE != OUT_DECONV) conv_finish_nyx(a, m, n, y, x);
        const float* gate = st.v0 + (long)n * a.Cs_out + r0;
        OCR_EP_SWEEP({
          const float4 r = *(const float4*)(gate + coff(rel));
          float u;
          u = wx * r.x; wx = u + wx;
          u = wy * r.y; wy = u + wy;
          u = wz * r.z; wz = u + wz;
          u = ww * r.w; ww = u + ww;
        })
      } break;
      case EP_ADDT: {  // tensor of the output's shape
        OCR_EP_SWEEP({ const float4 r = ld4<H16>(st.v0, opix + ooff(rel)); wx = wx + r.x; wy = wy + r.y; wz = wz + r.z; ww = ww + r.w; })
      } break;
      case EP_ADDUP: {  // nearest-upsampled coarser map (never after a deconv)
        if constexpr (MODE != OUT_DECONV) conv_finish_nyx(a, m, n, y, x);
        long spix = ((long)n * st.a2 + y / st.a0) * st.a1 + x / st.a0;
        if (a.rout.h) {  // ragged batch of images: the operand lives on the level log2(a0) coarser, same tables
          const int lu = 31 - __clz(st.a0);
          spix = (long)(a.rout.cw[n] >> (2 * (a.rout.shift + lu))) + (long)(y / st.a0) * (a.rout.w[n] >> (a.rout.shift + lu)) + x / st.a0;
        }
        const long upo = spix * a.Cs_out + r0;
        OCR_EP_SWEEP({ const float4 r = ld4<H16>(st.v0, upo + coff(rel)); wx = wx + r.x; wy = wy + r.y; wz = wz + r.z; ww = ww + r.w; })
      } break;
    }
  }
#undef OCR_EP_SWEEP
  if constexpr (MODE == OUT_HEAD) {
    // Fused softmax head, first half (canonical order of the row softmax, shared with the oracle and with
    // softmax_argmax_kernel): columns in groups of 128; inside a group two interleaved chains - the
    // columns with ((c >> 2) & 1) == 0 and == 1, i.e. the two half-waves of this layout - each summed in
    // ascending column order, then chain 0 + chain 1.  The logits never leave the registers.
    float mx = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = r0 + 32 * t + 8 * g + k;
          const float v = acc[t][4 * g + k];
          if (c < a.Cout && v > mx) { mx = v; mi = c; }  // ascending c: the first maximum wins
        }
    const float omx = __shfl_xor(mx, 32);
    const int omi = __shfl_xor(mi, 32);
    if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = r0 + 32 * t + 8 * g + k;
          if (c < a.Cout) part = part + ocr_expf(acc[t][4 * g + k] - mx);
        }
    const float other = __shfl_xor(part, 32);
    if (hb == 0) {
      const long q = m * (a.NTtot / NT) + nt0 / NT;
      a.head_max[q] = mx;
      a.head_sum[q] = part + other;
      a.head_idx[q] = mi;
    }
    return;
  }
  // ---- stores, after the last load
  if constexpr (MODE == OUT_C8I) {
    if (xp) {
      const int lane = threadIdx.x & 63;
      const long m_first = m_lane - (lane & 31);
      float* const mine = xp + (lane & 31) * OCR_XP_STRIDE + 4 * hb;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (!exists(32 * t)) break;  // uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          constexpr_rel(t, g);
          if (!exists(rel)) continue;
          float4 w = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
          if (a.Cout != a.Cs_out) {  // keep the pad channels of the octet layout at zero
            const int pc = r0 + rel;
            if (c8i_logical(pc) >= a.Cout) w.x = 0.f;
            if (c8i_logical(pc + 1) >= a.Cout) w.y = 0.f;
            if (c8i_logical(pc + 2) >= a.Cout) w.z = 0.f;
            if (c8i_logical(pc + 3) >= a.Cout) w.w = 0.f;
          }
          *(float4*)(mine + 8 * g) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int q = lane & 7, col = c0 + 32 * t + 4 * q;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int px = 8 * i + (lane >> 3);
          const long mm = m_first + px;
          if (mm < a.M && col < a.ColsStore) st4<H16>(a.out, mm * a.Cs_out + col, *(const float4*)(xp + px * OCR_XP_STRIDE + 4 * q));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      return;
    }
  }
  float* obase = a.out + (H16 && MODE != OUT_PLAIN ? 0 : opix);  // (f16 tensors are addressed by element index below)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      constexpr_rel(t, g);
      if (!exists(rel)) continue;
      float4 w = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
      float* dst = obase + ooff(rel);
      if constexpr (MODE == OUT_PLAIN) {
        // logical channel order, row stride = Cout (rows are not 16-byte aligned): scalar stores;
        // per-channel parameter vectors of plain outputs are padded to whole tiles
        const int R = r0 + rel;
        if (R < a.Cout) dst[0] = w.x;
        if (R + 1 < a.Cout) dst[1] = w.y;
        if (R + 2 < a.Cout) dst[2] = w.z;
        if (R + 3 < a.Cout) dst[3] = w.w;
      } else {
        if (a.Cout != a.Cs_out) {  // keep the pad channels of the octet layout at zero
          const int pc = r0 + coff(rel);
          if (c8i_logical(pc) >= a.Cout) w.x = 0.f;
          if (c8i_logical(pc + 1) >= a.Cout) w.y = 0.f;
          if (c8i_logical(pc + 2) >= a.Cout) w.z = 0.f;
          if (c8i_logical(pc + 3) >= a.Cout) w.w = 0.f;
        }
#ifdef OCR_PROBE_NOSTORE
        if (w.x == 12345.678f) *(float4*)dst = w;  // development probe: keep the value live, skip the traffic
#else
        if constexpr (H16) st4<true>(a.out, opix + ooff(rel), w);
        else *(float4*)dst = w;
#endif
      }
    }
  }
#undef constexpr_rel
}

// Per-channel epilogue vectors (bias, BN scale/shift) of this workgroup's NT*32 GEMM rows, copied to
// LDS at kernel start: spar[slot*NT*32 + row - nt0*32], slots handed out in stage order (1 per bias, 2 per BN).  An epilogue load from global memory
// queues behind every streaming load of the CU's 12-16 waves (probe: ~10k clocks each time);
// the LDS copy is fetched while the K loop runs and read back in ~100 clocks.
template <int NT>
__device__ __forceinline__ void conv_stage_params(const ConvArgs& a, const Epilogue& ep, int nt0, float* spar) {
  const int i = threadIdx.x;
  if (i < NT * 32) {
    const int R = nt0 * 32 + i;
    const int pc = R >= a.ColsStore ? 0 : (a.out_mode == OUT_DECONV ? R % a.CoutPadded : R);
    float v[2 * OCR_MAX_EP];  // every stage's loads are issued before the first is waited for
#pragma unroll
    for (int s = 0; s < OCR_MAX_EP; ++s) {
      const int kind = s < ep.n ? ep.st[s].kind : -1;
      v[2 * s] = (kind == EP_BIAS || kind == EP_BN) ? ep.st[s].v0[pc] : 0.f;
      v[2 * s + 1] = kind == EP_BN ? ep.st[s].v1[pc] : 0.f;
    }
    int vs = 0;
#pragma unroll
    for (int s = 0; s < OCR_MAX_EP; ++s) {
      const int kind = s < ep.n ? ep.st[s].kind : -1;
      if (kind == EP_BIAS || kind == EP_BN) { spar[vs * NT * 32 + i] = v[2 * s]; ++vs; }
      if (kind == EP_BN) { spar[vs * NT * 32 + i] = v[2 * s + 1]; ++vs; }
    }
  }
  __syncthreads();
}

}  // namespace ocr
