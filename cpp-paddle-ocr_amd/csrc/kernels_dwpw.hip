// Fused depthwise -> pointwise block (the PPLCNetV3 pair `depthwise_conv2d` + epilogue -> `conv2d 1x1` + epilogue
// of det ops 1..20 / rec ops 1..20) for gfx950.
//
// Unfused, the depthwise kernel writes a tensor that the 1x1 conv reads straight back: 2 x 1.7 GB per pair in the
// rec 240-channel block, 2 x 0.9..1.9 GB in the thin high-resolution layers (r1 profile: a third of the kernel time
// in depthwise kernels at 0.4 of the HBM peak).  Here the depthwise result never leaves the CU:
//
//   workgroup   : a TH x 16 tile of OUTPUT pixels of one image (32*WP pixels) x WC*NT*32 output channels
//   channel loop: chunks of CK input channels, software-pipelined, ONE barrier per chunk:
//       G  global -> registers   the haloed input region of chunk i+2 (16-byte pieces, zero outside the image),
//                                the chunk's K*K depthwise weights and its per-channel epilogue vectors
//       DW LDS -> VALU -> LDS    depthwise conv of chunk i+1 + its whole epilogue, written as the MFMA pixel operand:
//                                a thread owns PR vertically adjacent output pixels x 4 physical channels, taps in
//                                (ky, kx) ascending order from 0 - the contract's chain
//       MMA LDS -> MFMA          chunk i: v_mfma_f32_32x32x2_f32 against the 1x1 weights' fragment image (the same
//                                image conv_mfma_kernel reads; a wave = 32 pixels x NT column tiles), ascending k
//       S  registers -> LDS      chunk i+2's region into the buffer chunk i+1 has just left
//   epilogue    : conv_finish (the 1x1 conv's own epilogue and 16-byte stores), unchanged
//
// The pixel operand lives in LDS exactly as the C8I tensor would in HBM (pixel-major, 4 consecutive physical channels
// per lane), so the matrix pipe sees the same k-ordered chain as the unfused pair: results are bit-identical
// (tests/test_gpu_parity.py runs every plan with the fusion on and compares each materialised tensor).
// LDS rows are padded by 16 bytes: pixel stride = CK/4 + 1 sixteen-byte units (odd), so the 16-lane groups of
// ds_read_b128 / ds_write_b128 that walk consecutive pixels hit 16 distinct 4-bank slots.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "conv_device.h"
#include "kernels_net.h"

namespace ocr {

namespace {

struct F4 { ocr_f2 lo, hi; };

template <int K, int SH, int SW, int CK, bool WIDE>
struct DwPwGeom {
  static constexpr int WP = WIDE ? 2 : 4, WC = WIDE ? 2 : 1;
  static constexpr int TW = 16, TH = 2 * WP, P = TH * TW;
  static constexpr int PR = (WIDE && CK == 16) ? 1 : 2;  // output rows per depthwise item: keeps all 256 threads busy
  static constexpr int IH = (TH - 1) * SH + K, IW = (TW - 1) * SW + K;
  static constexpr int S = CK + 4, Q = CK / 4, C8S = CK / 8;
  static constexpr int IN_TILE = IH * IW * S, OP_TILE = P * S, WT = K * K * CK;
  static constexpr int G_PIECES = IH * IW * Q, G_PER = (G_PIECES + 255) / 256;
  static constexpr int NIT = (TH / PR) * TW * Q, IT_PER = (NIT + 255) / 256;
  static constexpr int VEC_SLOTS = 2;  // per-channel vectors of the 1x1 conv's epilogue (host checks)
  static constexpr size_t lds_floats(int nt) {
    return 2 * (size_t)IN_TILE + 2 * (size_t)OP_TILE + 2 * (size_t)WT + 4 * (size_t)CK + (size_t)WC * VEC_SLOTS * nt * 32;
  }
};

// per-channel epilogue vectors of the 1x1 conv for the WC column groups of this workgroup (conv_stage_params for
// more than one group): region wc holds slot*NT*32 + row, as conv_finish<NT> reads it
template <int NT, int WC, int SLOTS>
__device__ __forceinline__ void dwpw_stage_params(const ConvArgs& a, const Epilogue& ep, int nt0_wg, float* spar) {
  const int i = threadIdx.x;
  if (i < WC * NT * 32) {
    const int g = i / (NT * 32), ii = i - g * (NT * 32);
    const int R = (nt0_wg + g * NT) * 32 + ii;
    const int pc = R >= a.ColsStore ? 0 : R;
    float v[2 * OCR_MAX_EP];
#pragma unroll
    for (int s = 0; s < OCR_MAX_EP; ++s) {
      const int kind = s < ep.n ? ep.st[s].kind : -1;
      v[2 * s] = (kind == EP_BIAS || kind == EP_BN) ? ep.st[s].v0[pc] : 0.f;
      v[2 * s + 1] = kind == EP_BN ? ep.st[s].v1[pc] : 0.f;
    }
    float* dst = spar + g * (SLOTS * NT * 32) + ii;
    int vs = 0;
#pragma unroll
    for (int s = 0; s < OCR_MAX_EP; ++s) {
      const int kind = s < ep.n ? ep.st[s].kind : -1;
      if ((kind == EP_BIAS || kind == EP_BN) && vs < SLOTS) { dst[vs * NT * 32] = v[2 * s]; ++vs; }
      if (kind == EP_BN && vs < SLOTS) { dst[vs * NT * 32] = v[2 * s + 1]; ++vs; }
    }
  }
}

}  // namespace

template <int K, int SH, int SW, int CK, bool WIDE, int NT>
__global__ void __launch_bounds__(256, 2) dwpw_kernel(const DwPwArgs a, const Epilogue epd, const Epilogue epp) {
  using G_ = DwPwGeom<K, SH, SW, CK, WIDE>;
  constexpr int WC = G_::WC, TW = G_::TW, TH = G_::TH, PR = G_::PR, IW = G_::IW;
  constexpr int S = G_::S, Q = G_::Q, C8S = G_::C8S, IN_TILE = G_::IN_TILE, OP_TILE = G_::OP_TILE, WT = G_::WT;
  constexpr int G_PIECES = G_::G_PIECES, G_PER = G_::G_PER, NIT = G_::NIT, IT_PER = G_::IT_PER, SLOTS = G_::VEC_SLOTS;
  constexpr int WQ = K * K * Q;  // 16-byte pieces of the chunk's depthwise weights
  static_assert(WQ + 2 * Q <= 256, "one weight / vector piece per thread");
  static_assert(C8S % 2 == 0, "the fragment stream is walked two octets at a time");
  extern __shared__ float4 s_dwpw4[];
  float* s_in = (float*)s_dwpw4;      // [2][IH*IW][S]   haloed input region of a chunk
  float* s_op = s_in + 2 * IN_TILE;   // [2][P][S]       depthwise result = MFMA pixel operand
  float* s_w = s_op + 2 * OP_TILE;    // [2][K*K][CK]    depthwise weights of a chunk
  float* s_b = s_w + 2 * WT;          // [2][2][CK]      its per-channel vectors (bias | BN scale, BN shift)
  float* s_par = s_b + 4 * CK;        // [WC][SLOTS][NT*32]  the 1x1 conv's per-channel vectors

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  const ConvArgs& c = a.c;
  const int Cs = c.Cs_in;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);  // column block fastest, then tiles in raster order
  const unsigned cblocks = (unsigned)c.NTtot / (NT * WC);
  const int cb = (int)(lb % cblocks);
  unsigned tile = lb / cblocks;
  const int tx = (int)(tile % (unsigned)a.tiles_x);
  tile /= (unsigned)a.tiles_x;
  const int ty = (int)(tile % (unsigned)a.tiles_y), n = (int)(tile / (unsigned)a.tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  const int wp = WIDE ? (wave & 1) : wave, wc = WIDE ? (wave >> 1) : 0;
  const int nt0 = (cb * WC + wc) * NT;
  const int nch = Cs / CK;

  // ---- fill plan: this thread's pieces of the input region (float offset inside the image, -1 = zero)
  const float* img = a.dw_in + (long)n * a.H * a.W * Cs;
  const int iy0 = y0 * SH - a.PH, ix0 = x0 * SW - a.PW;
  int goff[G_PER];
#pragma unroll
  for (int i = 0; i < G_PER; ++i) {
    const int idx = tid + i * 256;
    const int px = idx / Q, q = idx - px * Q;
    const int py = px / IW, pxx = px - py * IW;
    const int iy = iy0 + py, ix = ix0 + pxx;
    const bool ok = idx < G_PIECES && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    goff[i] = ok ? (iy * a.W + ix) * Cs + 4 * q : -1;
  }
  float4 greg[G_PER];
  float4 wreg = make_float4(0.f, 0.f, 0.f, 0.f);
  auto G = [&](int ch) {
    const float* base = img + ch * CK;
#pragma unroll
    for (int i = 0; i < G_PER; ++i)
      greg[i] = goff[i] >= 0 ? *(const float4*)(base + goff[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < WQ) {
      const int t = tid / Q, q = tid - t * Q;
      wreg = *(const float4*)(a.dw_w + (long)t * Cs + ch * CK + 4 * q);
    } else if (tid < WQ + 2 * Q) {
      const int j = tid - WQ;
      const float* src = j < Q ? a.dw_v0 : a.dw_v1;
      wreg = src ? *(const float4*)(src + ch * CK + 4 * (j < Q ? j : j - Q)) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto Sfill = [&](int buf) {
    float* si = s_in + buf * IN_TILE;
#pragma unroll
    for (int i = 0; i < G_PER; ++i) {
      const int idx = tid + i * 256;
      const int px = idx / Q, q = idx - px * Q;
      if (G_PIECES % 256 == 0 || idx < G_PIECES) *(float4*)(si + px * S + 4 * q) = greg[i];
    }
    if (tid < WQ) *(float4*)(s_w + buf * WT + 4 * tid) = wreg;            // [tap][CK]: tap*CK + 4q = 4*tid
    else if (tid < WQ + 2 * Q) *(float4*)(s_b + buf * 2 * CK + 4 * (tid - WQ)) = wreg;
  };

  // ---- depthwise conv + epilogue of one chunk: s_in[buf] -> s_op[buf]
  auto DW = [&](int buf) {
    const float* si = s_in + buf * IN_TILE;
    float* so = s_op + buf * OP_TILE;
    const float* sw = s_w + buf * WT;
    const float* sb = s_b + buf * 2 * CK;
#pragma unroll
    for (int u = 0; u < IT_PER; ++u) {
      const int it = tid + u * 256;
      if (NIT % 256 != 0 && it >= NIT) break;
      const int txx = it & (TW - 1), r_ = it / TW;
      const int rp = r_ % (TH / PR), q = r_ / (TH / PR);
      F4 acc[PR];
#pragma unroll
      for (int o = 0; o < PR; ++o) { acc[o].lo = ocr_f2{0.f, 0.f}; acc[o].hi = ocr_f2{0.f, 0.f}; }
      const float* col = si + ((rp * PR * SH) * IW + txx * SW) * S + 4 * q;
      const float* wq = sw + 4 * q;
#pragma unroll
      for (int r = 0; r < (PR - 1) * SH + K; ++r) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 v = *(const float4*)(col + (r * IW + kx) * S);
#pragma unroll
          for (int o = 0; o < PR; ++o) {
            const int ky = r - o * SH;  // compile-time after unrolling
            if (ky < 0 || ky >= K) continue;
            const float4 w = *(const float4*)(wq + (ky * K + kx) * CK);
            acc[o].lo = __builtin_elementwise_fma(ocr_f2{v.x, v.y}, ocr_f2{w.x, w.y}, acc[o].lo);
            acc[o].hi = __builtin_elementwise_fma(ocr_f2{v.z, v.w}, ocr_f2{w.z, w.w}, acc[o].hi);
          }
        }
      }
      // the depthwise conv's epilogue (bias / BN, LAB scalars, activation), stage loop outside
      for (int s = 0; s < epd.n; ++s) {
        const EpStage& st = epd.st[s];
#define OCR_DP_SWEEP(BODY) _Pragma("unroll") for (int o = 0; o < PR; ++o) { F4& v = acc[o]; BODY }
        switch (st.kind) {
          case EP_BIAS: {
            const float4 b = *(const float4*)(sb + 4 * q);
            const ocr_f2 blo = {b.x, b.y}, bhi = {b.z, b.w};
            OCR_DP_SWEEP({ v.lo = v.lo + blo; v.hi = v.hi + bhi; })
          } break;
          case EP_SMUL: {
            const ocr_f2 k = {st.p0, st.p0};
            OCR_DP_SWEEP({ v.lo = k * v.lo; v.hi = k * v.hi; })
          } break;
          case EP_SADD: {
            const ocr_f2 k = {st.p0, st.p0};
            OCR_DP_SWEEP({ v.lo = v.lo + k; v.hi = v.hi + k; })
          } break;
          case EP_BN: {
            const float4 sc = *(const float4*)(sb + 4 * q), sh = *(const float4*)(sb + CK + 4 * q);
            const ocr_f2 clo = {sc.x, sc.y}, chi = {sc.z, sc.w}, hlo = {sh.x, sh.y}, hhi = {sh.z, sh.w};
            OCR_DP_SWEEP({
              ocr_f2 t;
              t = v.lo * clo; v.lo = t + hlo;
              t = v.hi * chi; v.hi = t + hhi;
            })
          } break;
          case EP_ACT: {
            const float p0 = st.p0, p1 = st.p1;
#define OCR_DP_ACT(KIND) OCR_DP_SWEEP({ v.lo.x = ocr_act(KIND, p0, p1, v.lo.x); v.lo.y = ocr_act(KIND, p0, p1, v.lo.y); v.hi.x = ocr_act(KIND, p0, p1, v.hi.x); v.hi.y = ocr_act(KIND, p0, p1, v.hi.y); })
            switch (st.act) {
              case ACT_RELU: OCR_DP_ACT(ACT_RELU) break;
              case ACT_HSWISH: {
                float mn = INFINITY, mx = 0.0f;
                OCR_DP_SWEEP({ ocr_absrange(mn, mx, v.lo.x, v.lo.y); ocr_absrange(mn, mx, v.hi.x, v.hi.y); })
                if (ocr_hsw_fast_ok(mn, mx)) { OCR_DP_SWEEP({ v.lo = ocr_hswish2_fast(v.lo); v.hi = ocr_hswish2_fast(v.hi); }) }
                else { OCR_DP_ACT(ACT_HSWISH) }
              } break;
              case ACT_HSIG: OCR_DP_ACT(ACT_HSIG) break;
              case ACT_SWISH: OCR_DP_ACT(ACT_SWISH) break;
              default: OCR_DP_ACT(ACT_SIGMOID) break;
            }
#undef OCR_DP_ACT
          } break;
          default: break;  // per-pixel stages never follow a fused depthwise conv (host checks)
        }
#undef OCR_DP_SWEEP
      }
#pragma unroll
      for (int o = 0; o < PR; ++o)
        *(float4*)(so + ((rp * PR + o) * TW + txx) * S + 4 * q) = make_float4(acc[o].lo.x, acc[o].lo.y, acc[o].hi.x, acc[o].hi.y);
    }
  };

  // ---- 1x1 conv of one chunk on the matrix cores: s_op[buf] x fragment image
  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const float4* p_w = (const float4*)c.wfrag + (long)nt0 * 64 + lane;
  const long wstride = (long)c.NTtot * 64;
  int w_left = nch * C8S - 1;  // fragment steps after the current one
  float4 b0[NT], b1[NT];
  auto loadB = [&](float4 (&bv)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = p_w[t * 64];
    const bool more = w_left > 0;  // past the end: stay on the last step (loaded, unused)
    p_w += more ? wstride : 0;
    w_left -= more;
  };
  auto mfma4 = [&](const float4 (&bv)[NT], const float4& av) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].x, av.x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].y, av.y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].z, av.z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].w, av.w, acc[t], 0, 0, 0);
    }
  };
  auto MMA = [&](int buf) {
    const float* so = s_op + buf * OP_TILE + (wp * 32 + p) * S + 4 * h;
#pragma unroll
    for (int j = 0; j < C8S; j += 2) {
      loadB(b1);
      mfma4(b0, *(const float4*)(so + 8 * j));
      loadB(b0);
      mfma4(b1, *(const float4*)(so + 8 * (j + 1)));
    }
  };

  // ---- pipeline
  G(0);
  dwpw_stage_params<NT, WC, SLOTS>(c, epp, cb * WC * NT, s_par);
  Sfill(0);
  __syncthreads();
  if (nch > 1) G(1);
  loadB(b0);
  for (int i = 0; i < nch; ++i) {
    DW(i & 1);
    if (i >= 1) MMA((i - 1) & 1);
    if (i + 1 < nch) Sfill((i + 1) & 1);
    if (i + 2 < nch) G(i + 2);
    __syncthreads();
  }
  MMA((nch - 1) & 1);

  const int pix = wp * 32 + p;
  const int oy = y0 + pix / TW, ox = x0 + (pix & (TW - 1));
  const long m = (oy < c.OH && ox < c.OW) ? ((long)n * c.OH + oy) * c.OW + ox : c.M;
  conv_finish<NT, OUT_C8I>(c, epp, acc, nt0, m, h, s_par + wc * (SLOTS * NT * 32));
}

namespace {

template <int K, int SH, int SW, int CK, bool WIDE, int NT>
bool launch_one(const DwPwArgs& a0, const Epilogue& epd, const Epilogue& epp, hipStream_t s, bool query) {
  using G_ = DwPwGeom<K, SH, SW, CK, WIDE>;
  DwPwArgs a = a0;
  a.tiles_x = (a.c.OW + G_::TW - 1) / G_::TW;
  a.tiles_y = (a.c.OH + G_::TH - 1) / G_::TH;
  const unsigned cblocks = (unsigned)a.c.NTtot / (NT * G_::WC);
  const dim3 grid((unsigned)((long)a.c.N * a.tiles_y * a.tiles_x * cblocks));
  const size_t lds = G_::lds_floats(NT) * sizeof(float);
  if (lds > 64 * 1024) {
    static unsigned char attr_state[64] = {};
    if (!raise_dynamic_lds((const void*)dwpw_kernel<K, SH, SW, CK, WIDE, NT>, (int)lds, attr_state)) return false;
  }
  if (query) return true;  // asked at bind time, on the device that will run it: a refusal falls back to the pair
  hipLaunchKernelGGL((dwpw_kernel<K, SH, SW, CK, WIDE, NT>), grid, dim3(256), lds, s, a, epd, epp);
  return true;
}

}  // namespace

// The instantiated shapes (everything else stays an unfused pair): K, strides, chunk width, workgroup shape
// (thin: 4 waves = 4 pixel groups of an 8x16 tile, all <= 4 column tiles per wave; wide: 2 pixel groups of a 4x16
// tile x 2 column groups), column tiles per wave.  `query` only answers whether the pair is on this path.
bool launch_dwpw(const DwPwArgs& a, const Epilogue& epd, const Epilogue& epp, hipStream_t s, bool query) {
  const int K = a.K, SH = a.SH, SW = a.SW, Cs = a.c.Cs_in, tiles = a.c.NTtot;
#define OCR_DWPW_CASE(K_, SH_, SW_, CK_, WIDE_, NT_, COND)                                                \
  if (K == K_ && SH == SH_ && SW == SW_ && Cs % CK_ == 0 && tiles % (NT_ * (WIDE_ ? 2 : 1)) == 0 && (COND)) \
    return launch_one<K_, SH_, SW_, CK_, WIDE_, NT_>(a, epd, epp, s, query);
  // thin layers: tiles <= 4, one wave owns every output column of its 32 pixels
  OCR_DWPW_CASE(3, 1, 1, 16, false, 1, tiles == 1)
  OCR_DWPW_CASE(3, 1, 1, 16, false, 2, tiles == 2)
  OCR_DWPW_CASE(3, 1, 1, 16, false, 3, tiles == 3)
  OCR_DWPW_CASE(3, 1, 1, 16, false, 4, tiles == 4)
  OCR_DWPW_CASE(3, 2, 1, 16, false, 4, tiles == 4)
  // wide layers: two column groups per workgroup (and further column blocks in the grid)
  OCR_DWPW_CASE(3, 1, 2, 16, true, 4, tiles == 8)
  OCR_DWPW_CASE(5, 1, 1, 16, true, 4, tiles == 8)
  OCR_DWPW_CASE(5, 1, 1, 32, true, 3, tiles == 6 || tiles == 12)
#undef OCR_DWPW_CASE
  return false;
}

}  // namespace ocr
