// Fused depthwise -> pointwise block (the PPLCNetV3 pair `depthwise_conv2d` + epilogue -> `conv2d 1x1` + epilogue
// of det ops 1..20 / rec ops 1..20) for gfx950.
//
// Unfused, the depthwise kernel writes a tensor that the 1x1 conv reads straight back: 2 x 1.7 GB per pair in the
// rec 240-channel block, 2 x 0.9..1.9 GB in the thin high-resolution layers (r1 profile: a third of the kernel time
// in depthwise kernels at 0.4 of the HBM peak).  Here the depthwise result never leaves the CU:
//
//   workgroup   : a TH x 16 tile of OUTPUT pixels of one image (32*WP pixels) x WC*NT*32 output channels
//   channel loop: chunks of CK input channels, software-pipelined, ONE barrier per chunk; in iteration k:
//       steps            the depthwise taps of chunk k issued INSIDE the MFMA stream of chunk k-1 (FUSED below):
//         DW  LDS -> VALU      a thread owns PR vertically adjacent output pixels x 4 physical channels, taps in
//                              (ky, kx) ascending order from 0 - the contract's chain - read
//                              TD steps ahead of their FMAs; the depthwise epilogue (after the LAB fold: + bias, the
//                              hard-swish's product), written to LDS as the MFMA pixel operand
//         MMA LDS -> MFMA      v_mfma_f32_32x32x2_f32 against the 1x1 weights' fragment image (the same image
//                              conv_mfma_kernel reads; a wave = 32 pixels x NT column tiles), ascending k; a fragment
//                              register is refilled with chunk k's fragment right after its last MFMA
//       S  registers -> LDS    chunk k+1's haloed input region (16-byte pieces, zero outside the image), its K*K
//                              depthwise weights and its per-channel epilogue vector
//       G  global -> registers the same for chunk k+2 (k+3 with two register sets)
//   epilogue    : the 1x1 conv's own folded chain (+ bias, hard-swish product, one fma) and 16-byte stores when a tile's last
//                 chunk has been multiplied
//
// The pixel operand lives in LDS exactly as the C8I tensor would in HBM (pixel-major, 4 consecutive physical channels
// per lane), so the matrix pipe sees the same k-ordered chain as the unfused pair: results are bit-identical
// (tests/test_gpu_parity.py runs every plan with the fusion on and compares each materialised tensor).
// LDS rows are padded by 16 bytes: pixel stride = CK/4 + 1 sixteen-byte units (odd), so the 16-lane groups of
// ds_read_b128 / ds_write_b128 that walk consecutive pixels hit 16 distinct 4-bank slots.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>

#include "conv_device.h"
#include "kernels_net.h"

// Compiled twice, as kernels_net.hip: as it stands the f32 contract; with -DOCR_TU_H16 (kernels_dwpw_h16.o) precision
// "fp16" - input and output tensors stored as f16, the 1x1 half as v_mfma_f32_32x32x8_f16 on operands rounded to f16, the
// depthwise taps, both epilogues and everything in LDS f32 as before.
#ifdef OCR_TU_H16
#define OCR_L(name) name##_h16
#define OCR_H16_TWIN(cond, call)
namespace ocr {
inline namespace h16 {
constexpr bool kH16 = true;
#else
#define OCR_L(name) name
#define OCR_H16_TWIN(cond, call) if (cond) return call;
namespace ocr {
constexpr bool kH16 = false;
#endif

#ifdef OCR_DWPW_CLOCKS  // development probe: clocks per phase of the steady-state step, summed over every wave
__device__ unsigned long long ocr_dwpw_clk[8];
#define OCR_CLK(i) { const long long t_ = clock64(); clk[i] += t_ - tlast; tlast = t_; }
#else
#define OCR_CLK(i)
#endif

namespace {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): an unrolled loop whose index is a constant expression
// (#pragma unroll gives up on the 30-step bodies below, and the arrays they index would land in scratch)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct F4 { ocr_f2 lo, hi; };

// Packed f32 arithmetic AS THE INSTRUCTION.  The compiler's last peephole (si-pre-emit-peephole) splits every
// v_pk_{fma,mul,add}_f32 that follows an MFMA inside the MFMA's latency shadow into two scalar instructions, on the
// assumption that scalar VALU work co-issues with the matrix pipe.  Measured on gfx950 (tools/micro/mfma_shadow.hip,
// round 4): a VALU instruction between two v_mfma_f32_32x32x2_f32 costs ~4.5 clocks of matrix time whether it is packed
// or not (the f32 matrix pipe and the f32 FMA lanes are the same multipliers), so the split DOUBLES what the depthwise
// taps cost the matrix stream (100 v_fma_f32 per 32 MFMAs in the 5x5 block instead of 50 v_pk_fma_f32).  Inline asm is
// opaque to that pass; the arithmetic is the same two IEEE operations per instruction (bit-identical, tests/).
#ifndef OCR_DWPW_NO_PKASM
__device__ __forceinline__ ocr_f2 pk_fma(ocr_f2 a, ocr_f2 b, ocr_f2 c) {
  ocr_f2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ ocr_f2 pk_add(ocr_f2 a, ocr_f2 b) {
  ocr_f2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ ocr_f2 pk_mul(ocr_f2 a, ocr_f2 b) {
  ocr_f2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
// (scalar pair in SGPRs: one constant-bus operand per VOP3P instruction)
__device__ __forceinline__ ocr_f2 pk_add_s(ocr_f2 s, ocr_f2 b) {
  ocr_f2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "s"(s), "v"(b));
  return d;
}
#else
__device__ __forceinline__ ocr_f2 pk_fma(ocr_f2 a, ocr_f2 b, ocr_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ ocr_f2 pk_add(ocr_f2 a, ocr_f2 b) { return a + b; }
__device__ __forceinline__ ocr_f2 pk_mul(ocr_f2 a, ocr_f2 b) { return a * b; }
__device__ __forceinline__ ocr_f2 pk_add_s(ocr_f2 s, ocr_f2 b) { return b + s; }
#endif

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
  static constexpr size_t lds_floats(int nttot) {
    return 2 * (size_t)IN_TILE + 2 * (size_t)OP_TILE + 2 * (size_t)WT + 2 * (size_t)CK + (size_t)nttot * 32;
  }
};

// position of a work unit (column block fastest, then tile x, tile y, image), advanced without divisions.
// RAG (ragged batch; kernels_net.h, RagLevel): an image is a text line of its own width, or (the detector on mixed
// sizes) an image of its own height and width - its tile counts are re-derived when the walk enters a sample; sizes and
// first pixels are read where they are used (g_setup, finish: once per unit).  All of this is wave-uniform (scalar
// registers, which these kernels have none to spare of: the uniform instantiations carry no ragged state at all).
// The sample tables of a ragged batch are read with wave-uniform indices: through the constant address space these are
// scalar loads (s_load_dword, waited for on lgkmcnt).  As plain global loads the compiler issues them as vector loads +
// v_readfirstlane behind `s_waitcnt vmcnt(0)` - which drains every load in flight, the next items' region included, once per
// unit (round 5: the thin layers of a ragged batch start a unit every one to four items).
__device__ __forceinline__ int rag_ld(const int* t, int i) { return ((const __attribute__((address_space(4))) int*)t)[i]; }
__device__ __forceinline__ int rag_w_s(const RagLevel& r, int n) { return r.h ? rag_ld(r.w, n) >> r.shift : rag_ld(r.w, n); }
__device__ __forceinline__ int rag_h_s(const RagLevel& r, int n, int Hu) { return r.h ? rag_ld(r.h, n) >> r.shift : Hu; }
__device__ __forceinline__ long rag_pix0_s(const RagLevel& r, int n, int Hu) { return r.h ? (long)(rag_ld(r.cw, n) >> (2 * r.shift)) : (long)rag_ld(r.cw, n) * Hu; }

template <bool RAG, int TH>
struct UnitPos {
  int cb, tx, ty, n;
  int txn, tyn;  // RAG: tile columns / rows of sample n
  __device__ __forceinline__ int cols(const DwPwArgs& a) const { return RAG ? txn : a.tiles_x; }
  __device__ __forceinline__ int rows(const DwPwArgs& a) const { return RAG ? tyn : a.tiles_y; }
  __device__ __forceinline__ void geom(const DwPwArgs& a) {  // (the tables carry one entry past N: the walk steps onto "sample N" after the last unit)
    txn = (rag_w_s(a.rout, n) + 15) >> 4;
    tyn = (rag_h_s(a.rout, n, a.c.OH) + TH - 1) / TH;
  }
  __device__ __forceinline__ void init(unsigned u, int cblocks, const DwPwArgs& a) {
    cb = (int)(u % (unsigned)cblocks);
    unsigned t = u / (unsigned)cblocks;
    if constexpr (RAG) {
      n = rag_line(a.rtiles, a.c.N, t, 1);
      geom(a);
      t -= (unsigned)a.rtiles[n];
      ty = (int)(t / (unsigned)txn);
      tx = (int)(t - (unsigned)ty * (unsigned)txn);
    } else {
      txn = tyn = 0;
      tx = (int)(t % (unsigned)a.tiles_x);
      t /= (unsigned)a.tiles_x;
      ty = (int)(t % (unsigned)a.tiles_y);
      n = (int)(t / (unsigned)a.tiles_y);
    }
  }
  __device__ __forceinline__ void next(int cblocks, const DwPwArgs& a) {
    if (++cb == cblocks) {
      cb = 0;
      if (++tx == cols(a)) {
        tx = 0;
        if (++ty == rows(a)) {
          ty = 0; ++n;
          if constexpr (RAG) geom(a);
        }
      }
    }
  }
  // size and first pixel of image n on the input / output side
  __device__ __forceinline__ int in_w(const DwPwArgs& a) const { return RAG ? rag_w_s(a.rin, n) : a.W; }
  __device__ __forceinline__ int in_h(const DwPwArgs& a) const { return RAG ? rag_h_s(a.rin, n, a.H) : a.H; }
  __device__ __forceinline__ int out_w(const DwPwArgs& a) const { return RAG ? rag_w_s(a.rout, n) : a.c.OW; }
  __device__ __forceinline__ int out_h(const DwPwArgs& a) const { return RAG ? rag_h_s(a.rout, n, a.c.OH) : a.c.OH; }
  __device__ __forceinline__ long in_pix(const DwPwArgs& a) const { return RAG ? rag_pix0_s(a.rin, n, a.H) : (long)n * a.H * a.W; }
  __device__ __forceinline__ long out_pix(const DwPwArgs& a) const { return RAG ? rag_pix0_s(a.rout, n, a.c.OH) : (long)n * a.c.OH * a.c.OW; }
};

// The depthwise half's epilogue after the LAB fold (net.hip, fold_lab): y = acc + b', u = y * clamp(y + 3, 0, 6) - the
// scale and shift that followed the hard-swish live in the 1x1 conv's weights and bias.  Ten VALU instructions per four
// values (the unfolded chain with its division-free hard-swish and range sweep: 34).
__device__ __forceinline__ void dw_hsw6(F4& v, const ocr_f2 blo, const ocr_f2 bhi) {
  const ocr_f2 three = {3.0f, 3.0f};
  const ocr_f2 ylo = pk_add(v.lo, blo), yhi = pk_add(v.hi, bhi);
  ocr_f2 tlo = pk_add_s(three, ylo), thi = pk_add_s(three, yhi);
  tlo.x = __builtin_amdgcn_fmed3f(tlo.x, 0.0f, 6.0f); tlo.y = __builtin_amdgcn_fmed3f(tlo.y, 0.0f, 6.0f);
  thi.x = __builtin_amdgcn_fmed3f(thi.x, 0.0f, 6.0f); thi.y = __builtin_amdgcn_fmed3f(thi.y, 0.0f, 6.0f);
  v.lo = pk_mul(ylo, tlo); v.hi = pk_mul(yhi, thi);
}

}  // namespace

// A workgroup owns a contiguous range of `upw` work units (unit = one pixel tile x one column block, in
// raster order, XCD-contiguous) and runs ONE software pipeline over all (unit, channel chunk) items of the range, so
// the loads of the next tile are in flight while the current one is multiplied and stored:
//   iteration k:  [taps(k) inside MMA(k-1), fragment refills for MMA(k)]  S(k+1)  G(k+2 | k+3)
//                 [the 1x1 conv's epilogue and stores when MMA(k-1) closes a unit]  barrier
// G runs one or two items (GD) ahead of S through as many register sets, S one item ahead of the taps through two LDS
// buffers.  TD = tap steps the LDS reads run ahead, LB = workgroups per CU the register budget is cut for.
// Everything per-thread that does not depend on the tile (LDS offsets of its pieces and items) is computed once.
template <int K, int SH, int SW, int CK, bool WIDE, int NT, bool DWACT, int GD, int TD, int LB, bool RAG, bool HALF = kH16>
__global__ void __launch_bounds__(256, LB) dwpw_kernel(const DwPwArgs a) {
  using G_ = DwPwGeom<K, SH, SW, CK, WIDE>;
  using WV = typename WFrag<HALF>::T;  // precision "fp16": f16 weight fragments, one v_mfma_f32_32x32x8_f16 per octet and column tile
  using PV = typename WFrag<HALF>::T;  // a region piece (4 channels of a pixel) as it travels: float4, or four halfs of an f16 tensor
  constexpr int ES = HALF ? 2 : 4;     // bytes per stored activation
  constexpr int WC = G_::WC, TW = G_::TW, TH = G_::TH, PR = G_::PR, IW = G_::IW;
  constexpr int S = G_::S, Q = G_::Q, C8S = G_::C8S, IN_TILE = G_::IN_TILE, OP_TILE = G_::OP_TILE, WT = G_::WT;
  constexpr int G_PIECES = G_::G_PIECES, G_PER = G_::G_PER, NIT = G_::NIT, IT_PER = G_::IT_PER;
  constexpr int WQ = K * K * Q;  // 16-byte pieces of the chunk's depthwise weights
  static_assert(WQ + Q <= 256, "one weight / bias piece per thread");
  static_assert(C8S % 2 == 0, "the fragment stream is walked two octets at a time");
  static_assert(NIT % 256 == 0, "every thread owns the same number of depthwise items");
  extern __shared__ float4 s_dwpw4[];
  float* s_in = (float*)s_dwpw4;      // [2][IH*IW][S]   haloed input region of a chunk
  float* s_op = s_in + 2 * IN_TILE;   // [2][P][S]       depthwise result = MFMA pixel operand
  float* s_w = s_op + 2 * OP_TILE;    // [2][K*K][CK]    depthwise weights of a chunk
  float* s_b = s_w + 2 * WT;          // [2][CK]         depthwise bias of a chunk
  float* s_par = s_b + 2 * CK;        // [NTtot*32]      the 1x1 conv's bias, every column

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  const ConvArgs& c = a.c;
  const int Cs = c.Cs_in;
  const int nch = Cs / CK;
  const int cblocks = c.NTtot / (NT * WC);
  const int wp = WIDE ? (wave & 1) : wave, wc = WIDE ? (wave >> 1) : 0;
  // ---- this workgroup's units
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned u0 = lb * a.upw;
  const unsigned u1 = u0 + a.upw < a.nunits ? u0 + a.upw : a.nunits;
  if (u0 >= u1) return;
  const int nunits = (int)(u1 - u0);
  const int total = nunits * nch;

  // ---- per-thread plans (tile-independent)
  int g_lds[G_PER], g_pos[G_PER];  // LDS float offset of piece i; (row << 16 | column << 8 | first channel) in the region, -1: none
#pragma unroll
  for (int i = 0; i < G_PER; ++i) {
    const int idx = tid + i * 256;
    const int px = idx / Q, q = idx - px * Q;
    const int py = px / IW, pxx = px - py * IW;
    g_lds[i] = px * S + 4 * q;
    g_pos[i] = idx < G_PIECES ? ((py << 16) | (pxx << 8) | (4 * q)) : -1;
  }
  int d_in[IT_PER], d_op[IT_PER], d_q[IT_PER];
#pragma unroll
  for (int u = 0; u < IT_PER; ++u) {
    const int it = tid + u * 256;
    const int txx = it & (TW - 1), r_ = it / TW;
    const int rp = r_ % (TH / PR), q = r_ / (TH / PR);
    d_in[u] = ((rp * PR * SH) * IW + txx * SW) * S + 4 * q;
    d_op[u] = (rp * PR * TW + txx) * S + 4 * q;
    d_q[u] = 4 * q;
  }

  // ---- G: global -> registers, item by item (its own unit / chunk counters run ahead of everything else)
  UnitPos<RAG, TH> g_pos_u;
  g_pos_u.init(u0, cblocks, a);
  int g_units = nunits, g_ch = 0;
  const char* g_img = (const char*)a.dw_in;
  int goff[G_PER];
  auto g_setup = [&]() __attribute__((always_inline)) {  // image offsets (elements) of this thread's pieces for unit g_pos_u (-1: zero)
    g_img = (const char*)a.dw_in + g_pos_u.in_pix(a) * Cs * ES;
    const int iy0 = g_pos_u.ty * TH * SH - a.PH, ix0 = g_pos_u.tx * TW * SW - a.PW;
    const int iwn = g_pos_u.in_w(a), ihn = g_pos_u.in_h(a);
#pragma unroll
    for (int i = 0; i < G_PER; ++i) {
      const int iy = iy0 + (g_pos[i] >> 16), ix = ix0 + ((g_pos[i] >> 8) & 0xff);
      const bool ok = g_pos[i] >= 0 && (unsigned)iy < (unsigned)ihn && (unsigned)ix < (unsigned)iwn;
      goff[i] = ok ? (iy * iwn + ix) * Cs + (g_pos[i] & 0xff) : -1;
    }
  };
  // Every call issues the same G_PER + 1 loads, unconditionally (pieces outside the image read a valid address and are
  // zeroed afterwards; calls past the last item re-read the last one): vmcnt retires in order, and only with a fixed
  // number of loads per call can the waits for the fragments (issued BEFORE this item's region) be exact counts.  With
  // loads under lane- or item-dependent branches the compiler falls back to vmcnt(0) in the middle of the matrix phase.
  const float* const w_piece = tid < WQ ? a.dw_w + (long)(tid / Q) * Cs + 4 * (tid % Q)
                                        : a.dw_ep.bias + (tid < WQ + Q ? 4 * (tid - WQ) : 0);
  struct GSet { PV r[G_PER]; float4 w; };  // one item's region pieces in flight
  // (round 4) a piece outside the image is LOADED FROM A ZERO PAGE - the address is selected, not the value - so the
  // fill below stores what arrived: no keep mask, no AND per component (21 VALU instructions per item less; every one
  // of them cost matrix time)
  const char* const zpage = (const char*)a.c.zeros;
  auto G = [&](GSet& gs) __attribute__((always_inline)) {
    PV (&greg)[G_PER] = gs.r;
    float4& wreg = gs.w;
    const char* base = g_img + g_ch * CK * ES;
#pragma unroll
    for (int i = 0; i < G_PER; ++i)
#ifdef OCR_DWPW_NO_G  // development probe (tools/micro/dwpw_probe.hip): no input traffic
      { greg[i] = PV{}; greg[i].x = (decltype(greg[i].x))((size_t)base + goff[i]); }
#else
      greg[i] = *(const PV*)(goff[i] >= 0 ? base + (long)goff[i] * ES : zpage);
#endif
    wreg = *(const float4*)(w_piece + g_ch * CK);
    if (g_units > 0 && ++g_ch == nch) {
      g_ch = 0;
      if (--g_units > 0) {
        g_pos_u.next(cblocks, a);
        g_setup();
      } else {
        g_ch = nch - 1;  // past the end: stay on the last chunk of the last unit
        g_units = 0;
      }
    }
  };
  auto Sfill = [&](int buf, const GSet& gs) __attribute__((always_inline)) {
    const PV (&greg)[G_PER] = gs.r;
    const float4& wreg = gs.w;
    float* si = s_in + buf * IN_TILE;
#pragma unroll
    for (int i = 0; i < G_PER; ++i)
      if (G_PIECES % 256 == 0 || g_pos[i] >= 0) {
        if constexpr (HALF) {  // up to f32 on the way into LDS (exact): the taps below are the f32 ones
          const ocr_f4v f = __builtin_convertvector(ocr_as_h4(greg[i]), ocr_f4v);
          *(float4*)(si + g_lds[i]) = make_float4(f.x, f.y, f.z, f.w);
        } else {
          *(float4*)(si + g_lds[i]) = greg[i];
        }
      }
    if (tid < WQ) *(float4*)(s_w + buf * WT + 4 * tid) = wreg;            // [tap][CK]: tap*CK + 4q = 4*tid
    else if (tid < WQ + Q) *(float4*)(s_b + buf * CK + 4 * (tid - WQ)) = wreg;
  };

  // ---- depthwise conv + its LAB epilogue of one chunk: s_in[buf] -> s_op[buf]
  auto DW = [&](int buf) __attribute__((always_inline)) {
    const float* si = s_in + buf * IN_TILE;
    float* so = s_op + buf * OP_TILE;
    const float* sw = s_w + buf * WT;
    const float* sb = s_b + buf * CK;
#pragma unroll
    for (int u = 0; u < IT_PER; ++u) {
      F4 acc[PR];
#pragma unroll
      for (int o = 0; o < PR; ++o) { acc[o].lo = ocr_f2{0.f, 0.f}; acc[o].hi = ocr_f2{0.f, 0.f}; }
      const float* col = si + d_in[u];
      const float* wq = sw + d_q[u];
#ifdef OCR_DWPW_NO_TAPS  // development probe: epilogue and operand write only
      acc[0].lo.x = col[0] + wq[0];
#else
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
#endif
      const float4 b = *(const float4*)(sb + d_q[u]);
      const ocr_f2 blo = {b.x, b.y}, bhi = {b.z, b.w};
#pragma unroll
      for (int o = 0; o < PR; ++o) {
        dw_hsw6(acc[o], blo, bhi);
        *(float4*)(so + d_op[u] + o * TW * S) = make_float4(acc[o].lo.x, acc[o].lo.y, acc[o].hi.x, acc[o].hi.y);
      }
    }
  };

  // ---- 1x1 conv on the matrix cores: s_op[buf] x fragment image; its own unit / step counters (B runs a step ahead)
  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const long wstride = (long)c.NTtot * 64;
  const int KK = nch * C8S;
  UnitPos<RAG, TH> b_pos, m_pos;
  b_pos.init(u0, cblocks, a);
  m_pos = b_pos;
  int b_units = nunits, b_step = 0;
  const WV* const w_lane = (const WV*)c.wfrag + (long)wc * NT * 64 + lane;
  const WV* p_w = w_lane + (long)b_pos.cb * WC * NT * 64;
  // The fragments of a WHOLE chunk are fetched at the top of the iteration that multiplies it, before the depthwise
  // phase: the L2 round trip (the streamed input evicts them from L1) hides behind the depthwise arithmetic.
  WV bq[C8S][NT];
  auto advanceB = [&]() __attribute__((always_inline)) {  // p_w -> the fragments of the item after the one just fetched
    b_step += C8S;
    if (b_step == KK) {  // next unit: back to the first step of ITS column block (past the end: the last one again)
      b_step = 0;
      if (b_units > 1) { --b_units; b_pos.next(cblocks, a); }
      p_w = w_lane + (long)b_pos.cb * WC * NT * 64;
    } else {
      p_w += C8S * wstride;
    }
  };
  auto loadB = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < C8S; ++j) {
#pragma unroll
#ifdef OCR_DWPW_NO_B  // development probe: no fragment traffic
      for (int t = 0; t < NT; ++t) { bq[j][t] = WV{}; bq[j][t].x = (decltype(bq[j][t].x))(size_t)p_w; }
#else
      for (int t = 0; t < NT; ++t) bq[j][t] = p_w[j * wstride + t * 64];
#endif
    }
    advanceB();
  };
  auto mfma4 = [&](const WV (&bv)[NT], const float4& av) __attribute__((always_inline)) {
#ifdef OCR_DWPW_NO_MMA  // development probe: operands fetched, matrix pipe idle
    acc[0][0] += (float)bv[0].x * av.x + (float)bv[NT - 1].y * av.w;
    return;
#endif
    if constexpr (HALF) {
      const ocr_h4 ah = ocr_to_h4(av);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(ocr_as_h4(bv[t]), ah, acc[t], 0, 0, 0);
    } else
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].x, av.x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].y, av.y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].z, av.z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].w, av.w, acc[t], 0, 0, 0);
    }
  };
  const float ps6 = a.pw_ep.s6, pa1 = a.pw_ep.a1;
  const int pix = wp * 32 + p, pix_y = pix / TW, pix_x = pix & (TW - 1);
  const int op_off = pix * S + 4 * h;
  // the 1x1 conv's epilogue after the LAB fold - y = acc + b', u = y * clamp(y + 3, 0, 6), fmaf(u, s6, a1) - and 16-byte
  // stores for this lane's pixel: acc[t][4g..4g+3] = physical channels nt0*32 + 32t + 8g + 4h .. +3 (conv_device.h, conv_finish)
  auto finish = [&]() __attribute__((always_inline)) {
    const int nt0 = (m_pos.cb * WC + wc) * NT;
    const int oy = m_pos.ty * TH + pix_y, ox = m_pos.tx * TW + pix_x;
    const int own = m_pos.out_w(a);
    const bool inside = oy < m_pos.out_h(a) && ox < own;
    const int r0 = nt0 * 32 + 4 * h;
    const float* sp = s_par + r0;
    const long oidx = (m_pos.out_pix(a) + (long)oy * own + ox) * c.Cs_out + r0;  // element index of the lane's first column
    const ocr_f2 S6 = {ps6, ps6}, A1 = {pa1, pa1};
    // a column tile's four bias vectors as one group of LDS reads (left alone, each read is sunk next to its use:
    // 4*NT dependent round trips per tile)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float4 bias4[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bias4[g] = *(const float4*)(sp + 32 * t + 8 * g);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = bias4[g];
        ocr_f2 tl = ocr_f2{acc[t][4 * g], acc[t][4 * g + 1]} + ocr_f2{b.x, b.y};
        ocr_f2 th = ocr_f2{acc[t][4 * g + 2], acc[t][4 * g + 3]} + ocr_f2{b.z, b.w};
        tl = ocr_hsw6_2(tl); th = ocr_hsw6_2(th);
        tl = __builtin_elementwise_fma(tl, S6, A1); th = __builtin_elementwise_fma(th, S6, A1);
#ifdef OCR_PROBE_NOSTORE
        if (tl.x == 12345.678f)
#endif
        if (inside && nt0 * 32 + 32 * t + 8 * g < c.ColsStore) st4<HALF>(c.out, oidx + 32 * t + 8 * g, make_float4(tl.x, tl.y, th.x, th.y));
        acc[t][4 * g] = 0.f; acc[t][4 * g + 1] = 0.f; acc[t][4 * g + 2] = 0.f; acc[t][4 * g + 3] = 0.f;
      }
    }
  };
  int m_ch = 0;
  auto mma_end = [&]() __attribute__((always_inline)) {
    if (++m_ch == nch) {  // the unit is complete
      finish();
      m_ch = 0;
      m_pos.next(cblocks, a);
    }
  };
  auto MMA = [&](int buf) __attribute__((always_inline)) {
    const float* so = s_op + buf * OP_TILE + op_off;
    float4 av[C8S];
#pragma unroll
    for (int j = 0; j < C8S; ++j) av[j] = *(const float4*)(so + 8 * j);
#pragma unroll
    for (int j = 0; j < C8S; ++j) mfma4(bq[j], av[j]);
    mma_end();
  };

  // ---- steady state: everything else of an iteration INSIDE the matrix instructions of chunk k-1.
  // Run one after the other, a wave spends the depthwise phase waiting for LDS (one round trip per tap, 25 for a 5x5:
  // 4960 of the 12300 clocks of an item, measured with clock64 around the phases), the matrix phase waiting for the
  // matrix pipe, then the fragment loads, the LDS fill and the next global loads, and 2-3 waves per SIMD cannot cover
  // all of that.  Here the iteration is a sequence of steps, each followed by its share of the chunk's MFMAs and kept
  // apart by sched_barrier, so that the step issues in the shadow of the 64-cycle matrix instructions:
  //   tap steps      the LDS reads of tap s+D, the packed FMAs of tap s (chunk k)
  //   fill / load    registers -> LDS of item k+1, global loads of item k+2
  //   epilogue       the depthwise epilogue of chunk k and its operand write
  // and the fragment registers of chunk k-1 are refilled with chunk k's as soon as their last MFMA has issued.
  // Arithmetic and its order are those of DW and MMA.
  static_assert(IT_PER == 1, "one depthwise item per thread and chunk");
#ifdef OCR_DWPW_CLOCKS
  long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  auto FUSED = [&](int dbuf, int mbuf, int fbuf, GSet& gs) __attribute__((always_inline)) {
    const float* col = s_in + dbuf * IN_TILE + d_in[0];
    const float* wq = s_w + dbuf * WT + d_q[0];
    const float* sb = s_b + dbuf * CK;
    float* so = s_op + dbuf * OP_TILE;
    const float* som = s_op + mbuf * OP_TILE + op_off;
    constexpr int MPF = HALF ? 1 : 4;  // matrix instructions per fragment register (f16: one per octet)
    constexpr int ROWS = (PR - 1) * SH + K, NS = ROWS * K, D = TD, RS = D + 1, NM = C8S * NT * MPF, NST = NS + 1;
    const WV* const pb = p_w;  // the next item's fragments
    float4 av[C8S];
#pragma unroll
    for (int j = 0; j < C8S; ++j) av[j] = *(const float4*)(som + 8 * j);
    ocr_h4 ah[C8S];
    if constexpr (HALF) {
#pragma unroll
      for (int j = 0; j < C8S; ++j) ah[j] = ocr_to_h4(av[j]);
    }
    F4 dacc[PR];
#pragma unroll
    for (int o = 0; o < PR; ++o) { dacc[o].lo = ocr_f2{0.f, 0.f}; dacc[o].hi = ocr_f2{0.f, 0.f}; }
    float4 tv[RS], tw[RS][PR];
    auto fetch = [&](int st, int slot) __attribute__((always_inline)) {
      const int r = st / K, kx = st - r * K;
      tv[slot] = *(const float4*)(col + (r * IW + kx) * S);
#pragma unroll
      for (int o = 0; o < PR; ++o) {
        const int ky = r - o * SH;
        if (ky >= 0 && ky < K) tw[slot][o] = *(const float4*)(wq + (ky * K + kx) * CK);
      }
    };
    auto comp = [](const float4& v, int c) __attribute__((always_inline)) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; };
#ifndef OCR_DWPW_NO_TAPS
#pragma unroll
    for (int st = 0; st < D; ++st) fetch(st, st % RS);
#endif
    __builtin_amdgcn_sched_barrier(0);
    static_for<NST>([&](auto st_) __attribute__((always_inline)) {
      constexpr int st = decltype(st_)::value;
      if constexpr (st < NS) {
#ifndef OCR_DWPW_NO_TAPS  // development probe (tools/micro/dwpw_probe.hip): no tap reads, no tap FMAs in the steady state
        if (st + D < NS) fetch(st + D, (st + D) % RS);
        const int r = st / K, slot = st % RS;
        const float4 v = tv[slot];
#pragma unroll
        for (int o = 0; o < PR; ++o) {
          const int ky = r - o * SH;
          if (ky < 0 || ky >= K) continue;
          const float4 w = tw[slot][o];
          dacc[o].lo = pk_fma(ocr_f2{v.x, v.y}, ocr_f2{w.x, w.y}, dacc[o].lo);
          dacc[o].hi = pk_fma(ocr_f2{v.z, v.w}, ocr_f2{w.z, w.w}, dacc[o].hi);
        }
#endif
      } else {  // the depthwise epilogue and the operand write, as in DW
        const float4 b = *(const float4*)(sb + d_q[0]);
        const ocr_f2 blo = {b.x, b.y}, bhi = {b.z, b.w};
#pragma unroll
        for (int o = 0; o < PR; ++o) {
          dw_hsw6(dacc[o], blo, bhi);
          *(float4*)(so + d_op[0] + o * TW * S) = make_float4(dacc[o].lo.x, dacc[o].lo.y, dacc[o].hi.x, dacc[o].hi.y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<(st + 1) * NM / NST - st * NM / NST>([&](auto m_) __attribute__((always_inline)) {  // k-ascending per accumulator: octet, column tile, component
        constexpr int m = st * NM / NST + decltype(m_)::value;
        constexpr int j = m / (MPF * NT), t = (m / MPF) % NT, c4 = m % MPF;
#ifdef OCR_DWPW_NO_MMA  // development probe: operands fetched, matrix pipe idle
        if (m == 0) acc[t][0] += (float)bq[j][t].x * av[j].x;
#else
        if constexpr (HALF) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(ocr_as_h4(bq[j][t]), ah[j], acc[t], 0, 0, 0);
        else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(bq[j][t], c4), comp(av[j], c4), acc[t], 0, 0, 0);
#endif
        if (c4 == MPF - 1)  // the last use of this fragment register: refill it
#ifdef OCR_DWPW_NO_B
          { bq[j][t] = WV{}; bq[j][t].x = (decltype(bq[j][t].x))(size_t)pb; }
#else
          bq[j][t] = pb[j * wstride + t * 64];
#endif
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    OCR_CLK(0)
    advanceB();
    OCR_CLK(1)
    // (inside the step sequence these two cost 40-60 registers - measured - and with them the second or third wave)
    Sfill(fbuf, gs);
    OCR_CLK(2)
    G(gs);
    OCR_CLK(3)
    mma_end();
  };

  // ---- pipeline
  for (int i = tid; i < c.NTtot * 32; i += 256) s_par[i] = i < c.ColsStore ? a.pw_ep.bias[i] : 0.f;
  g_setup();
  // Iteration k: depthwise of item k inside the matrix work of item k-1, then the fragments of item k (used by the
  // next iteration), the LDS fill of item k+1 and the global loads of item k+2 (k+3 with two register sets).  The
  // steady-state step has NO branch around a vector-memory instruction - G, Sfill and loadB run unconditionally (past
  // the last item they re-read it / fill a buffer nobody reads) and the first and last iterations are peeled - so that
  // every s_waitcnt vmcnt the compiler places is an exact in-order count instead of vmcnt(0).
  auto step = [&](int k, GSet& gs) __attribute__((always_inline)) {
    OCR_CLK(7)
    FUSED(k & 1, (k - 1) & 1, (k + 1) & 1, gs);
    OCR_CLK(5)
    __syncthreads();
    OCR_CLK(4)
  };
  GSet gA;
#pragma unroll
  for (int i = 0; i < G_PER; ++i) gA.r[i] = PV{};
  gA.w = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (GD == 2) {
    GSet gB = gA;
    G(gA);  // item 0
    G(gB);  // item 1
    Sfill(0, gA);
    G(gA);  // item 2
    __syncthreads();
    DW(0);
    loadB();
    Sfill(1, gB);
    G(gB);  // item 3
    __syncthreads();
    int k = 1;
    for (; k + 1 < total; k += 2) {  // odd k: set A holds item k+1, even k: set B
      step(k, gA);
      step(k + 1, gB);
    }
    if (k < total) step(k, gA);
  } else {  // one register set: G runs one item ahead of S (long iterations: the matrix work covers the round trip)
    G(gA);  // item 0
    Sfill(0, gA);
    G(gA);  // item 1
    __syncthreads();
    DW(0);
    loadB();
    Sfill(1, gA);
    G(gA);  // item 2
    __syncthreads();
    for (int k = 1; k < total; ++k) step(k, gA);
  }
  MMA((total - 1) & 1);
#ifdef OCR_DWPW_CLOCKS
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) atomicAdd(&ocr_dwpw_clk[i], (unsigned long long)clk[i]);
    atomicAdd(&ocr_dwpw_clk[6], (unsigned long long)(total - 1));
  }
#endif
}

namespace {

template <int K, int SH, int SW, int CK, bool WIDE, int NT, bool DWACT, int GD, int TD, int LB, bool RAG, bool HALF>
bool launch_one(const DwPwArgs& a0, hipStream_t s, bool query) {
  using G_ = DwPwGeom<K, SH, SW, CK, WIDE>;
#ifdef OCR_DWPW_LDSPAD  // development probe: extra dynamic LDS so that a CU holds one workgroup
  const size_t lds = G_::lds_floats(a0.c.NTtot) * sizeof(float) + OCR_DWPW_LDSPAD;
#else
  const size_t lds = G_::lds_floats(a0.c.NTtot) * sizeof(float);
#endif
  // per device: the dynamic-LDS limit, and how many of these workgroups a CU holds.  One instantiation serves several
  // column-tile counts (its parameter block, and with it the LDS size, grows with NTtot): the attribute memo re-raises
  // for a larger request (lds_attr.h) and the occupancy memo is per LDS size.  Detector lanes and pool workers launch
  // from several host threads: the memo is guarded.
  static LdsAttrMemo attr_state;
  struct Occ { size_t lds; int per_cu, cus; };
  static Occ occ[64][2] = {};
  static std::mutex occ_mu;
  const int dev = rt_current_device();  // logical (hip_guard.h)
  if (dev < 0 || dev >= 64) return false;
  if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)dwpw_kernel<K, SH, SW, CK, WIDE, NT, DWACT, GD, TD, LB, RAG, HALF>, (int)lds, attr_state)) return false;
  int per_cu_dev = 0, cus_dev = 0;
  {
    std::lock_guard<std::mutex> lk(occ_mu);
    Occ* e = nullptr;
    for (Occ& o : occ[dev])
      if (o.per_cu && o.lds == lds) e = &o;
    if (!e) {
      int nb = 0;
      hipDeviceProp_t prop;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dwpw_kernel<K, SH, SW, CK, WIDE, NT, DWACT, GD, TD, LB, RAG, HALF>, 256, lds) != hipSuccess || nb < 1 ||
          hipGetDeviceProperties(&prop, rt_physical_device(dev)) != hipSuccess) { (void)hipGetLastError(); return false; }
      e = occ[dev][0].per_cu ? &occ[dev][1] : &occ[dev][0];  // (two sizes per instantiation on the plans' shapes; a third replaces the second)
      *e = Occ{lds, nb, prop.multiProcessorCount};
    }
    per_cu_dev = e->per_cu;
    cus_dev = e->cus;
  }
  if (query) return true;  // asked at bind time, on the device that will run it: a refusal falls back to the pair
  DwPwArgs a = a0;
  a.tiles_x = (a.c.OW + G_::TW - 1) / G_::TW;
  a.tiles_y = (a.c.OH + G_::TH - 1) / G_::TH;
  const unsigned cblocks = (unsigned)a.c.NTtot / (NT * G_::WC);
  static_assert(G_::TW == 16, "DwPwArgs::rtiles counts tiles of 16 pixel columns");
  const long nunits = (a.rtiles ? (long)a.rtiles_total : (long)a.c.N * a.tiles_x * a.tiles_y) * cblocks;
  if (nunits <= 0 || nunits > 0x7fffffffL) return false;
  a.nunits = (unsigned)nunits;
  // Units per workgroup: enough (unit, chunk) items for the pipeline to run in steady state, few enough that the grid
  // stays many times larger than the chip - other streams' kernels take CU slots at any moment, and a grid sized to
  // "exactly resident" then runs its leftover workgroups as a second full-length wave (measured: 2x on rec.07 with
  // the odd-width lane active).  OCR_DWPW_ITEMS overrides the target item count (A/B; results are identical).
  const int target = rt_options().dwpw_items;
  const int nch = a.c.Cs_in / CK;
  long upw = (target + nch - 1) / nch;
  const long resident = (long)cus_dev * per_cu_dev;
  while (upw > 1 && (nunits + upw - 1) / upw < 8 * resident) --upw;  // small problems: keep every CU busy
  // OCR_DWPW_FORCE_UPW (tests): units per workgroup regardless of the problem size, so that small inputs run the
  // multi-unit pipelines (unit boundaries inside a workgroup, a shorter last workgroup) that production batches run
  if (rt_options().dwpw_force_upw > 0) upw = rt_options().dwpw_force_upw;
  a.upw = (unsigned)upw;
  const dim3 grid((unsigned)((nunits + upw - 1) / upw));
  hipLaunchKernelGGL((dwpw_kernel<K, SH, SW, CK, WIDE, NT, DWACT, GD, TD, LB, RAG, HALF>), grid, dim3(256), lds, s, a);
  return true;
}

}  // namespace

#ifndef OCR_TU_H16
#include "dwpw2_kernel.h"

namespace {
template <int K, int SH, int SW, int CK, bool WIDE, int NT, int NB, int TD, int LB, bool RAG>
bool launch_two(const DwPwArgs& a0, hipStream_t s, bool query) {
  using G_ = DwPw2Geom<K, SH, SW, CK, WIDE, NB>;
  if (!(CK == 16 ? a0.dw_wq16 : a0.dw_wq32)) return false;
  // whole-line stores through per-wave LDS tiles (dwpw2_kernel.h, XP): the layers with ONE column tile only.  Measured on
  // every instance whose LDS leaves room (tools/micro/dwpw_probe.hip, seeded data): 16 -> 32 0.372 -> 0.358 ms (480 x 480:
  // 0.69 -> 0.645), 32 -> 64 0.75 -> 0.73 in the probe and nothing in the step (a third workgroup per CU no longer fits),
  // 64 -> 64 and 128 -> 240 unchanged, the 5x5 240-channel block +1.2 %.
  constexpr bool XP = K == 3 && CK == 16 && !WIDE && NT == 1;
  const size_t lds = G_::lds_bytes(a0.c.NTtot, XP);
  static LdsAttrMemo attr_state;
  struct Occ { size_t lds; int per_cu, cus; };
  static Occ occ[64][2] = {};
  static std::mutex occ_mu;
  const int dev = rt_current_device();  // logical (hip_guard.h)
  if (dev < 0 || dev >= 64) return false;
  if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)dwpw2_kernel<K, SH, SW, CK, WIDE, NT, NB, TD, LB, RAG, XP>, (int)lds, attr_state)) return false;
  int per_cu_dev = 0, cus_dev = 0;
  {
    std::lock_guard<std::mutex> lk(occ_mu);
    Occ* e = nullptr;
    for (Occ& o : occ[dev])
      if (o.per_cu && o.lds == lds) e = &o;
    if (!e) {
      int nb = 0;
      hipDeviceProp_t prop;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dwpw2_kernel<K, SH, SW, CK, WIDE, NT, NB, TD, LB, RAG, XP>, 256, lds) != hipSuccess || nb < 1 ||
          hipGetDeviceProperties(&prop, rt_physical_device(dev)) != hipSuccess) { (void)hipGetLastError(); return false; }
      e = occ[dev][0].per_cu ? &occ[dev][1] : &occ[dev][0];
      *e = Occ{lds, nb, prop.multiProcessorCount};
    }
    per_cu_dev = e->per_cu;
    cus_dev = e->cus;
  }
  if (query) return true;
  DwPwArgs a = a0;
  a.tiles_x = (a.c.OW + G_::TW - 1) / G_::TW;
  a.tiles_y = (a.c.OH + G_::TH - 1) / G_::TH;
  const unsigned cblocks = (unsigned)a.c.NTtot / (NT * G_::WC);
  const long nunits = (a.rtiles ? (long)a.rtiles_total : (long)a.c.N * a.tiles_x * a.tiles_y) * cblocks;
  if (nunits <= 0 || nunits > 0x7fffffffL) return false;
  a.nunits = (unsigned)nunits;
  const int target = rt_options().dwpw_items;  // (units per workgroup: as launch_one)
  const int nch = a.c.Cs_in / CK;
  long upw = (target + nch - 1) / nch;
  const long resident = (long)cus_dev * per_cu_dev;
  while (upw > 1 && (nunits + upw - 1) / upw < 8 * resident) --upw;
  if (rt_options().dwpw_force_upw > 0) upw = rt_options().dwpw_force_upw;
  a.upw = (unsigned)upw;
  const dim3 grid((unsigned)((nunits + upw - 1) / upw));
  hipLaunchKernelGGL((dwpw2_kernel<K, SH, SW, CK, WIDE, NT, NB, TD, LB, RAG, XP>), grid, dim3(256), lds, s, a);
  return true;
}
}  // namespace

bool lab_from_epilogue(const Epilogue& ep, LabEp& out) {
  out = LabEp{};
  if (ep.n != 2 && ep.n != 3) return false;
  if (ep.st[0].kind != EP_BIAS || ep.st[1].kind != EP_ACT || ep.st[1].act != ACT_HSW6) return false;
  out.bias = ep.st[0].v0;
  if (ep.n == 3) {
    if (ep.st[2].kind != EP_SFMA) return false;
    out.sfma = 1;
    out.s6 = ep.st[2].p0;
    out.a1 = ep.st[2].p1;
  }
  return true;
}
#endif  // OCR_TU_H16

// The instantiated shapes (everything else stays an unfused pair): K, strides, chunk width, workgroup shape
// (thin: 4 waves = 4 pixel groups of an 8x16 tile, all <= 4 column tiles per wave; wide: 2 pixel groups of a 4x16
// tile x 2 column groups), column tiles per wave.  `query` only answers whether the pair is on this path.
// the instance table: launches, answers the bind-time query, or (rows_only) just names the pixel-tile height
static int dwpw_dispatch(const DwPwArgs& a, hipStream_t s, bool query, bool rows_only) {
  const int K = a.K, SH = a.SH, SW = a.SW, Cs = a.c.Cs_in, tiles = a.c.NTtot;
  if (!a.pw_ep.sfma || a.dw_ep.sfma) return 0;  // the pairs on the hot path after the LAB fold: bias | hsw6 in the depthwise half, bias | hsw6 | sfma behind the 1x1 conv
#ifndef OCR_TU_H16
  // second form (dwpw2_kernel.h): region and weights by LDS-DMA.  Same tile shapes as the first form's instances below (the
  // ragged tile tables are shared); f32 build, a "dwq16:" image of the depthwise weights present.  OCR_DWPW2=0: first form (A/B).
#ifndef OCR_DWPW2_55  // region buffers, TD, LB of the 240-channel 5x5 block (tools/micro/dwpw_probe.hip builds variants)
#define OCR_DWPW2_55 3, 2, 2
#endif
#define OCR_DWPW2_CASE(K_, SH_, SW_, CK_, WIDE_, NT_, NB_, TD_, LB_, COND)                                  \
  if (K == K_ && SH == SH_ && SW == SW_ && Cs % CK_ == 0 && tiles % (NT_ * (WIDE_ ? 2 : 1)) == 0 && (COND)) { \
    if (rows_only) return DwPw2Geom<K_, SH_, SW_, CK_, WIDE_, NB_>::TH;                                         \
    if ((a.rtiles ? launch_two<K_, SH_, SW_, CK_, WIDE_, NT_, NB_, TD_, LB_, true>(a, s, query)                \
                  : launch_two<K_, SH_, SW_, CK_, WIDE_, NT_, NB_, TD_, LB_, false>(a, s, query))) return 1;   \
  }
#define OCR_DWPW2_CASE_X(...) OCR_DWPW2_CASE(__VA_ARGS__)
  // (tile shapes, chunk widths and TD as the first form's table below; three region buffers where two workgroups per CU
  // still fit in LDS, else two.  A refusal - no weights image of that chunk width - falls through to the first form)
  if ((a.dw_wq16 || a.dw_wq32) && rt_options().dwpw2 && !a.c.half) {
    if (!rt_options().dwpw_t4_thin) {
      OCR_DWPW2_CASE(3, 1, 1, 32, true, 2, 3, 1, 2, tiles == 4)
      OCR_DWPW2_CASE(3, 2, 1, 32, true, 2, 2, 1, 2, tiles == 4)
    }
#ifndef OCR_DWPW2_THIN  // region buffers, TD, LB of the thin 3x3 layers with one / two column tiles (probe variants)
#define OCR_DWPW2_THIN 2, 2, 3
#endif
    OCR_DWPW2_CASE_X(3, 1, 1, 16, false, 1, OCR_DWPW2_THIN, tiles == 1)
    OCR_DWPW2_CASE_X(3, 1, 1, 16, false, 2, OCR_DWPW2_THIN, tiles == 2)
    OCR_DWPW2_CASE(3, 1, 1, 16, false, 3, 3, 2, 2, tiles == 3)
    OCR_DWPW2_CASE(3, 1, 1, 16, false, 4, 3, 2, 2, tiles == 4)
    OCR_DWPW2_CASE(3, 2, 1, 16, false, 4, 2, 2, 2, tiles == 4)
    OCR_DWPW2_CASE(3, 1, 2, 16, true, 4, 3, 1, 2, tiles == 8)
    OCR_DWPW2_CASE_X(5, 1, 1, 16, true, 4, OCR_DWPW2_55, tiles == 8)
    OCR_DWPW2_CASE(5, 1, 1, 32, true, 3, 2, 2, 2, tiles == 6 || tiles == 12)
  }
#undef OCR_DWPW2_CASE_X
#undef OCR_DWPW2_CASE
#endif
#define OCR_DWPW_CASE(K_, SH_, SW_, CK_, WIDE_, NT_, GD_, TD_, LB_, COND)                                  \
  if (K == K_ && SH == SH_ && SW == SW_ && Cs % CK_ == 0 && tiles % (NT_ * (WIDE_ ? 2 : 1)) == 0 && (COND)) { \
    if (rows_only) return DwPwGeom<K_, SH_, SW_, CK_, WIDE_>::TH;                                                \
    return (a.rtiles ? launch_one<K_, SH_, SW_, CK_, WIDE_, NT_, true, GD_, TD_, LB_, true, kH16>(a, s, query)           \
                     : launch_one<K_, SH_, SW_, CK_, WIDE_, NT_, true, GD_, TD_, LB_, false, kH16>(a, s, query)) ? 1 : 0;  \
  }
  // TD = how many tap steps ahead the LDS reads run, LB = workgroups per CU the register budget is cut for (3: 168
  // registers, 2: 256): per shape, whichever measured faster (tools/micro/dwpw_probe) - a third wave per SIMD where the
  // kernel fits without spilling, deeper read-ahead where it does not.
  // 4-tile layers: a 4x16 tile with 32-channel chunks and 2 x 2 column tiles per wave beats the thin shape by 15-25 %
  // (half the B-fragment traffic per MFMA); OCR_DWPW_T4=thin keeps the thin shape for A/B runs
  if (!rt_options().dwpw_t4_thin) {
    OCR_DWPW_CASE(3, 1, 1, 32, true, 2, 1, 1, 3, tiles == 4)
    OCR_DWPW_CASE(3, 2, 1, 32, true, 2, 2, 1, 2, tiles == 4)  // (ragged batch, 1024 lines: 1/2/2 0.846 ms, 2/1/2 0.810, 1/1/3 1.15)
  }
  // thin layers: tiles <= 4, one wave owns every output column of its 32 pixels
  OCR_DWPW_CASE(3, 1, 1, 16, false, 1, 2, 2, 2, tiles == 1)
  OCR_DWPW_CASE(3, 1, 1, 16, false, 2, 2, 1, 3, tiles == 2)
  OCR_DWPW_CASE(3, 1, 1, 16, false, 3, 2, 2, 2, tiles == 3)
  OCR_DWPW_CASE(3, 1, 1, 16, false, 4, 1, 2, 2, tiles == 4)
  OCR_DWPW_CASE(3, 2, 1, 16, false, 4, 1, 2, 2, tiles == 4)
  // wide layers: two column groups per workgroup (and further column blocks in the grid)
  OCR_DWPW_CASE(3, 1, 2, 16, true, 4, 2, 1, 2, tiles == 8)  // (ragged batch in the step: 1/2/2 1.92 ms, 2/1/2 1.81, 2/2/2 1.83)
#ifndef OCR_DWPW_55  // GD, TD, LB of the 240-channel 5x5 blocks (tools/micro/dwpw_probe.hip builds variants with -DOCR_DWPW_55=g,t,l)
#define OCR_DWPW_55 2, 2, 2
#endif
#define OCR_DWPW_CASE_X(...) OCR_DWPW_CASE(__VA_ARGS__)
#ifdef OCR_DWPW_TRY32  // development probe: 32-channel chunks with two pixels per thread on a 256-column layer
  OCR_DWPW_CASE_X(5, 1, 1, 32, true, 4, OCR_DWPW_TRY32, tiles == 8 && Cs % 32 == 0)
#endif
  OCR_DWPW_CASE_X(5, 1, 1, 16, true, 4, OCR_DWPW_55, tiles == 8)  // (round 3, ragged batch, in the step: 1/1/3 3.13 ms, 2/2/2 3.05, 1/2/3 3.67, 2/1/3 3.90)
  OCR_DWPW_CASE(5, 1, 1, 32, true, 3, 1, 2, 2, tiles == 6 || tiles == 12)
#undef OCR_DWPW_CASE_X
#undef OCR_DWPW_CASE
  return 0;
}
bool OCR_L(launch_dwpw)(const DwPwArgs& a, hipStream_t s, bool query) {
  OCR_H16_TWIN(a.c.half, launch_dwpw_h16(a, s, query))
  return dwpw_dispatch(a, s, query, false) != 0;
}
#ifndef OCR_TU_H16
int dwpw_tile_rows(const DwPwArgs& a) { return dwpw_dispatch(a, nullptr, true, true); }  // (the tile shapes are the same in both builds)
#endif

#ifdef OCR_TU_H16
}  // namespace h16
#endif
}  // namespace ocr
