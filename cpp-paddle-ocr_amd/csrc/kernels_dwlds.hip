// Depthwise 5x5 conv with the input region staged through LDS (gfx950), for the recognizer's low, wide maps: rec ops 26 / 31 /
// 33 - 6- and 3-row maps of 480 channels at the 48-pixel line height (and the classifier's 3-row layers).
//
// dw_conv_kernel (kernels_net.hip) gives a thread a register patch and lets it read the patch's haloed rows straight from
// global memory: on a 6-row map a 2-row band re-reads four of its six input rows, and the three bands of a line are too far
// apart in time for the 4 MB L2 of an XCD to hold what the first one fetched (rec op 26: 4.13 GB read for a 1.89 GB input,
// round 3 PMC).  Full-height register patches removed the re-reads and were SLOWER (a third as many threads, each walking
// 20 strips: round 4).  Here a workgroup is ONE wave (its barriers cost nothing; 7-12 of them per CU by
// their LDS) that owns a WHOLE image - a text line of HR = 6 or 3 rows - x a 32-channel chunk and walks it LEFT TO RIGHT in
// tiles of 16 output columns:
//   G   global -> registers   the haloed input region of the NEXT tile (HR rows x 20 columns; a pixel's 32 channels are one
//                             128-byte line; a lane owns the same three 16-byte pieces of every row), issued before the
//                             taps of the current tile so that the loads travel while it is computed
//   taps LDS -> VALU          a thread owns an R x TO patch of output pixels x 4 physical channels and walks its input rows
//                             out of LDS exactly as dw_conv_kernel walks them out of global memory - for every output the
//                             taps arrive in (ky, kx) ascending order from 0, the contract's chain;
//                             rows outside the image are one shared zero row
//   epilogue                  the plan's stage list on the patch (dw_epilogue.h, shared with dw_conv_kernel), 16-byte stores
//   row sums (ROWSUM)         the tile's outputs go through LDS (the region's space, dead by then) to the lane that owns
//                             (row, channel quad), which adds them in x order: s = s + v from x = 0, the global average
//                             pool's first pass in the contract's order, carried across the tiles in a register
//   S   registers -> LDS      the next tile's region
// Every input row is read from HBM once (rec op 26: 2.14 GB read for its 1.89 GB input; the K - 1 columns two neighbouring tiles
// share come back from L1 / L2), and the taps cost LDS reads instead of L1 round trips.  Results are bit-identical to
// dw_conv_kernel's (tests/test_gpu_parity.py: every plan, and OCR_DW_LDS=0 in the A/B test).
#include <hip/hip_runtime.h>

#include "conv_device.h"
#include "dw_epilogue.h"
#include "kernels_net.h"

// Compiled twice, as kernels_net.hip: as it stands the f32 contract; with -DOCR_TU_H16 precision "fp16" (f16 tensors).
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

namespace {

template <int K, int SH, int R, int NB, int TO, int NS>
struct DwLdsGeom {
  static constexpr int CC = 32, CQ = CC / 4;                   // channels / 16-byte quads per chunk (a pixel's chunk = one 128-byte line)
  static constexpr int TY = R * NB, TX = TO * NS;              // output rows / columns of a tile
  static constexpr int NTHR = NB * NS * CQ;                    // a thread = (quad, band, strip)
  static constexpr int RH = (TY - 1) * SH + K, RW = TX + K - 1;  // haloed input region of a tile (stride 1 along x)
  static constexpr int PS = CC;                                // floats per staged pixel (the 8 lanes of a pixel read 128 contiguous bytes: no padding needed)
  static constexpr int WT = K * K * CC;
  // LDS: `slots` pixel slots for the region's real rows (and, in their place once the taps are done, the row-sum exchange
  // tile), one zero row for the rows outside the image, the chunk's weights
  static constexpr int slots(int rows_real) { return rows_real * RW > TY * TX ? rows_real * RW : TY * TX; }
  static constexpr size_t lds_floats(int rows_real) { return (size_t)(slots(rows_real) + RW) * PS + WT; }
};

}  // namespace

// zslot: the pixel slot where the zero row starts = the slots in front of it (launcher: DwLdsGeom::slots(min(RH, H)))
// HR: the image's height (compile time: 3 | 6) - the whole image is ONE band group, every region row that exists is one of
// its HR rows, and the next tile's pieces (HR rows x NIT per lane) travel in registers while this tile is computed
template <int K, int SH, int R, int NB, int TO, int NS, int HR, bool ROWSUM, bool RAG, bool H16 = kH16>
__global__ void __launch_bounds__(NB * NS * 8) __attribute__((amdgpu_waves_per_eu(2, 4))) dw_lds_kernel(const DwArgs a, const Epilogue ep, const int zslot) {
  using G_ = DwLdsGeom<K, SH, R, NB, TO, NS>;
  constexpr int CC = G_::CC, CQ = G_::CQ, TY = G_::TY, TX = G_::TX, NTHR = G_::NTHR, RH = G_::RH, RW = G_::RW, PS = G_::PS;
  constexpr int NIN = TO + K - 1, NROWS = (R - 1) * SH + K;
  using F4 = DwF4;
  using RV = typename WFrag<H16>::T;
  extern __shared__ float4 s_dwl4[];
  float* s_reg = (float*)s_dwl4;                      // [zslot + RW][PS]: real rows, then the zero row
  float* s_w = s_reg + (size_t)(zslot + RW) * PS;      // [K*K][CC]
  const int tid = threadIdx.x, q = tid & (CQ - 1), slot = tid / CQ, band = slot / NS, strip = slot - band * NS;
  const int Cs = a.Cs;
  const int nch = (Cs + CC - 1) / CC, bgs = (a.OH + TY - 1) / TY;
  unsigned wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ch = (int)(wg % (unsigned)nch);
  wg /= (unsigned)nch;
  const int bg = (int)(wg % (unsigned)bgs), n = (int)(wg / (unsigned)bgs);
  const int c0 = ch * CC, pc = c0 + 4 * q;
  const bool qok = pc < Cs;
  const int pcs = qok ? pc : 0;  // (idle quads of the last chunk read valid parameters and store nothing)
  int IW = a.W, OW = a.OW;
  long ipix0 = (long)n * a.H * a.W, opix0 = (long)n * a.OH * a.OW, orow0 = (long)n * a.OH;
  if constexpr (RAG) {  // ragged batch of lines: the line's own widths (one height)
    IW = rag_w(a.rin, n); OW = rag_w(a.rout, n);
    ipix0 = rag_pix0(a.rin, n, a.H); opix0 = rag_pix0(a.rout, n, a.OH); orow0 = rag_row0(a.rout, n, a.OH);
  }
  const int IH = a.H, OHn = a.OH;
  const int yb = bg * TY, iy0 = yb * SH - a.PH;  // first output row of the band group, first input row of its region
  // region row ry <-> input row iy0 + ry; the rows inside the image are [ry_lo, ry_hi) and live in LDS rows 0 .. ry_hi - ry_lo - 1
  const int ry_lo = iy0 < 0 ? -iy0 : 0, ry_hi = IH - iy0 < RH ? IH - iy0 : RH;
  constexpr int nreal = HR;  // (host: H == HR, one band group; ry_hi - ry_lo == HR, HR * RW <= zslot)
  // ---- once: the chunk's weights [tap][CC] and the zero row
  for (int i = tid; i < K * K * CQ; i += NTHR) {
    const int tap = i / CQ, qq = i - tap * CQ, pcw = c0 + 4 * qq;
    *(float4*)(s_w + tap * CC + 4 * qq) = pcw < Cs ? *(const float4*)(a.w + (long)tap * Cs + pcw) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < RW * CQ; i += NTHR) *(float4*)(s_reg + (size_t)zslot * PS + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
  // ---- the region of a tile, global -> registers -> LDS, two region rows at a time.  A row is RW * CQ 16-byte pieces (a
  // pixel's chunk = CQ consecutive pieces = one 128-byte line); a lane owns the same NIT pieces of EVERY row, so what it
  // needs per piece is fixed for the whole kernel: column, channel quad, LDS offset inside the row.  (Indexed over the whole
  // region instead, the per-piece address arithmetic is hoisted out of the tile loop: 100 live registers and spills.)
  constexpr int RPIECES = RW * CQ, NIT = (RPIECES + NTHR - 1) / NTHR;
  static_assert(PS == 4 * CQ, "a row piece index is its LDS float4 index inside the row");
  static_assert(NTHR == NB * NS * 8, "the launch bound above (a macro argument cannot hold the geometry's template arguments)");
  int p_rx[NIT], p_off[NIT];
  bool p_ok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int p = tid + it * NTHR;
    p_rx[it] = p / CQ;
    p_off[it] = 4 * (p % CQ);
    p_ok[it] = p < RPIECES && c0 + p_off[it] < Cs;
  }
  RV greg[HR][NIT];
  auto G = [&](int tx) __attribute__((always_inline)) {
    const int ix0 = tx * TX - a.PW;
    long col[NIT];
    bool okx[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int ix = ix0 + p_rx[it];
      okx[it] = p_ok[it] && ix >= 0 && ix < IW;
      col[it] = (long)(okx[it] ? ix : 0) * Cs + c0 + p_off[it];
    }
#pragma unroll
    for (int ry = 0; ry < HR; ++ry) {
      const long rowbase = (ipix0 + (long)(iy0 + ry_lo + ry) * IW) * Cs;
#pragma unroll
      for (int it = 0; it < NIT; ++it) greg[ry][it] = okx[it] ? ld4_raw<H16>(a.in, rowbase + col[it]) : RV{};
    }
  };
  auto S = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ry = 0; ry < HR; ++ry)
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (tid + it * NTHR < RPIECES) *(float4*)(s_reg + ((size_t)ry * RW + p_rx[it]) * PS + p_off[it]) = up4(greg[ry][it]);
  };
  // ---- the thread's patch: output rows yb + band * R + r, columns tile * TX + strip * TO + o
  const int y0 = yb + band * R;
  const float* wq = s_w + 4 * q;
  F4 rsum;
  rsum.lo = ocr_f2{0.f, 0.f}; rsum.hi = ocr_f2{0.f, 0.f};
  const int rs_row = tid / CQ, rs_q = tid - rs_row * CQ;  // ROWSUM: this thread adds up output row yb + rs_row, quad rs_q
  const int ntx = (OW + TX - 1) / TX;
  G(0);
  S();
  for (int tx = 0; tx < ntx; ++tx) {
    __syncthreads();  // the region of tile tx (first tile: weights and zero row too) is in LDS
    if (tx + 1 < ntx) G(tx + 1);  // the next tile's pieces travel while this one is computed
    const int x0 = tx * TX + strip * TO;
    F4 acc[R][TO];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int o = 0; o < TO; ++o) { acc[r][o].lo = ocr_f2{0.f, 0.f}; acc[r][o].hi = ocr_f2{0.f, 0.f}; }
    // the patch's input rows top to bottom, one at a time (a fully unrolled walk keeps every row live: 500 registers)
#pragma unroll
    for (int j = 0; j < NROWS; ++j) {
      const int ry = band * R * SH + j;  // region row; rows outside the image: the zero row
      const int rslot = (ry >= ry_lo && ry < ry_hi) ? (ry - ry_lo) * RW : zslot;
      const float* row = s_reg + (size_t)(rslot + strip * TO) * PS + 4 * q;
      float4 in[NIN];
#pragma unroll
      for (int c = 0; c < NIN; ++c) in[c] = *(const float4*)(row + c * PS);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ky = j - r * SH;  // uniform
        if (ky < 0 || ky >= K) continue;
        const float* wrow = wq + ky * K * CC;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 wv = *(const float4*)(wrow + kx * CC);
          const ocr_f2 wlo = {wv.x, wv.y}, whi = {wv.z, wv.w};
#pragma unroll
          for (int o = 0; o < TO; ++o) {
            const float4 v = in[o + kx];
            acc[r][o].lo = __builtin_elementwise_fma(ocr_f2{v.x, v.y}, wlo, acc[r][o].lo);
            acc[r][o].hi = __builtin_elementwise_fma(ocr_f2{v.z, v.w}, whi, acc[r][o].hi);
          }
        }
      }
    }
    const long obase = (opix0 + (long)y0 * OW + x0) * Cs + pcs;
    const long orow = (long)OW * Cs;
    dw_patch_epilogue<R, TO, H16>(acc, ep, pcs, n, Cs, obase, orow, y0, x0, OHn, OW);
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int o = 0; o < TO; ++o)
        if (qok && y0 + r < OHn && x0 + o < OW)
          st4<H16>(a.out, obase + r * orow + (long)o * Cs, make_float4(acc[r][o].lo.x, acc[r][o].lo.y, acc[r][o].hi.x, acc[r][o].hi.y));
    __syncthreads();  // every tap of this tile has been read: the region's space is free
    if constexpr (ROWSUM) {
      float* s_out = s_reg;  // [TY][TX][PS]
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int o = 0; o < TO; ++o)
          *(float4*)(s_out + ((size_t)(band * R + r) * TX + strip * TO + o) * PS + 4 * q) =
              make_float4(acc[r][o].lo.x, acc[r][o].lo.y, acc[r][o].hi.x, acc[r][o].hi.y);
      __syncthreads();
      if (tid < TY * CQ && yb + rs_row < OHn) {
        const int nx = OW - tx * TX < TX ? OW - tx * TX : TX;
        const float* src = s_out + (size_t)rs_row * TX * PS + 4 * rs_q;
        for (int x = 0; x < nx; ++x) {  // s = s + v, x ascending: the pool's first pass
          const float4 v = *(const float4*)(src + x * PS);
          rsum.lo = rsum.lo + ocr_f2{v.x, v.y}; rsum.hi = rsum.hi + ocr_f2{v.z, v.w};
        }
      }
      __syncthreads();  // the exchange tile has been read
    }
    if (tx + 1 < ntx) S();
  }
  if constexpr (ROWSUM) {
    if (tid < TY * CQ && yb + rs_row < OHn && c0 + 4 * rs_q < Cs)
      *(float4*)(a.rowsum + (orow0 + yb + rs_row) * Cs + c0 + 4 * rs_q) = make_float4(rsum.lo.x, rsum.lo.y, rsum.hi.x, rsum.hi.y);
  }
}

namespace {

template <int K, int SH, int R, int NB, int TO, int NS, int HR, bool ROWSUM, bool RAG>
bool launch_dwlds_one(const DwArgs& a, const Epilogue& ep, hipStream_t s, bool query) {
  using G_ = DwLdsGeom<K, SH, R, NB, TO, NS>;
  if (a.H != HR || a.OH > G_::TY) return false;  // the whole image is one band group
  const int rows_real = HR;
  const size_t lds = G_::lds_floats(rows_real) * sizeof(float);
  static LdsAttrMemo attr_state;
  if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)dw_lds_kernel<K, SH, R, NB, TO, NS, HR, ROWSUM, RAG>, (int)lds, attr_state)) return false;
  if (lds > 150 * 1024) return false;
  if (query) return true;
  const long wgs = (long)a.N * ((a.OH + G_::TY - 1) / G_::TY) * ((a.Cs + G_::CC - 1) / G_::CC);
  if (wgs <= 0 || wgs > 0x7fffffffL) return false;
  hipLaunchKernelGGL((dw_lds_kernel<K, SH, R, NB, TO, NS, HR, ROWSUM, RAG>), dim3((unsigned)wgs), dim3(G_::NTHR), lds, s, a, ep, G_::slots(rows_real));
  return true;
}

template <int K, int SH, int R, int NB, int TO, int NS, int HR>
bool launch_dwlds_shape(const DwArgs& a, const Epilogue& ep, hipStream_t s, bool query) {
  const bool rag = a.rout.w != nullptr;
  if (a.rowsum) return rag ? launch_dwlds_one<K, SH, R, NB, TO, NS, HR, true, true>(a, ep, s, query) : launch_dwlds_one<K, SH, R, NB, TO, NS, HR, true, false>(a, ep, s, query);
  return rag ? launch_dwlds_one<K, SH, R, NB, TO, NS, HR, false, true>(a, ep, s, query) : launch_dwlds_one<K, SH, R, NB, TO, NS, HR, false, false>(a, ep, s, query);
}

}  // namespace

// The shapes on this path (everything else stays with dw_conv_kernel): 5x5, pad 2, stride (1 | 2, 1), a uniform batch or a
// ragged batch of LINES, maps of exactly 6 or 3 input rows (the recognizer's ops 26 / 31 / 33 at its 48-pixel line height) -
// the whole image is one band group.  Taller maps (the classifier's 12 and 24 rows, rec op 21) measured no better than
// dw_conv_kernel in this form; the detector's depthwise layers are square maps hundreds of rows high.  query = true only asks.
bool OCR_L(launch_dw_lds)(const DwArgs& a, const Epilogue& ep, hipStream_t s, bool query) {
  OCR_H16_TWIN(a.h16, launch_dw_lds_h16(a, ep, s, query))
  if (a.K != 5 || a.SW != 1 || a.PH != 2 || a.PW != 2 || (a.SH != 1 && a.SH != 2)) return false;
  if (a.rout.h || a.rin.h) return false;  // ragged batch of images: per-sample heights
  for (int i = 0; i < ep.n; ++i) if (ep.st[i].kind == EP_ADDUP) return false;
  if (a.H == 6 && a.SH == 1 && a.OH == 6) return launch_dwlds_shape<5, 1, 3, 2, 4, 4, 6>(a, ep, s, query);   // 6 -> 6 rows: two bands of three, four strips of four columns
  if (a.H == 6 && a.SH == 2 && a.OH == 3) return launch_dwlds_shape<5, 2, 3, 1, 2, 8, 6>(a, ep, s, query);   // 6 -> 3 rows: one band, eight strips of two columns
  if (a.H == 3 && a.SH == 1 && a.OH == 3) return launch_dwlds_shape<5, 1, 3, 1, 2, 8, 3>(a, ep, s, query);   // 3 -> 3 rows
  return false;
}

#ifdef OCR_TU_H16
}  // namespace h16
#endif
}  // namespace ocr
