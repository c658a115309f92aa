// Host-visible launch interface of kernels_net.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "hip_guard.h"
#include "lds_attr.h"
#include "ocr_common.h"
#include "rt_options.h"

namespace ocr {

enum : int { OUT_C8I = 0, OUT_PLAIN = 1, OUT_DECONV = 2, OUT_HEAD = 3 };

// Ragged batch (the recognizer, /root/reference/src/ocr_rec.cpp:47-72: every batch of 16 lines has its own tensor
// width): the N samples of ONE launch are text lines of one height H and their own widths.  A tensor at a "width
// level" (input, /2 after the stem, /4, /8 = CTC steps) stores line n as a dense [H][w[n]][Cs] block that starts at
// pixel H * cw[n]; cw = exclusive prefix sums of w, N + 1 entries.  Pointwise ops (1x1 conv, linear, layer norm, the
// CTC head) see one row axis of H * cw[N] pixels and need nothing else; spatial ops take the line's own width for
// their zero padding, SE pools and attention their own counts.  Every line's arithmetic is what a launch of that
// line alone would compute: results do not depend on what else is in the batch.
//
// The same for the detector on MIXED IMAGE SIZES (round 3: `h != null`): sample n is an image of its own height and width.
// Detector inputs are multiples of 32 (ResizeImgType0) and every level of its network is an exact power-of-two scaling of
// the input, so ONE set of tables at input resolution serves all levels: at a level `shift` image n is
// (h[n] >> shift) x (w[n] >> shift) and starts at pixel cw[n] >> 2*shift; cw = prefix sums of h*w, ch = prefix sums of h
// (rows before the image: the row-sum buffers of the SE pools are [rows][Cs]).  w and h carry one entry past N (the tile
// walks step onto "image N" after their last unit).
struct RagLevel {
  const int* w = nullptr;   // [N]   lines: widths at this level.  images: widths at input resolution.  null = uniform batch
  const int* cw = nullptr;  // [N+1] lines: prefix sums of w.      images: prefix sums of h * w at input resolution
  const int* h = nullptr;   // images only: heights at input resolution (null: lines of one height, the launch's H)
  const int* ch = nullptr;  // images only: [N+1] prefix sums of h at input resolution
  int shift = 0;            // images only: this level = input resolution >> shift
};

struct ConvArgs {
  const float* in;   // [N,H,W,Cs_in] C8I
  float* out;
  const float* wfrag;  // [tap][C8][NTtot][64 lanes][4], NTtot padded to a multiple of the launch's NT
  const float* zeros;  // >= Cs_in zero floats (source of padded taps / rows beyond M)
  long M;              // N*OH*OW GEMM rows
  int N, H, W, Cs_in, C8;
  int OH, OW, Cs_out;
  int Cout;        // logical channels per quadrant (columns >= Cout inside CoutPadded store 0)
  int CoutPadded;  // deconv: column = q*CoutPadded + ch
  int ColsStore;   // GEMM columns that are stored
  int NTtot;       // 32-wide column tiles in the fragment image
  int KH, KW, PH, PW;
  int out_mode;
  int need_nyx;
  // OUT_HEAD (linear + softmax fused, nothing of the logits is stored): per (row, 128-column group)
  // the group's max logit, sum of exp(logit - max) and arg max; [M][NTtot/NT] each
  float* head_max;
  float* head_sum;
  int* head_idx;
  // SE gate folded into the A operand of a 1x1 conv (the `ew mulc` pass it replaces computed x * gate[n][c],
  // one rounding, and this conv then read the product): [N][Cs_in] physical channel order, null = none
  const float* gate;
  int gate_hw;  // pixels per image (row m belongs to image m / gate_hw)
  // ragged batch (rin.w != null): N samples - lines of in.H rows each and their own widths, or images of their own sizes
  RagLevel rin, rout;
  // 3x3 tile kernels on a ragged batch of images: [N+1] prefix sums of the images' 8x16-pixel tile counts
  const int* rtiles = nullptr;
  int rtiles_total = 0;
  // precision "fp16": in / out (C8I) and the epilogue's tensor operands are f16 tensors, wfrag is the f16 fragment image
  // ([...][64 lanes][4 halfs]) and the products run as v_mfma_f32_32x32x8_f16 with f32 accumulation; plain outputs, head
  // partials, gates and parameter vectors stay f32.  0: the f32 contract.  (The launchers hand such a call to their
  // `_h16` twin, below.)
  int half = 0;
  // precision "fp16", single-tap convs with an even number of octets: the fragment image for v_mfma_f32_32x32x16_f16
  // ([octet pair][column tile][64 lanes][8 halfs]: lane (p, h) holds the eight channels of octet 2s + h), or null
  const float* wfrag_x16 = nullptr;
  // a concat folded into the fill of the 3x3 LDS-tile kernels (net.hip; the DB neck's `concat -> conv 3x3 96 -> 24`): `in` is
  // null and the tile's physical channels cat_cs * j .. cat_cs * (j + 1) - 1 come from source j, a [N][H / up][W / up][cat_cs]
  // tensor read with nearest upsampling by cat_up[j] (a power of two) - what concat_kernel would have written
  const float* cat_src[4] = {nullptr, nullptr, nullptr, nullptr};
  int cat_up[4] = {1, 1, 1, 1};
  int cat_n = 0, cat_cs = 0;
};
// false: the combination (gated input / multi-tap conv with a plain or deconv output) is not instantiated
bool launch_conv_mfma(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s);
// 1x1 conv with few input channels whose output is only wanted as the global average pool's row sums (the detector's
// RSE blocks, net.hip): part[(n*H + y)][Cs_out] = sum over x, left to right, of the conv's outputs - what gap_rows
// would leave after reading the materialised tensor.  w: [Cs_in (logical k, zero rows past Cin)][Cs_out physical].
struct ConvRowsumArgs {
  const float* in;
  const float* w;
  float* part;
  long rows;  // N * H (ragged batch: every sample's rows)
  int W, Cin, Cs_in, Cs_out;
  int N, H;      // ragged batch: samples, uniform height (lines mode)
  RagLevel rag;  // ragged batch: the level of the conv's input = output
  int h16 = 0;   // precision "fp16": `in` is an f16 tensor (the row sums stay f32)
};
bool launch_conv_rowsum(const ConvRowsumArgs& a, hipStream_t s);
// the same with two 32-pixel tiles per wave (big single-tap convs with a C8I output, nt = 3 | 4); false: not this shape
bool launch_conv_mfma_mt2(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s);
// LDS-staged variant of the same GEMM (default when K = taps*Cs_in >= 64)
void launch_conv_lds(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s);
// 3x3 s1 p1 conv with the input tile resident in LDS; returns false when the shape is not on that path
// (query = true here and below: shape checks and the device's LDS attribute only - asked at bind time, nothing is launched)
bool launch_conv3x3_tile(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s, bool query = false);
// 3x3 s1 p1 conv 96 -> 24 channels on 4x4x1 matrix blocks (no zero columns); wimg from conv3x3_c24_image
bool launch_conv3x3_c24(const ConvArgs& a, const Epilogue& ep, const float* wimg, hipStream_t s, bool query = false);
std::vector<float> conv3x3_c24_image(const float* w, int co, int ci);
// column tiles per wave for a GEMM with `tiles` 32-wide column tiles
inline int conv_nt_for(int tiles) {
  return tiles <= 4 ? tiles : (tiles % 4 == 0 ? 4 : (tiles % 3 == 0 ? 3 : 4));
}

struct StemArgs {
  const float* in;  // [N,H,W,3] plain
  float* out;       // [N,OH,OW,Cs_out] C8I
  const float* w;   // [KH*KW*3][Cs_out] physical order
  long M;
  int N, H, W, OH, OW, Cs_out, KH, KW, SH, SW, PH, PW;
  RagLevel rin, rout;  // ragged batch: W / OW per line
  int h16 = 0;         // precision "fp16": `out` is an f16 tensor (the image stays f32)
};
void launch_stem(const StemArgs& a, const Epilogue& ep, hipStream_t s);

struct DwArgs {
  const float* in;
  float* out;
  const float* w;  // [K*K][Cs] physical order
  long M;
  int N, H, W, OH, OW, Cs, K, SH, SW, PH, PW;
  // non-null: the conv also leaves the global average pool's first pass here - for every output row its sum over x,
  // left to right, [N*OH][Cs] (what gap_rows_kernel computes from the tensor in a second full read); a thread then
  // owns whole rows (all strips of its band, in x order) instead of one patch
  float* rowsum = nullptr;
  RagLevel rin, rout;          // ragged batch: sizes per sample
  const int* rwork = nullptr;  // ragged: [N+1] prefix sums of the samples' work items - bands (rowsum) or bands x strips of the launcher's patch (dw_patch_to / dw_patch_r)
  int rwork_total = 0;         // rwork[N] (host copy: the launcher sizes the grid with it)
  int h16 = 0;                 // precision "fp16": in / out (and an ADDT operand) are f16 tensors; rowsum stays f32
};
void launch_dw(const DwArgs& a, const Epilogue& ep, hipStream_t s);
int dw_patch_to(int OW, int SW, int OH, int K);  // output pixels per thread along x the launcher will pick
// the same conv with the haloed input region staged through LDS (kernels_dwlds.hip: 5x5, stride (1 | 2, 1), uniform batch or
// ragged batch of lines, low maps); false: the shape is not on that path (launch_dw takes it).  query = true only asks.
bool launch_dw_lds(const DwArgs& a, const Epilogue& ep, hipStream_t s, bool query = false);
// the pool's second pass alone (column-sequential sums of the row sums, / count): after a depthwise conv with `rowsum`
void launch_gap_cols(const float* part, float* out, int N, int H, int W, int Cs, hipStream_t s, RagLevel rag = RagLevel());

// Fused depthwise conv (+ its epilogue) -> 1x1 conv (+ its epilogue), kernels_dwpw.hip.  Both epilogues are the
// PPLCNetV3 "learnable affine block" chain of the plans AFTER the loader's fold (net.hip, fold_lab), fixed at compile time
// (a generic stage interpreter made the kernel VALU-bound on its own bookkeeping):
//   depthwise half:  y = x + bias[c];  u = y * clamp(y + 3, 0, 6)                      (its scale / shift live in the 1x1 conv)
//   1x1 half:        y = x + bias[c];  u = y * clamp(y + 3, 0, 6);  out = fmaf(u, s6, a1)
struct LabEp {
  const float* bias;  // [Cs] physical channel order (the folded bias)
  float s6, a1;       // sfma: out = fmaf(u, s6, a1)
  int sfma;           // 1: the fma follows the hard-swish product; 0: the chain ends with u (an absorbed depthwise conv)
};
struct DwPwArgs {
  ConvArgs c;          // the 1x1 conv as launch_conv_mfma would get it (c.in unused: that tensor never exists; c.zeros used)
  const float* dw_in;  // [N,H,W,Cs] C8I, Cs = c.Cs_in
  const float* dw_w;   // [K*K][Cs] physical order
  // the same weights and the depthwise bias chunk by chunk, [Cs / CK][K*K taps | bias][CK] for CK = 16 and 32: what one LDS-DMA
  // burst of dwpw2_kernel fetches per item (net.hip "dwq16:" / "dwq32:"; the latter only where Cs % 32 == 0); null: the first form only
  const float* dw_wq16 = nullptr;
  const float* dw_wq32 = nullptr;
  LabEp dw_ep, pw_ep;
  int H, W, K, SH, SW, PH, PW;
  int tiles_x, tiles_y;  // filled by the launcher
  unsigned nunits;       // work units = pixel tiles x column blocks (filled by the launcher)
  unsigned upw;          // units per workgroup (filled by the launcher)
  // ragged batch: sample sizes in / out and [N+1] prefix sums of the samples' tile counts ceil(OH_n / TH) * ceil(OW_n / 16)
  // for the instance's TH (dwpw_tile_rows)
  RagLevel rin, rout;
  const int* rtiles = nullptr;
  int rtiles_total = 0;  // rtiles[N] (host copy: the launcher sizes the grid with it)
};
// host: is this stage list one of the two chains above?  fills `out`
bool lab_from_epilogue(const Epilogue& ep, LabEp& out);
// false: the shape is not instantiated (the caller launches the unfused pair).  query = true only asks.
bool launch_dwpw(const DwPwArgs& a, hipStream_t s, bool query = false);

// ---- expand 1x1 conv (BN, hard-swish) -> depthwise 5x5 (BN, hard-swish) of one small map per workgroup (kernels_xdw.hip):
// the expanded tensor exists only in LDS.  f32 contract, uniform batches.
struct XdwArgs {
  const float* x;       // [N, Hin, W, Cs_in] C8I: the 1x1 conv's input
  const float* wfrag;   // the 1x1 conv's fragment image ([Cs_in / 8][NTtot][64][4])
  const float* e_sc;    // the 1x1 conv's BN scale / shift, [Cs_e] physical order
  const float* e_sh;
  const float* dw_w;    // [K*K][Cs_e] physical order
  const float* d_sc;    // the depthwise conv's BN scale / shift
  const float* d_sh;
  float* out;           // [N, Hout, W, Cs_e] C8I: the depthwise conv's output
  float* part;          // null, or the pool's row sums [N * Hout][Cs_e] (as DwArgs::rowsum)
  int N, Hin, Hout, W, Cs_in, Cs_e, NTtot, K, SH, SW, PH, PW;
};
// false: the shape is not instantiated (the caller launches the two ops).  query = true only asks.
bool launch_xdw(const XdwArgs& a, hipStream_t s, bool query = false);
// rows of the pixel tile (16 columns wide) of the instance that takes this shape; 0: the shape is not on the fused path
int dwpw_tile_rows(const DwPwArgs& a);
int dw_patch_r(int OH, int K);  // output rows per thread the depthwise launcher will pick (ragged batch: OH = the lowest sample)

// rag (ragged batch of N lines): the per-image stages (channel gate) find their line from the row index
// h16 (here and below): precision "fp16" - the C8I tensors among the arguments are stored as f16
void launch_ew(const float* in, float* out, long M, int H, int W, int Cs, const Epilogue& ep, hipStream_t s, int N = 0,
               RagLevel rag = RagLevel(), bool h16 = false);
void launch_gap(const float* in, float* part, float* out, int N, int H, int W, int Cs, hipStream_t s, RagLevel rag = RagLevel(),
                long rows = 0, bool h16 = false);  // rows: total rows of the batch (0: N * H)

struct SeArgs {
  const float* in;  // [N][Cs]
  float* out;       // [N][Cs]
  const float *w1, *b1, *w2, *b2;  // logical: w1 [R][C], w2 [C][R]
  int C, Cs, R;
  float slope, offset;
};
void launch_sefc(const SeArgs& a, int N, hipStream_t s);

struct ConcatArgs {
  const float* src[4];
  float* out;
  long M;
  int H, W, Cs, nsrc;
  int coff[4], scs[4], up[4];
  RagLevel rout;  // ragged batch of images: the output's level; source j lives log2(up[j]) levels coarser
  int N = 0;
  int h16 = 0;
};
void launch_concat(const ConcatArgs& a, hipStream_t s);

struct PoolArgs {
  const float* in;
  float* out;
  long M;
  int N, H, W, OH, OW, Cs, KH, KW, SH, SW, is_max;
  RagLevel rin, rout;  // ragged batch: W / OW per line
  int h16 = 0;
};
void launch_pool(const PoolArgs& a, hipStream_t s);

void launch_ln(const float* in, float* out, long rows, int C, int Cs, float eps, const float* g, const float* b,
               hipStream_t s, bool h16 = false);
// rag (ragged batch): line n is a sequence of rag.w[n] tokens starting at row rag.cw[n]; T = the longest line
void launch_attn(const float* qkv, float* out, int N, int T, int heads, int hd, int Cs_in, int Cs_out, float scale,
                 hipStream_t s, RagLevel rag = RagLevel(), bool h16 = false);
// does the ragged form of the attention kernel take sequences of this length? (its working set must fit LDS)
bool attn_ragged_fits(int T);
// second half of the fused head: rows x groups partials -> arg max / max probability per row
void launch_head_combine(const float* hmax, const float* hsum, const int* hidx, long rows, int groups, int* amax, float* pmax,
                         hipStream_t s);
void launch_softmax_argmax(const float* logits, float* probs, int* amax, float* pmax, long rows, int C, hipStream_t s);

struct DetTailArgs {
  const float* in;  // [N,H,W,Cs] C8I
  float* prob;      // [N,2H,2W]
  uint8_t* bitmap;  // [N,2H,2W] {0,1} or null
  const float* w;   // [C][4] (dy*2+dx)
  long M;
  int N, H, W, C, Cs;
  float bias;
  int ithresh;  // floor(det_db_thresh*255): bit = trunc(p*255) > ithresh
  int h16 = 0;  // precision "fp16": `in` is an f16 tensor (the map stays f32)
};
void launch_det_tail(const DetTailArgs& a, hipStream_t s);

// The DB head's two transposed convs as ONE kernel: deconv 2x2 s2 (C -> C) + bias + BN + relu, then the tail above
// (deconv 2x2 s2 C -> 1 + bias + sigmoid + 8-bit threshold).  A thread owns one input pixel = a 4x4 block of the
// probability map; the C-channel map between the two (64 x 480 x 480 x 24 floats at configs[1]: 1.4 GB written and
// read back) never exists.  Same fma chains as the matrix-core deconv followed by det_tail_kernel: k ascending from 0.
struct DbHeadArgs {
  const float* in;    // [N,H,W,Cs] C8I
  float* prob;        // [N,4H,4W]
  uint8_t* bitmap;    // [N,4H,4W] {0,1} or null
  const float* w1;    // [C k][4 q][C c] logical channels, q = dy*2+dx
  const float* wfrag = nullptr;  // or the first stage as a matrix-core fragment image with the head's column order ("dbhf:")
  const float* bias1; // [Cs] physical order
  const float* bn_s;  // [Cs] physical order: scale
  const float* bn_t;  // [Cs] physical order: shift
  const float* w2;    // [C][4]
  long M;             // N*H*W input pixels
  int N, H, W, Cs;
  float bias2;
  int ithresh;
  RagLevel rin;       // ragged batch of images: the input's level (the map is two levels finer)
  int h16 = 0;        // precision "fp16": `in` is an f16 tensor (the map stays f32)
};
bool launch_db_head(const DbHeadArgs& a, int C, hipStream_t s);  // false: C is not on this path (24 only)

void launch_c8i_to_plain(const float* in, float* out, long M, int C, int Cs, hipStream_t s, bool h16 = false);
void launch_probe(const float* a, const float* b, float* out, int n, hipStream_t s);

// precision "fp16": kernels_net.hip and kernels_dwpw.hip are compiled a second time with -DOCR_TU_H16 (f16 tensor storage,
// f16 matrix instructions, f32 accumulation and epilogues); that build's kernels live in this inline namespace and its
// launchers carry the suffix.  Not called directly: the launchers above forward a call whose f16 flag is set.
inline namespace h16 {
bool launch_conv_mfma_h16(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s);
bool launch_conv_mfma_mt2_h16(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s);
bool launch_conv3x3_tile_h16(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s, bool query);
bool launch_conv_rowsum_h16(const ConvRowsumArgs& a, hipStream_t s);
void launch_stem_h16(const StemArgs& a, const Epilogue& ep, hipStream_t s);
void launch_dw_h16(const DwArgs& a, const Epilogue& ep, hipStream_t s);
bool launch_dwpw_h16(const DwPwArgs& a, hipStream_t s, bool query);
bool launch_dw_lds_h16(const DwArgs& a, const Epilogue& ep, hipStream_t s, bool query);
void launch_ew_h16(const float* in, float* out, long M, int H, int W, int Cs, const Epilogue& ep, hipStream_t s, int N, RagLevel rag, bool h16);
void launch_gap_h16(const float* in, float* part, float* out, int N, int H, int W, int Cs, hipStream_t s, RagLevel rag, long rows, bool h16);
void launch_concat_h16(const ConcatArgs& a, hipStream_t s);
void launch_pool_h16(const PoolArgs& a, hipStream_t s);
void launch_ln_h16(const float* in, float* out, long rows, int C, int Cs, float eps, const float* g, const float* b, hipStream_t s, bool h16);
void launch_attn_h16(const float* qkv, float* out, int N, int T, int heads, int hd, int Cs_in, int Cs_out, float scale, hipStream_t s,
                     RagLevel rag, bool h16);
void launch_det_tail_h16(const DetTailArgs& a, hipStream_t s);
bool launch_db_head_h16(const DbHeadArgs& a, int C, hipStream_t s);
void launch_c8i_to_plain_h16(const float* in, float* out, long M, int C, int Cs, hipStream_t s, bool h16);
}  // namespace h16

}  // namespace ocr
