// Host-visible launch interface of srv_kernels.hip: the kernel family of the "server" networks (BASELINE configs[4]:
// ResNet50-vd DB detector + SVTR-large recognizer, precision fp16; plans/srv_*.plan - hand-written, NOT reference artifacts).
//
// Data layout: activations are plain NHWC tensors of element type T - f16 in the product's mode
// (precision "fp16", the mode the config names), f32 in the PARITY TWIN (same kernels instantiated on float: every contraction
// is then the oracle's ascending-k fma chain on v_mfma_f32_32x32x2_f32 and results equal the oracle's bit for bit, which pins
// indexing, padding, strides and fusions exactly; the f16 build differs from it by rounding only).  Channels are stored padded
// to a multiple of 8 (pads hold zeros); one 16-byte GRANULE = 8 halfs / 4 floats is the unit of every load.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace ocr {
namespace srv {

enum : int { SACT_NONE = 0, SACT_RELU = 1, SACT_GELU = 2, SACT_HSWISH = 3, SACT_SIGMOID = 4 };

// Implicit-GEMM convolution / linear / 2x2 transposed conv:  Y[m][n] = epilogue( sum_k W[n][k] * X[m][k] ),
// m = output pixel (N*OH*OW), n = output channel, k = ((ky*KW)+kx)*Cin + ci ascending.  Both operand tiles are staged through
// LDS by LDS-DMA (buffer_load ... lds, 16 bytes per lane, three stages in flight across raw barriers), XOR-swizzled so that
// the fragment reads (ds_read_b128) are conflict-free; the matrix instruction is v_mfma_f32_32x32x16_f16 (f32 twin:
// v_mfma_f32_32x32x2_f32) with the WEIGHTS as the A operand, so a lane owns a pixel and the accumulator registers run over
// output channels; the epilogue goes through an f32 LDS tile and leaves as whole 16-byte chunks of pixel rows.
struct GemmArgs {
  const void* x = nullptr;   // [N][H][W][Cin] T
  unsigned long long x_bytes = 0;
  const void* w = nullptr;   // weight image [nkt][Npad][8 granules, XOR-swizzled by row] T (srv_net.hip: weight_image)
  unsigned long long w_bytes = 0;
  void* y = nullptr;         // [N][OH][OW][Cs_out] T (out_f32: float)
  unsigned long long y_bytes = 0;
  long M = 0;
  int K = 0, nkt = 0;        // K = KH*KW*Cin (Cin = stored channels), K tiles of 8 granules
  int Npad = 0;              // rows of the weight image: GEMM rows padded to a multiple of 256
  int Ncols = 0;             // GEMM rows that exist (conv / linear: Cout; deconv: 4 * CoutD)
  int Cs_out = 0;            // stored channels per output pixel
  int N = 0, H = 0, W = 0, Cin = 0, OH = 0, OW = 0, KH = 1, KW = 1, SH = 1, SW = 1, PH = 0, PW = 0;
  int cin_shift = -1;        // -1: a K tile lies inside one tap (Cin a multiple of the tile); else a tile spans taps and a lane finds
                             // its granule's (tap, channel) by shift (>= 0: log2 Cin) or by division (-2)
  int deconv = 0;            // 2x2 stride-2 transposed conv: GEMM row = (dy*2+dx)*CoutD + co, scattered to pixel (2y+dy, 2x+dx)
  int CoutD = 0;
  int x1 = 0;                // 1x1 stride-1 unpadded conv / linear with K a multiple of the K tile: X rows are contiguous
  // epilogue, in this order (each optional): + bias[n] ; * scale[n] then + shift[n] (batch norm) ; + res[m][n] (res_up = 1) or
  // + res[pixel (y/2, x/2)][n] (res_up = 2: the FPN's nearest-upsampled add) ; activation
  const float* bias = nullptr;
  const float* scale = nullptr;
  const float* shift = nullptr;
  const void* res = nullptr;
  unsigned long long res_bytes = 0;
  int res_up = 0;
  int act = SACT_NONE;
  int out_f32 = 0;
  // CTC head (f32-output launches): instead of writing the logits, every workgroup leaves per output row ONE partial of its column
  // tile - (max logit, sum of exp(logit - max), index of the first maximum) as 4 floats at ctc_part[(row * ctc_slots + n0 / 64) * 4] -
  // which launch_ctc_reduce folds into the row's arg max and its softmax probability: the logits (2.2 GB per 1024 lines) never exist
  // halo form only: the input is the CONCATENATION of cat_n 64-channel tensors, source j nearest-upsampled by 2^cat_sh[j] (the DB
  // neck's `concat up=8,4,2,1` in front of the head conv): channel tile j of the K order (channel tile, tap) IS source j, so the
  // patch of tile j is fetched from source j at (y >> sh, x >> sh) - the 256-channel concatenation is never written
  const void* cat_x[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned long long cat_bytes[4] = {0, 0, 0, 0};
  int cat_sh[4] = {0, 0, 0, 0};
  int cat_n = 0;
  float* ctc_part = nullptr;
  int ctc_slots = 0;
  int group_m = 0;           // tile order: 0 = the column tiles of a pixel tile are neighbours (the whole weight slab streams past each pixel tile);
                             // G > 0 = pixel tiles in groups of G, a column tile's G pixel tiles neighbours, then the next column tile - the slab
                             // streams once per GROUP while the group's G pixel panels stay in L2 (slabs that do not fit L2: the CTC head's 5 MB)
  int korder = 0;            // K order of a 3x3 stride-1 conv's weight image: 0 = (tap, channel) - the oracle's ascending chain, every f32 launch;
                             // 1 = (64-channel tile, tap, channel in tile) - the f16 build: a channel tile's nine taps are consecutive K tiles,
                             // so the halo form (srv_conv3_kernel) streams ONE input patch per channel tile; both forms accumulate in this order
};
int gemm_num_configs();
int gemm_config_bn(int cfg);  // column-tile width of a configuration (the CTC partials' slot step is bn / 64)
const char* gemm_config_name(int cfg);
// can tile configuration `cfg` run this problem? (shape divisibility, LDS attribute) - asked at bind time
bool gemm_config_ok(const GemmArgs& a, bool half, int cfg);
bool launch_gemm(const GemmArgs& a, bool half, int cfg, hipStream_t s, std::string& err);

// SVTR's MLP in one launch (f16 build): y = x + fc2(gelu(fc1(x) + b1)) + b2, x / y [M][C] f16, the weight images of the two linears
// as launch_gemm takes them; C in {192, 256, 512}.  query: instantiation / LDS attribute check only.  Bit-identical to the two
// launches it replaces (srv_mlp.h).
bool launch_mlp(const void* x, unsigned long long x_bytes, const void* w1, unsigned long long w1_bytes, int w1_npad, const void* w2,
                unsigned long long w2_bytes, int w2_npad, const float* b1, const float* b2, void* y, long M, int C, hipStream_t s, bool query,
                std::string& err);
// the same with the LayerNorm of its input absorbed (srv_mlp.h MlpArgs::ln_*): x = the raw sum, w1 = the image of diag(gamma) W1
struct MlpLn { const float *g = nullptr, *b = nullptr, *s = nullptr, *c = nullptr; float eps = 0.f; };
bool launch_mlp_ln(const void* x, unsigned long long x_bytes, const void* w1, unsigned long long w1_bytes, int w1_npad, const void* w2,
                   unsigned long long w2_bytes, int w2_npad, const MlpLn& ln, const float* b2, void* y, long M, int C, hipStream_t s, bool query,
                   std::string& err);

// f32 [N][H][W][3] (the normalised image, what the pre-processing kernels write) -> T [N][H][W][8], channels 3..7 zero
void launch_pack_input(const float* x, void* y, long pixels, bool half, hipStream_t s);
// max / average pool (window kh x kw, stride, padding; positions outside the image take no part)
void launch_pool(const void* x, void* y, int N, int H, int W, int Cs, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw,
                 bool is_max, bool half, hipStream_t s);
// channel concat of up to 4 tensors of cs channels each, source j read with nearest upsampling by up[j]
void launch_concat_up(const void* const src[4], const int up[4], int nsrc, int cs, void* y, int N, int OH, int OW, bool half, hipStream_t s);
// y = x + pos[(h*W + w)][c] (the position embedding, f32 parameter [H*W][Cs])
void launch_addpos(const void* x, const float* pos, void* y, long pixels, int hw, int Cs, bool half, hipStream_t s);
// layer norm over the channels of every pixel (C = Cs: no pad channels in these tensors)
void launch_layernorm(const void* x, const float* g, const float* b, void* y, long rows, int C, float eps, bool half, hipStream_t s);
// SVTR mixing: qkv [N][T][3*D] (q | k | v, head-major inside each) -> out [N][T][D]; token grid gh x gw, local window lh x lw
// (lh = 0: global).  hd must be 32.
bool launch_attention(const void* qkv, void* out, int N, int T, int heads, int hd, float scale, int gh, int gw, int lh, int lw, bool half,
                      hipStream_t s, std::string& err);
// the DB head's last layer: 2x2 stride-2 transposed conv Cin -> 1 + bias + sigmoid, output the f32 probability map [N][2H][2W]
void launch_deconv_to_map(const void* x, const float* w4 /* [4 taps][Cs] */, float bias, float* prob, int N, int H, int W, int Cs, bool half,
                          hipStream_t s);
// CTC head's tail: per row of f32 logits [rows][ld] (C valid): arg max (first maximum) and its softmax probability
// the DB head's two transposed convs (64 -> 64 + bias + ReLU, 64 -> 1 + bias + sigmoid) in one launch, f16 build: x [N][H][W][64] f16,
// w1img = the first one's weight image (256 rows = tap * 64 + channel, one K tile), b1 [256], w4 [4 taps][64] -> prob [N][4H][4W] f32
bool launch_head_tail(const void* x, const void* w1img, const float* b1, const float* w4, float bias2, float* prob, int N, int H, int W, hipStream_t s,
                      bool query, std::string& err);
void launch_argmax_softmax(const float* logits, long rows, int C, int ld, int* amax, float* pmax, hipStream_t s);
// folds the partials of a CTC-mode head launch (GemmArgs::ctc_part): slots 0, step, 2 step, .. < slots of every row
void launch_ctc_reduce(const float* part, long rows, int slots, int step, int* amax, float* pmax, hipStream_t s);
void launch_argmax_softmax(const float* logits, long rows, int C, int ld, int* amax, float* pmax, bool half, hipStream_t s);  // half: the one-pass form
// copies a T tensor to f32 dropping the pad channels (parity taps)
void launch_to_f32(const void* x, float* y, long pixels, int Cs, int C, bool half, hipStream_t s);

}  // namespace srv
}  // namespace ocr
