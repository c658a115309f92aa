// Kernels of the "server" networks (BASELINE configs[4]: ResNet50-vd DB detector + SVTR-large recognizer, precision fp16;
// plans/srv_*.plan are hand-written, NOT reference artifacts - tools/make_server_plans.py).  gfx950 only.
//
// What the reference would run here is TensorRT fp16 behind `precision` (/root/reference/src/ocr_det.cpp:50-57): dense
// convolutions, linears and attention - matrix-core work, unlike the depthwise-dominated mobile graphs of kernels_net.hip.
//   * srv_gemm_kernel: ONE implicit-GEMM family for 1x1 / 3x3 / strided convs, linears and 2x2 transposed convs on
//     v_mfma_f32_32x32x16_f16, both operand tiles through LDS by LDS-DMA (three stages in flight across raw barriers, counted
//     vmcnt), XOR-swizzled images, conv + bias + batch-norm + residual / upsampled add + activation in the epilogue;
//   * srv_attn_kernel: SVTR's local / global mixing, K and V of a (line, head) resident in LDS, S^T = K Q^T so that a lane owns a
//     query (row max / sum without LDS), the probabilities handed to the P V product as the accumulator-as-operand B fragment,
//     V through ds_read_b64_tr_b16; key tiles outside the 7 x 11 window are skipped;
//   * the rest (pack, pools, concat, layer norm, position add, map head, arg-max softmax) streams.
// Every kernel is a template on the element type: f16 = the product's mode, float = the PARITY TWIN whose arithmetic is the
// oracle's operation for operation (srv_kernels.h).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <string>

#include "lds_attr.h"
#include "ocr_common.h"
#include "srv_kernels.h"

namespace ocr {
namespace srv {

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16x __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <typename T> struct ET;
template <> struct ET<_Float16> { static constexpr int KG = 8, BK = 64; };
template <> struct ET<float> { static constexpr int KG = 4, BK = 32; };

#define SRV_OOB 0xfffffff0u  // beyond every num_records: the lane receives zeros

// one LDS-DMA instruction: lane l's 16 bytes at (rsrc base + soff + voff) -> LDS byte address lds + 16 l (lds wave-uniform).
// Invisible to the compiler's s_waitcnt bookkeeping: the waits are counted by hand (srv_gemm_kernel).
__device__ __forceinline__ void srv_dma16(unsigned lds, unsigned voff, v4u rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void srv_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ unsigned srv_xcd(unsigned bid, unsigned nblk) {  // blocks that share an XCD get consecutive tiles
  const unsigned q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return x * q + (x < r ? x : r) + i;
}

// erf of the exact GELU: the oracle's operations in the oracle's order (the ocr_erff of the oracle's network restatement)
__device__ __forceinline__ float srv_erff(float x) {
  const float ax = fabsf(x);
  const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p = p * t;
  const float e = ocr_expf(-(ax * ax));
  const float r = fmaf(-p, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float srv_act(int act, float y) {
  switch (act) {
    case SACT_RELU: return fmaxf(y, 0.0f);
    case SACT_GELU: { const float hx = 0.5f * y; const float z = y * 0.70710678118654752f; const float e1 = 1.0f + srv_erff(z); return hx * e1; }
    case SACT_HSWISH: { const float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f); const float u = y * t; return u / 6.0f; }
    case SACT_SIGMOID: { const float e = ocr_expf(-y); const float d = 1.0f + e; return 1.0f / d; }
    default: return y;
  }
}

// the f16 build's activations: results are rounded to 11 bits anyway - hardware exp2 / rcp instead of the contract's exp and the
// IEEE quotients (GELU: the same Abramowitz-Stegun erf, ~14 instructions instead of ~45; it sits in the epilogue of SVTR's
// widest linears and inside the fused MLP, 32 values per lane per 128 hidden units)
__device__ __forceinline__ float srv_act_h(int act, float y) {
  switch (act) {
    case SACT_RELU: return fmaxf(y, 0.0f);
    case SACT_GELU: {
      // x * sigmoid(2 sqrt(2 / pi) (x + 0.044715 x^3)) - the tanh form: within 4.7e-4 of the erf form everywhere, below the f16
      // rounding of the values it produces (2^-11 relative); seven instructions per value instead of seventeen.  The hidden
      // units of SVTR's MLP pass through here M x 4 C times: at the erf form's cost the activation alone was 3500 cycles per
      // 128-unit chunk per wave against 1536 cycles of matrix instructions (stamps: tools/micro/srv_mlp_probe)
      const float x2 = y * y;
      const float w = y * fmaf(x2, -0.10294324f, -2.3022082f);   // -(1.5957691 + 0.0713548 x^2) x log2(e)
      return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(w));
    }
    case SACT_HSWISH: return y * fminf(fmaxf(y + 3.0f, 0.0f), 6.0f) * 0.16666666666666667f;
    case SACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-y * 1.44269504088896341f));
    default: return y;
  }
}

// GELU of TWO values on the packed-f32 instructions (v_pk_mul_f32 / v_pk_fma_f32: one instruction, two lanes' worth), no
// transcendental (v_exp / v_rcp issue at a quarter of the rate: the tanh form above is 5 + 2 x 4 = 13 issue slots per value, this is 7):
//   gelu(x) = x / 2 + E(|x|),  E(t) = t (Phi(t) - 1/2)  - even, smooth, and t / 2 from where Phi is 1 -
//   E(t) ~ t^2 P(t^2) for t = min(|x|, 4) (degree-6 P, minimax fit: tools/micro/gelu_fit.py), + (|x| - t) / 2 beyond
// max |error| 1.9e-4 over the whole line (the tanh form: 4.7e-4), below the f16 rounding of what it produces from |x| ~ 0.4 up.
// EIGHT values at a time, the four pairs' Horner chains advanced in lock step: a packed-f32 instruction that reads the result of
// the one before it costs a wait state (the compiler pads with s_nop 0), and pair after pair the chain of eight was nothing else.
__device__ __forceinline__ void srv_gelu8(float (&v)[8]) {
  auto k2 = [](float c) { f2v w = {c, c}; return w; };
  f2v y[4], t[4], tail[4], t2[4], q[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    y[p] = f2v{v[2 * p], v[2 * p + 1]};
    t[p][0] = fminf(fabsf(y[p][0]), 4.0f);
    t[p][1] = fminf(fabsf(y[p][1]), 4.0f);
    tail[p][0] = fabsf(y[p][0]) - t[p][0];
    tail[p][1] = fabsf(y[p][1]) - t[p][1];
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) t2[p] = t[p] * t[p];
#pragma unroll
  for (int p = 0; p < 4; ++p) q[p] = __builtin_elementwise_fma(k2(2.27814575e-08f), t2[p], k2(-1.59860303e-06f));
  constexpr float CK[5] = {4.79555300e-05f, -8.14015556e-04f, 8.77238884e-03f, -6.45731141e-02f, 3.97883359e-01f};
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int p = 0; p < 4; ++p) q[p] = __builtin_elementwise_fma(q[p], t2[p], k2(CK[k]));
#pragma unroll
  for (int p = 0; p < 4; ++p) y[p] = y[p] * k2(0.5f);
#pragma unroll
  for (int p = 0; p < 4; ++p) y[p] = __builtin_elementwise_fma(t2[p], q[p], y[p]);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    y[p] = __builtin_elementwise_fma(tail[p], k2(0.5f), y[p]);
    v[2 * p] = y[p][0];
    v[2 * p + 1] = y[p][1];
  }
}

// 8 consecutive elements of a T tensor <-> 8 floats
__device__ __forceinline__ void ld8(const _Float16* p, float (&v)[8]) {
  const h8v t = *(const h8v*)p;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
  const f4v a = *(const f4v*)p, b = *(const f4v*)(p + 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
// f16 stores saturate at +-65504 (an inf in a tensor is a NaN in the next layer: DESIGN.md section 9, Range)
__device__ __forceinline__ void st8(_Float16* p, const float (&v)[8]) {
  h8v t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = (_Float16)__builtin_amdgcn_fmed3f(v[j], -65504.0f, 65504.0f);
  *(h8v*)p = t;
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
  *(f4v*)p = f4v{v[0], v[1], v[2], v[3]};
  *(f4v*)(p + 4) = f4v{v[4], v[5], v[6], v[7]};
}

// =================================================================================================== implicit GEMM
template <typename T, int BM, int BN, int WM, int WN, int NS>
struct GemmGeom {
  static constexpr int NW = WM * WN, NT = 64 * NW;
  static constexpr int KG = ET<T>::KG, BK = ET<T>::BK;
  static constexpr int TN = BN / WN / 32, TM = BM / WM / 32;
  static constexpr int WI = BN / 8 / NW, XI = BM / 8 / NW, LPS = WI + XI;
  static constexpr unsigned STG = (unsigned)(BN + BM) * 128u;
  static constexpr int SP = BN + 4;  // epilogue tile: floats per pixel row (+4: consecutive pixels 16 bytes apart in the bank row)
  // BIG: a tile whose f32 epilogue image would not fit LDS (256 x 256) - such a configuration has the register epilogue only
  // (f16 build, no transposed conv, no f32 output, no scale / shift pair: gemm_go refuses the rest) and fetches its residual where it is used
  static constexpr bool BIG = (unsigned)(BM * SP * 4) > 160u * 1024u || NT % (BN / 8) != 0;  // (or a 192-column tile: the staged form's thread -> chunk map needs NT % (BN / 8) == 0)
  static constexpr unsigned LDS = BIG ? NS * STG : ((NS * STG > (unsigned)(BM * SP * 4)) ? NS * STG : (unsigned)(BM * SP * 4));
  static_assert(BN % (32 * WN) == 0 && BM % (32 * WM) == 0, "wave tiles are whole 32 x 32 blocks");
  static_assert((BN / 8) % NW == 0 && (BM / 8) % NW == 0, "every wave issues the same number of DMA instructions per stage");
  static_assert(NW % 2 == 0, "the swizzle term of a lane's DMA rows must not depend on the instruction index");
  static_assert((32 * 4 * TN) % 64 == 0, "a 32-pixel block of a wave is a whole number of 64-chunk store instructions");
};

template <typename T, int BM, int BN, int WM, int WN, int NS>
__global__ void __launch_bounds__(64 * WM * WN) srv_gemm_kernel(const GemmArgs a) {
  using G = GemmGeom<T, BM, BN, WM, WN, NS>;
  constexpr int NW = G::NW, NT = G::NT, KG = G::KG, BK = G::BK, TN = G::TN, TM = G::TM, WI = G::WI, XI = G::XI, LPS = G::LPS, SP = G::SP;
  constexpr unsigned STG = G::STG;
  constexpr bool HALF = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  // ---- tile of this workgroup: column (n) tiles of one pixel tile are neighbours (they share the X panel in L2)
  const unsigned nb_n = (unsigned)((a.Ncols + BN - 1) / BN);
  const unsigned bid = srv_xcd(blockIdx.x, gridDim.x);
  unsigned tn = bid % nb_n, tm = bid / nb_n;
  if (a.group_m > 0) {  // (wave-uniform)
    const unsigned nb_m = (unsigned)((a.M + BM - 1) / BM), G = (unsigned)a.group_m;
    const unsigned per = G * nb_n, grp = bid / per, first = grp * G;
    const unsigned gsz = nb_m - first < G ? nb_m - first : G;
    const unsigned in = bid - grp * per;
    tm = first + in % gsz;
    tn = in / gsz;
  }
  const int n0 = (int)tn * BN;
  const long m0 = (long)tm * BM;

  // ---- buffer descriptors
  v4u rs_w, rs_x;
  {
    const unsigned long long wb = (unsigned long long)a.w, xb = (unsigned long long)a.x;
    rs_w.x = __builtin_amdgcn_readfirstlane((unsigned)wb);
    rs_w.y = __builtin_amdgcn_readfirstlane((unsigned)(wb >> 32));
    rs_w.z = __builtin_amdgcn_readfirstlane((unsigned)a.w_bytes);
    rs_w.w = 0x00020000u;
    rs_x.x = __builtin_amdgcn_readfirstlane((unsigned)xb);
    rs_x.y = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    rs_x.z = __builtin_amdgcn_readfirstlane((unsigned)a.x_bytes);
    rs_x.w = 0x00020000u;
  }
  // ---- DMA plan of this lane.  Instruction i of a wave fills rows 8 (wave + NW i) .. + 7 of a tile, lane l -> row + (l >> 3),
  // LDS slot l & 7; the slot holds granule (slot ^ ((row >> 1) & 7)) of the row (NW even: the XOR term is the lane's own)
  const int gq = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);  // source granule of this lane (pixel rows)
  unsigned wvo[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) wvo[i] = (unsigned)((wave + NW * i) * 1024 + lane * 16);  // (weights: the image is stored swizzled)
  // pixel rows of this lane
  int iy0[XI], ix0[XI];
  unsigned pb[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const long m = m0 + 8 * (wave + NW * j) + (lane >> 3);
    if (m >= a.M) { iy0[j] = -(1 << 28); ix0[j] = 0; pb[j] = 0; continue; }
    if (a.x1) {
      iy0[j] = 0; ix0[j] = 0;
      pb[j] = (unsigned)((unsigned long long)m * (unsigned)a.Cin * sizeof(T)) + (unsigned)(gq * 16);
    } else {
      const int ohw = a.OH * a.OW;
      const int n = (int)(m / ohw);
      const int rem = (int)(m - (long)n * ohw);
      const int oy = rem / a.OW, ox = rem - oy * a.OW;
      iy0[j] = oy * a.SH - a.PH;
      ix0[j] = ox * a.SW - a.PW;
      pb[j] = (unsigned)n * (unsigned)(a.H * a.W);
    }
  }
  // the tap / channel of the next K tile to issue (a K tile inside one tap: cin_shift < 0)
  int is_kt = 0, is_ky = 0, is_kx = 0, is_c0 = 0, is_buf = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    const unsigned st = lds0 + (unsigned)is_buf * STG;
    const unsigned wso = (unsigned)(((unsigned long long)is_kt * (unsigned)a.Npad + (unsigned)n0) * 128ull);
#pragma unroll
    for (int i = 0; i < WI; ++i) srv_dma16(st + (unsigned)(wave + NW * i) * 1024u, wvo[i], rs_w, wso);
    if (a.x1) {
      const unsigned xso = (unsigned)is_kt * (unsigned)(BK * sizeof(T));
#pragma unroll
      for (int j = 0; j < XI; ++j)
        srv_dma16(st + (unsigned)BN * 128u + (unsigned)(wave + NW * j) * 1024u, iy0[j] < 0 ? SRV_OOB : pb[j], rs_x, xso);
    } else {
      int ky, kx, c;
      const int k = is_kt * BK + gq * KG;
      if (a.cin_shift != -1) {
        const int tap = a.cin_shift >= 0 ? (k >> a.cin_shift) : (k / a.Cin);
        c = k - tap * a.Cin;
        ky = tap / a.KW;
        kx = tap - ky * a.KW;
      } else {
        ky = is_ky; kx = is_kx; c = is_c0 + gq * KG;
      }
      const bool kok = k < a.K;
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        const int iy = iy0[j] + ky, ix = ix0[j] + kx;
        const bool ok = kok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const unsigned off = ((pb[j] + (unsigned)(iy * a.W + ix)) * (unsigned)a.Cin + (unsigned)c) * (unsigned)sizeof(T);
        srv_dma16(st + (unsigned)BN * 128u + (unsigned)(wave + NW * j) * 1024u, ok ? off : SRV_OOB, rs_x, 0u);
      }
      if (a.cin_shift == -1) {
        if (a.korder) {  // (channel tile, tap): the taps of a channel tile are consecutive K tiles
          if (++is_kx == a.KW) { is_kx = 0; if (++is_ky == a.KH) { is_ky = 0; is_c0 += BK; } }
        } else {
          is_c0 += BK;
          if (is_c0 >= a.Cin) { is_c0 = 0; if (++is_kx == a.KW) { is_kx = 0; ++is_ky; } }
        }
      }
    }
    ++is_kt;
    if (++is_buf == NS) is_buf = 0;
  };

  // ---- fragment addresses: row r of a 32-row block, granule (2 s + h) (f32 twin: granule g, both halves) at its swizzled slot
  const int swz = (r >> 1) & 7;
  f16x acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int nkt = a.nkt;
  // ---- what the register epilogue (f16 build, below) needs from memory is requested BEFORE the K loop: its parameters and
  // the residual chunks of this lane.  Loaded where they are used, each tile paid a full memory latency between its last
  // matrix instruction and its first store (probe: 0.34 ms of a 1.06 ms launch on SVTR's 192 -> 768 linear).
  // (the f16 build folds batch norm into weights and bias - srv_net.hip - so a scale / shift pair only reaches the staged form)
  const bool direct = HALF && !a.deconv && !a.out_f32 && !a.scale;
  constexpr bool PRE_R = !G::BIG;  // (a 256 x 256 tile's residual is 64 registers per lane: fetched when the K loop is over instead)
  // line layout of a wave's 32-pixel block (the epilogue below): CPR 16-byte chunks per pixel row, lane -> (row t RPI + lane / CPR, chunk lane % CPR)
  // (lane l of instruction t = chunk 64 t + l of the block, row-major; CPR a power of two - 64 / CPR whole rows per instruction - or 12)
  constexpr int CPR = 4 * TN, NI = 32 * CPR / 64;
  constexpr bool CPOW2 = (CPR & (CPR - 1)) == 0;
  constexpr int RPB = TN >= 4 ? 1 : (CPOW2 ? 4 / TN : 1);  // pixel rows per 256 bytes of LDS: the XOR term changes every RPB rows
  auto l_row = [&](int t) __attribute__((always_inline)) { return (64 * t + lane) / CPR; };
  auto l_chk = [&](int t) __attribute__((always_inline)) { return (64 * t + lane) % CPR; };
  auto cell_off = [&](int row, int k) __attribute__((always_inline)) {  // a 16-byte cell of a wave's block image (unswizzled where CPR is 12)
    return (unsigned)row * (unsigned)(CPR * 16) + (unsigned)((CPOW2 ? (k ^ ((row / RPB) & (CPR - 1))) : k) * 16);
  };
  float pre_b[TN][2][8];
  h8v res_l[TM][NI];
  auto fetch_res = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int t = 0; t < NI; ++t) {
        const long m = m0 + wm * TM * 32 + j * 32 + l_row(t);
        const int n = n0 + wn * TN * 32 + 8 * l_chk(t);
        long rp = m < a.M ? m : a.M - 1;
        if (a.res_up == 2) {
          const int ohw = a.OH * a.OW;
          const int ni = (int)(rp / ohw);
          const int rem = (int)(rp - (long)ni * ohw);
          const int oy = rem / a.OW, ox = rem - oy * a.OW;
          rp = ((long)ni * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1);
        }
        if constexpr (HALF) res_l[j][t] = n < a.Cs_out ? *(const h8v*)((const _Float16*)a.res + rp * a.Cs_out + n) : h8v{0, 0, 0, 0, 0, 0, 0, 0};
      }
  };
  const bool ctc_regs = HALF && a.ctc_part != nullptr;  // the CTC head's partials, from the registers (below)
  if (direct || ctc_regs) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int n = n0 + wn * TN * 32 + i * 32 + 8 * h + 16 * c;
#pragma unroll
        for (int e = 0; e < 8; ++e) pre_b[i][c][e] = 0.f;
        if (a.bias && n + 8 <= a.Ncols) ld8(a.bias + n, pre_b[i][c]);
        else if (a.bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (n + e < a.Ncols) pre_b[i][c][e] = a.bias[n + e];
        }
    }
    if (direct && PRE_R && a.res_up) fetch_res();
  }
  // prologue: NS - 1 stages in flight
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue();
  int cbuf = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    // stage kt has landed when at most the stages issued after it are outstanding
    if (kt + NS - 1 <= nkt) srv_wait_vm<LPS * (NS - 2)>();   // (steady state: NS - 2 younger stages in flight)
    else srv_wait_vm<0>();                                  // (tail: fewer were issued)
    __builtin_amdgcn_s_barrier();                           // every wave's pieces of stage kt are in LDS; stage kt - 1 is read out
    asm volatile("" ::: "memory");
    // The next stage's DMA: with three ring slots it goes out BEHIND this stage's matrix instructions (below) - a DMA piece costs
    // its wave ~150 issue cycles (stamps: tools/micro/srv_mlp_probe), which then pass while the matrix pipe works; with two slots
    // it is the only stage in flight and goes out first (probe: 128x64/s2 0.97 -> 1.02 ms on 192 -> 768 when it went out last)
    if (NS == 2 && kt + NS - 1 < nkt) issue();
    const unsigned char* sw = smem + (unsigned)cbuf * STG + (unsigned)(wn * TN * 32 + r) * 128u;
    const unsigned char* sx = smem + (unsigned)cbuf * STG + (unsigned)BN * 128u + (unsigned)(wm * TM * 32 + r) * 128u;
    if constexpr (HALF) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const unsigned ko = (unsigned)(((2 * s + h) ^ swz) * 16);
        h8v fa[TN], fb[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) fa[i] = *(const h8v*)(sw + i * 4096 + ko);
#pragma unroll
        for (int j = 0; j < TM; ++j) fb[j] = *(const h8v*)(sx + j * 4096 + ko);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const unsigned ko = (unsigned)((g ^ swz) * 16);
        f4v fa[TN], fb[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) fa[i] = *(const f4v*)(sw + i * 4096 + ko);
#pragma unroll
        for (int j = 0; j < TM; ++j) fb[j] = *(const f4v*)(sx + j * 4096 + ko);
        // v_mfma_f32_32x32x2_f32: lane half h supplies k = h; two instructions per granule, k ascending
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
              const float av = h ? fa[i][2 * e + 1] : fa[i][2 * e];
              const float bv = h ? fb[j][2 * e + 1] : fb[j][2 * e];
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
            }
      }
    }
    if (NS > 2 && kt + NS - 1 < nkt) issue();               // into the buffer stage kt - 1 occupied (every wave is past this stage's barrier)
    if (++cbuf == NS) cbuf = 0;
  }
#ifdef SRV_PROBE_NOEPI  // development probe: the K loop alone (one value per lane keeps the accumulators alive)
  {
    float keep = 0.f;
    for (int i = 0; i < TN; ++i) for (int j = 0; j < TM; ++j) for (int q = 0; q < 16; ++q) keep += acc[i][j][q];
    if (keep == 123456.75f) ((float*)a.y)[tid] = keep;
    return;
  }
#endif
  // ---- epilogue, f16 build, from the registers (round 6: the staged form below cost 0.28-0.5 ms of a 1.1 ms launch on SVTR's fc1
  // shapes - tools/micro/srv_gemm_probe, knock-out table in DESIGN.md section 10 - an f32 tile written to LDS at ~80 bytes per
  // clock, two barriers, nothing in flight meanwhile).  A lane (pixel r, half h) holds channels 8q + 4h + e of a 32-column block
  // in register 4q + e; one v_permlane32_swap per register pair (q, q + 1) hands each lane EIGHT consecutive channels - h = 0:
  // 0..7 and 16..23, h = 1: 8..15 and 24..31 - a 16-byte chunk of the pixel's row.
  // Memory sees WHOLE LINES.  Stored straight from that layout (the round's first form) an instruction is 32 pixel rows x 32 bytes
  // - 32 partial lines for the address unit and L2, where the same kilobyte as 8 rows x 128 bytes is 8, and the DMA fills of the
  // CU's other workgroups go through that same unit (probe: the stores of a 512 -> 2048 linear cost 0.35 ms of 1.09, on every
  // tile shape alike; profiles/r6_gemm_probe_line_stores.txt: 6-15 % per launch).  So a wave passes its 32-pixel block through
  // 2-8 KB of LDS of its own (the ring's memory: one barrier, after it no wave reads a stage any more): the residual - fetched
  // before the K loop in line layout, lane -> (row, chunk) with a row's chunks on neighbouring lanes - goes in as lines and is
  // read as chunks (pixel r, chunk 4 i + 2 c + h); the result overwrites its chunk and leaves as lines.  Wave-private: LDS
  // executes a wave's instructions in order, no barrier inside.
  if constexpr (HALF) {
    if (ctc_regs) {
      // ---- CTC head (GemmArgs::ctc_part): no logits leave the workgroup.  A lane folds the 16 TN logits it holds of its row (bias added;
      // columns past Ncols do not exist) into (max, first arg max, sum of exp(x - max)), its half-wave partner adds the other half, the
      // WN waves of a pixel row meet in LDS, and one lane per row writes the column tile's partial.  Big tiles (256 x 256) may run this:
      // nothing is staged
      constexpr float L2E = 1.44269504088896341f;
      __syncthreads();
      float* const part = (float*)smem;  // [wave][j][row r] x (max, sum, arg max, -)
      auto fold = [&](float& mx, float& sum, int& mi, float om, float os, int oi) __attribute__((always_inline)) {
        const float M2 = fmaxf(mx, om);
        const float sa = mx == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f((mx - M2) * L2E);
        const float sb = om == -INFINITY ? 0.f : os * __builtin_amdgcn_exp2f((om - M2) * L2E);
        if (om > mx || (om == mx && oi < mi)) mi = oi;
        mx = M2;
        sum = sa + sb;
      };
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        float mx = -INFINITY, sum = 0.f;
        int mi = 0x7fffffff;
        // (the eight swaps of the storing epilogue first: the prefetched bias vectors are laid out for its chunks - registers 8 c + e of
        // block i = columns 32 i + 16 c + 8 h + e, ascending in (i, c, e))
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          f16x& A = acc[i][j];
          float a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5], a6 = A[6], a7 = A[7];
          float b0 = A[8], b1 = A[9], b2 = A[10], b3 = A[11], b4 = A[12], b5 = A[13], b6 = A[14], b7 = A[15];
          asm volatile(
              "s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
              "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\tv_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\ts_nop 1"
              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
          A[0] = a0; A[1] = a1; A[2] = a2; A[3] = a3; A[4] = a4; A[5] = a5; A[6] = a6; A[7] = a7;
          A[8] = b0; A[9] = b1; A[10] = b2; A[11] = b3; A[12] = b4; A[13] = b5; A[14] = b6; A[15] = b7;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int n = n0 + wn * TN * 32 + i * 32 + 8 * h + 16 * c;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = n + e < a.Ncols ? A[8 * c + e] + pre_b[i][c][e] : -INFINITY;
              A[8 * c + e] = v;
              if (v > mx) { mx = v; mi = n + e; }  // (ascending columns within the lane: the first maximum stays)
            }
          }
        }
        if (mx != -INFINITY) {
#pragma unroll
          for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) sum += __builtin_amdgcn_exp2f((acc[i][j][q] - mx) * L2E);
        }
        {
          float am = mx, bm = mx, as = sum, bs = sum;
          float ai = __int_as_float(mi), bi = ai;  // the other half-wave's lanes hold the same rows' other columns
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane32_swap_b32 %4, %5\n\ts_nop 1"
                       : "+v"(am), "+v"(bm), "+v"(as), "+v"(bs), "+v"(ai), "+v"(bi));
          // after the swap a lane of half 0 holds (its own, the partner's) in (a, b), a lane of half 1 (the partner's, its own)
          const float om = h ? am : bm, os = h ? as : bs;
          const int oi = __float_as_int(h ? ai : bi);
          fold(mx, sum, mi, om, os, oi);
        }
        if (h == 0) *(f4v*)(part + ((wave * TM + j) * 32 + r) * 4) = f4v{mx, sum, __int_as_float(mi), 0.f};
      }
      __syncthreads();
      if (wn == 0 && h == 0) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const long m = m0 + wm * TM * 32 + j * 32 + r;
          f4v p4 = *(const f4v*)(part + (((wm * WN) * TM + j) * 32 + r) * 4);
          float mx = p4[0], sum = p4[1];
          int mi = __float_as_int(p4[2]);
#pragma unroll
          for (int w2 = 1; w2 < WN; ++w2) {
            p4 = *(const f4v*)(part + (((wm * WN + w2) * TM + j) * 32 + r) * 4);
            fold(mx, sum, mi, p4[0], p4[1], __float_as_int(p4[2]));
          }
          if (m < a.M) *(f4v*)(a.ctc_part + ((size_t)m * (size_t)a.ctc_slots + (size_t)(n0 >> 6)) * 4) = f4v{mx, sum, __int_as_float(mi), 0.f};
        }
      }
      return;
    }
    if (direct) {
      __syncthreads();
      unsigned char* const scr = smem + (unsigned)wave * (unsigned)(32 * CPR * 16);
      if (!PRE_R && a.res_up) fetch_res();
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        // the residual arrives the way the output leaves - whole lines (fetched before the K loop) - and takes the same block the other way round
        if (a.res_up) {
#pragma unroll
          for (int t = 0; t < NI; ++t) *(h8v*)(scr + cell_off(l_row(t), l_chk(t))) = res_l[j][t];
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int nb = n0 + wn * TN * 32 + i * 32 + 8 * h;
          {
            f16x& A = acc[i][j];
            float a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5], a6 = A[6], a7 = A[7];
            float b0 = A[8], b1 = A[9], b2 = A[10], b3 = A[11], b4 = A[12], b5 = A[13], b6 = A[14], b7 = A[15];
            asm volatile(
                "s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
                "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\tv_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\ts_nop 1"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
                  "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
            A[0] = a0; A[1] = a1; A[2] = a2; A[3] = a3; A[4] = a4; A[5] = a5; A[6] = a6; A[7] = a7;
            A[8] = b0; A[9] = b1; A[10] = b2; A[11] = b3; A[12] = b4; A[13] = b5; A[14] = b6; A[15] = b7;
          }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int n = nb + 16 * c;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * c + e];
            if (a.bias) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] + pre_b[i][c][e];
            }
            const int k = 4 * i + 2 * c + h;
            unsigned char* const cell = scr + cell_off(r, k);
            if (a.res_up) {
              const h8v rv = *(const h8v*)cell;
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] + (float)rv[e];
            }
            if (a.act == SACT_GELU) {
              srv_gelu8(v);
            } else if (a.act != SACT_NONE) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = srv_act_h(a.act, v[e]);
            }
            h8v hv;
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = n + e >= a.Ncols ? (_Float16)0.f : (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
            *(h8v*)cell = hv;
          }
        }
#pragma unroll
        for (int t = 0; t < NI; ++t) {
          const int row = l_row(t), kk = l_chk(t);
          const h8v hv = *(const h8v*)(scr + cell_off(row, kk));
          const long m = m0 + wm * TM * 32 + j * 32 + row;
          const int n = n0 + wn * TN * 32 + 8 * kk;
#ifdef SRV_PROBE_NOSTORE
          if (hv[0] == (_Float16)12345.f)
#endif
          if (m < a.M && n < a.Cs_out) *(h8v*)((_Float16*)a.y + m * a.Cs_out + n) = hv;
        }
      }
      return;
    }
  }
  if constexpr (G::BIG) return;  // (refused on the host)
  // ---- epilogue (f32 twin, transposed convs, f32 outputs): accumulators -> f32 LDS tile [pixel][channel] -> whole 16-byte chunks of pixel rows
  __syncthreads();  // (no DMA is outstanding: the last iteration waited for vmcnt(0))
  float* const tile = (float*)smem;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // register 4 q + e of lane (r, h) = channel 8 q + 4 h + e of the block's 32, pixel r
        float* d = tile + (wm * TM * 32 + j * 32 + r) * SP + wn * TN * 32 + i * 32 + 8 * q + 4 * h;
        *(f4v*)d = f4v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      }
  __syncthreads();
  constexpr int CP = BN / 8;  // 8-channel chunks per pixel row of the tile
  const int ch = tid % CP;
  const int n = n0 + 8 * ch;
  const int ncols_store = a.deconv ? a.Ncols : a.Cs_out;
  if (n >= ncols_store) return;
  float pbias[8], pscale[8], pshift[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    pbias[e] = a.bias ? a.bias[n + e] : 0.f;
    pscale[e] = a.scale ? a.scale[n + e] : 1.f;
    pshift[e] = a.scale ? a.shift[n + e] : 0.f;
  }
  int dq = 0, co = n;
  if (a.deconv) { dq = n / a.CoutD; co = n - dq * a.CoutD; }
  for (int p = tid / CP; p < BM; p += NT / CP) {
    const long m = m0 + p;
    if (m >= a.M) break;
    float v[8];
    {
      const f4v lo = *(const f4v*)(tile + p * SP + 8 * ch), hi = *(const f4v*)(tile + p * SP + 8 * ch + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e]; v[4 + e] = hi[e]; }
    }
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] + pbias[e];
    }
    if (a.scale) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[e] * pscale[e]; v[e] = t + pshift[e]; }
    }
    long opix = m;
    if (a.res_up == 2 || a.deconv) {
      const int ohw = a.OH * a.OW;
      const int ni = (int)(m / ohw);
      const int rem = (int)(m - (long)ni * ohw);
      const int oy = rem / a.OW, ox = rem - oy * a.OW;
      if (a.deconv) opix = ((long)ni * (2 * a.OH) + 2 * oy + (dq >> 1)) * (2 * a.OW) + 2 * ox + (dq & 1);
      if (a.res_up == 2) {
        const long rp = ((long)ni * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1);
        float rv[8];
        ld8((const T*)a.res + rp * a.Cs_out + n, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] + rv[e];
      }
    }
    if (a.res_up == 1) {
      float rv[8];
      ld8((const T*)a.res + m * a.Cs_out + n, rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] + rv[e];
    }
    if (a.act != SACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = srv_act(a.act, v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (n + e >= a.Ncols) v[e] = 0.f;  // pad channels hold zeros
#ifdef SRV_PROBE_NOSTORE  // development probe (tools/micro/srv_gemm_probe.hip): everything but the output stores
    if (v[0] == 123456.75f) st8((T*)a.y + opix * a.Cs_out + co, v);
#else
    if (a.out_f32) st8((float*)a.y + opix * a.Cs_out + co, v);
    else st8((T*)a.y + opix * a.Cs_out + co, v);
#endif
  }
}

#include "srv_pgemm.h"
#include "srv_mlp.h"

// =================================================================================================== 3x3 convs on a halo patch
// srv_conv3_kernel<BN>: 3x3 stride-1 pad-1 conv, Cin a multiple of 64, f16 build, weight image in K order (channel tile, tap)
// (GemmArgs::korder = 1).  The implicit-GEMM form above fetches a pixel tile once per TAP - nine DMA passes over (nearly) the same
// pixels, 16 KB per 128 pixels each, and on the 64-column layers (res2's 3x3, the FPN / head 256 -> 64 convs: 2.8 ms of the detector)
// that traffic through the CU's address unit, not the matrix pipe, sets the time.  Here a workgroup owns a 16 x 16 pixel tile and
// DMAs its 18 x 18 x 64-channel halo patch ONCE per channel tile (41 KB, double-buffered: the next channel tile's patch travels
// during this one's nine taps); a tap is then an address offset into the patch - row (y + dy) 18 + x + dx, the row's own XOR
// swizzle - and only the weight tiles (BN rows x 128 bytes per tap, a kernel ROW's three taps per stage and barrier) stream through a three-slot ring.  Per tap and workgroup:
// BN x 128 + 41 K / 9 bytes by DMA instead of (BN + 256) x 128.  Same accumulation order as the implicit-GEMM form on the same
// image: bit-identical (test_every_tile_configuration_gives_the_same_bits).  8 waves: wave = (wm = wave / 2: pixel rows 4 wm .. + 3
// of the tile = two 32-pixel blocks, wn = wave % 2: BN / 2 columns).  Epilogue as srv_gemm_kernel's (registers, whole lines out).
template <int BN>
struct Conv3Geom {
  static constexpr int NW = 8, NT = 512, TM = 2, TN = BN / 64, WI = BN / 8 / NW;
  static constexpr int PROWS = 328, PINSTR = 41;                     // 18 x 18 = 324 patch pixels, in 8-row DMA pieces
  static constexpr unsigned PBUF = PROWS * 128u, WSLOT = 3 * BN * 128u;  // a ring slot = the three taps of one kernel row
  static constexpr unsigned WRING = 2 * PBUF, LDS = WRING + 3 * WSLOT;
  static_assert(BN == 64, "column tile (128 columns: three-tap slots do not fit beside the patches)");
};
template <int BN>
__global__ void __launch_bounds__(512) srv_conv3_kernel(const GemmArgs a) {
  using G = Conv3Geom<BN>;
  constexpr int NW = G::NW, TM = G::TM, TN = G::TN, WI = G::WI;
  constexpr unsigned PBUF = G::PBUF, WSLOT = G::WSLOT, WRING = G::WRING;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave & 1, wm = wave >> 1;
  // ---- tile: column tiles of one pixel tile are neighbours in the grid
  const unsigned nb_n = (unsigned)((a.Ncols + BN - 1) / BN);
  const unsigned bid = srv_xcd(blockIdx.x, gridDim.x);
  const int n0 = (int)(bid % nb_n) * BN;
  const unsigned pt = bid / nb_n;
  const unsigned tpx = (unsigned)((a.W + 15) >> 4), tpy = (unsigned)((a.H + 15) >> 4);
  const int tx0 = (int)(pt % tpx) << 4;
  const int ty0 = (int)((pt / tpx) % tpy) << 4;
  const int img = (int)(pt / (tpx * tpy));
  v4u rs_w, rs_x;
  {
    const unsigned long long wb = (unsigned long long)a.w, xb = (unsigned long long)a.x;
    rs_w.x = __builtin_amdgcn_readfirstlane((unsigned)wb);
    rs_w.y = __builtin_amdgcn_readfirstlane((unsigned)(wb >> 32));
    rs_w.z = __builtin_amdgcn_readfirstlane((unsigned)a.w_bytes);
    rs_w.w = 0x00020000u;
    rs_x.x = __builtin_amdgcn_readfirstlane((unsigned)xb);
    rs_x.y = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    rs_x.z = __builtin_amdgcn_readfirstlane((unsigned)a.x_bytes);
    rs_x.w = 0x00020000u;
  }
  // ---- DMA plans.  Patch piece k (of 41) = patch rows 8 k .. 8 k + 7, wave w issues pieces w, w + 8, ..; lane l -> row + (l >> 3),
  // LDS slot l & 7 holding source granule (l & 7) ^ ((row >> 1) & 7) (8 | the piece stride in rows x 4: the XOR term is the lane's own)
  const int gq = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  const int npp = wave < G::PINSTR - 5 * NW ? 6 : 5;  // pieces of this wave
  unsigned pvo[6];
  int piy[6], pix[6];  // (concatenated input: the piece's image pixel, -1 = outside)
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = 8 * (wave + NW * i) + (lane >> 3);
    const int py = q / 18, px = q - py * 18;
    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
    const bool ok = q < 324 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    pvo[i] = ok ? (unsigned)(((unsigned long long)((long)img * a.H + iy) * a.W + ix) * (unsigned)a.Cin * 2ull) + (unsigned)(gq * 16) : SRV_OOB;
    piy[i] = ok ? iy : -1;
    pix[i] = ix;
  }
  unsigned wvo[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) wvo[i] = (unsigned)((wave + NW * i) * 1024 + lane * 16);
  const int NC = a.Cin >> 6, S = 3 * NC;  // a stage = (channel tile, kernel row): three taps, one barrier
  int issued = 0;
  auto issue_patch = [&](int c) __attribute__((always_inline)) {
    const unsigned dst = lds0 + (unsigned)(c & 1) * PBUF;
    if (a.cat_n) {  // channel tile c = source c of the concatenation, at (y >> sh, x >> sh) of its own map
      const int cc = c < 3 ? c : 3;
      const unsigned long long xb = (unsigned long long)a.cat_x[cc];
      v4u rs_c;
      rs_c.x = __builtin_amdgcn_readfirstlane((unsigned)xb);
      rs_c.y = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
      rs_c.z = __builtin_amdgcn_readfirstlane((unsigned)a.cat_bytes[cc]);
      rs_c.w = 0x00020000u;
      const int sh = a.cat_sh[cc], hc = a.H >> sh, wc = a.W >> sh;
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i < npp) {
          const unsigned off = piy[i] < 0 ? SRV_OOB
                                          : (unsigned)(((unsigned long long)((long)img * hc + (piy[i] >> sh)) * wc + (pix[i] >> sh)) * 128ull) + (unsigned)(gq * 16);
          srv_dma16(dst + (unsigned)(wave + NW * i) * 1024u, off, rs_c, 0u);
        }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        if (i < npp) srv_dma16(dst + (unsigned)(wave + NW * i) * 1024u, pvo[i], rs_x, (unsigned)c * 128u);
    }
    issued += npp;
  };
  auto issue_w = [&](int st) __attribute__((always_inline)) {  // K tiles 3 st .. 3 st + 2 = (channel tile st / 3, kernel row st % 3, dx = 0 .. 2)
    const unsigned dst = lds0 + WRING + (unsigned)(st % 3) * WSLOT;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const unsigned wso = (unsigned)(((unsigned long long)(3 * st + d) * (unsigned)a.Npad + (unsigned)n0) * 128ull);
#pragma unroll
      for (int i = 0; i < WI; ++i) srv_dma16(dst + (unsigned)d * (unsigned)(BN * 128) + (unsigned)(wave + NW * i) * 1024u, wvo[i], rs_w, wso);
    }
    issued += 3 * WI;
  };
  // ---- this lane's pixel rows in the patch (tap (0, 0)): block j of the wave = tile rows 4 wm + 2 j, + 1
  int q0[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) q0[j] = (4 * wm + 2 * j + (r >> 4)) * 18 + (r & 15);
  const int swz = (r >> 1) & 7;
  f16x acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  float pre_b[TN][2][8];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = n0 + wn * TN * 32 + i * 32 + 8 * h + 16 * c;
#pragma unroll
      for (int e = 0; e < 8; ++e) pre_b[i][c][e] = 0.f;
      if (a.bias) ld8(a.bias + n, pre_b[i][c]);
    }
  // ---- prologue: patch of channel tile 0, weights of taps 0 and 1
  issue_patch(0);
  issue_w(0);
  int mark_w[3] = {0, 0, 0};  // `issued` after the weights of the stage in ring slot u went out
  mark_w[0] = issued;
  if (S > 1) { issue_w(1); mark_w[1] = issued; }
  int c = 0, dy = 0;
  for (int st = 0; st < S; ++st) {
    // the weights of stage st have landed when at most what was issued after them is outstanding (the patch of this channel tile is older)
    srv_wait_vm_le(__builtin_amdgcn_readfirstlane(issued - mark_w[st % 3]));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (dy == 0 && c + 1 < NC) issue_patch(c + 1);  // into the buffer channel tile c - 1 was read from: every wave is past it
    const unsigned char* sp = smem + (unsigned)(c & 1) * PBUF;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const unsigned char* sw = smem + WRING + (unsigned)(st % 3) * WSLOT + (unsigned)dx * (unsigned)(BN * 128) + (unsigned)(wn * TN * 32 + r) * 128u;
      unsigned po[TM], ps[TM];
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int q = q0[j] + dy * 18 + dx;
        po[j] = (unsigned)q * 128u;
        ps[j] = (unsigned)((q >> 1) & 7);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        h8v fa[TN], fb[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) fa[i] = *(const h8v*)(sw + i * 4096 + (((2 * s4 + h) ^ swz) << 4));
#pragma unroll
        for (int j = 0; j < TM; ++j) fb[j] = *(const h8v*)(sp + po[j] + ((((unsigned)(2 * s4 + h)) ^ ps[j]) << 4));
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    // the weights of stage st + 2, into the slot stage st - 1 left (every wave is past it).  (Issued by one wave of each SIMD before and by
    // the other behind the matrix instructions - so that DMA issue and matrix work of the pair overlap - measured the same: 0.683 / 0.679 ms.)
    if (st + 2 < S) { issue_w(st + 2); mark_w[(st + 2) % 3] = issued; }
    if (++dy == 3) { dy = 0; ++c; }
  }
  // ---- epilogue (srv_gemm_kernel's): bias, activation, one rounding; a wave's 32-pixel block through wave-private LDS, out as whole lines
  __syncthreads();
  constexpr int CPR = 4 * TN, RPI = 64 / CPR, NI = 32 / RPI;
  constexpr int RPB = TN >= 4 ? 1 : 4 / TN;
  unsigned char* const scr = smem + (unsigned)wave * (unsigned)(32 * CPR * 16);
#pragma unroll
  for (int j = 0; j < TM; ++j) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n0 + wn * TN * 32 + i * 32 + 8 * h;
      {
        f16x& A = acc[i][j];
        float a0 = A[0], a1 = A[1], a2 = A[2], a3 = A[3], a4 = A[4], a5 = A[5], a6 = A[6], a7 = A[7];
        float b0 = A[8], b1 = A[9], b2 = A[10], b3 = A[11], b4 = A[12], b5 = A[13], b6 = A[14], b7 = A[15];
        asm volatile(
            "s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
            "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\tv_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\ts_nop 1"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3),
              "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        A[0] = a0; A[1] = a1; A[2] = a2; A[3] = a3; A[4] = a4; A[5] = a5; A[6] = a6; A[7] = a7;
        A[8] = b0; A[9] = b1; A[10] = b2; A[11] = b3; A[12] = b4; A[13] = b5; A[14] = b6; A[15] = b7;
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int n = nb + 16 * cc;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * cc + e] + pre_b[i][cc][e];
        if (a.act == SACT_GELU) {
          srv_gelu8(v);
        } else if (a.act != SACT_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = srv_act_h(a.act, v[e]);
        }
        h8v hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = n + e >= a.Ncols ? (_Float16)0.f : (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
        const int k = 4 * i + 2 * cc + h;
        *(h8v*)(scr + (unsigned)r * (unsigned)(CPR * 16) + (unsigned)((k ^ ((r / RPB) & (CPR - 1))) * 16)) = hv;
      }
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int row = t * RPI + lane / CPR, kk = lane % CPR;
      const h8v hv = *(const h8v*)(scr + (unsigned)row * (unsigned)(CPR * 16) + (unsigned)((kk ^ ((row / RPB) & (CPR - 1))) * 16));
      const int oy = ty0 + 4 * wm + 2 * j + (row >> 4), ox = tx0 + (row & 15);
      const int n = n0 + wn * TN * 32 + 8 * kk;
      if (oy < a.H && ox < a.W && n < a.Cs_out) *(h8v*)((_Float16*)a.y + (((long)img * a.H + oy) * a.W + ox) * a.Cs_out + n) = hv;
    }
  }
}
template <int BN>
static bool conv3_go(const GemmArgs& a, bool half, hipStream_t s, bool query, std::string& err) {
  using G = Conv3Geom<BN>;
  if (!half || !a.korder || a.KH != 3 || a.KW != 3 || a.SH != 1 || a.SW != 1 || a.PH != 1 || a.PW != 1 || a.Cin % 64 || a.cin_shift != -1 || a.deconv ||
      a.out_f32 || a.scale || a.res_up || a.OH != a.H || a.OW != a.W) { err = "the halo form takes f16 3x3 stride-1 convs on whole channel tiles"; return false; }
  if (a.cat_n) {
    if (a.cat_n != a.Cin / 64 || a.cat_n > 4) { err = "concatenated input: one 64-channel source per channel tile"; return false; }
    for (int j = 0; j < a.cat_n; ++j)
      if (a.cat_bytes[j] >= 0xfffffff0ull || ((a.H >> a.cat_sh[j]) << a.cat_sh[j]) != a.H || ((a.W >> a.cat_sh[j]) << a.cat_sh[j]) != a.W) { err = "concatenated input: source shape"; return false; }
  }
  static LdsAttrMemo memo;
  if (!raise_dynamic_lds((const void*)srv_conv3_kernel<BN>, (int)G::LDS, memo)) { err = "dynamic LDS attribute refused"; return false; }
  if (query) return true;
  const long nb = (long)a.N * ((a.H + 15) / 16) * ((a.W + 15) / 16) * ((a.Ncols + BN - 1) / BN);
  if (nb <= 0 || nb > 0x7fffffffL) { err = "grid"; return false; }
  hipLaunchKernelGGL(srv_conv3_kernel<BN>, dim3((unsigned)nb), dim3(512), G::LDS, s, a);
  return true;
}

// ---- tile configurations (autotuned per layer at bind time: srv_net.hip)
struct GemmCfg { const char* name; int bm, bn, nt; unsigned lds_h, lds_f; };
#define SRV_CFGS(X)          \
  X(0, 256, 128, 2, 4, 3)    \
  X(1, 128, 128, 2, 2, 2)    \
  X(2, 256, 64, 4, 2, 3)     \
  X(3, 128, 64, 2, 2, 2)     \
  X(4, 128, 256, 1, 8, 3)    \
  X(5, 128, 128, 2, 4, 3)    \
  X(6, 256, 128, 4, 2, 3)    \
  X(7, 256, 128, 4, 2, 2)    \
  X(8, 128, 64, 2, 2, 3)     \
  X(9, 64, 64, 2, 2, 3)
// the persistent form (srv_pgemm.h; x1 problems only): id, f16 tile and waves, f32-twin tile and waves
#define SRV_PCFGS(X)                        \
  X(10, 128, 128, 2, 4, 64, 64, 2, 2)       \
  X(11, 128, 128, 2, 2, 64, 64, 2, 2)
// two column blocks per wave on the small tiles (a wave's pixel row = 128 bytes: whole lines in the epilogue above)
#define SRV_CFGS2(X)         \
  X(14, 128, 64, 4, 1, 2)    \
  X(15, 128, 128, 4, 2, 3)   \
  X(16, 256, 64, 4, 1, 3)    \
  X(17, 128, 64, 2, 2, 5)    \
  X(18, 64, 64, 2, 2, 6)     \
  X(19, 128, 64, 4, 1, 4)
// the halo form of the 3x3 stride-1 convs (srv_conv3_kernel: 16 x 16 pixel tile, BN columns; f16 build, K order 1 only)
#define SRV_HCFGS(X) \
  X(20, 64)
// big tiles (f16 build only; the f32 twin of such a choice runs 128x128/2x2/s2 - its bits do not depend on the tile)
#define SRV_BCFGS(X)         \
  X(12, 256, 256, 2, 4, 2)   \
  X(13, 256, 256, 4, 2, 2)
// 192-column tiles (f16 build, register epilogues only): SVTR's qkv widths are multiples of 192 (576, 768, 1536) - a 128-column tile
// leaves the last column tile of 576 half empty, and these launches' time goes with their tile COUNT (ids follow the table's order)
#define SRV_BCFGS2(X)        \
  X(21, 128, 192, 2, 2, 2)   \
  X(22, 256, 192, 4, 2, 2)
static const GemmCfg g_cfgs[] = {
#define X(id, BM, BN, WM, WN, NS) {#BM "x" #BN "/" #WM "x" #WN "/s" #NS, BM, BN, 64 * WM * WN, GemmGeom<_Float16, BM, BN, WM, WN, NS>::LDS, GemmGeom<float, BM, BN, WM, WN, NS>::LDS},
    SRV_CFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, FM, FN, FWM, FWN) {"p" #BM "x" #BN "/" #WM "x" #WN, BM, BN, 64 * WM * WN, PGeom<_Float16, BM, BN, WM, WN, 3, false>::LDS, PGeom<float, FM, FN, FWM, FWN, 3, false>::LDS},
    SRV_PCFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, NS) {#BM "x" #BN "/" #WM "x" #WN "/s" #NS, BM, BN, 64 * WM * WN, GemmGeom<_Float16, BM, BN, WM, WN, NS>::LDS, GemmGeom<float, 128, 128, 2, 2, 2>::LDS},
    SRV_BCFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, NS) {#BM "x" #BN "/" #WM "x" #WN "/s" #NS, BM, BN, 64 * WM * WN, GemmGeom<_Float16, BM, BN, WM, WN, NS>::LDS, GemmGeom<float, BM, BN, WM, WN, NS>::LDS},
    SRV_CFGS2(X)
#undef X
#define X(id, BN) {"halo16x16x" #BN, 256, BN, 512, Conv3Geom<BN>::LDS, 0u},
    SRV_HCFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, NS) {#BM "x" #BN "/" #WM "x" #WN "/s" #NS, BM, BN, 64 * WM * WN, GemmGeom<_Float16, BM, BN, WM, WN, NS>::LDS, GemmGeom<float, 128, 128, 2, 2, 2>::LDS},
    SRV_BCFGS2(X)
#undef X
};
int gemm_num_configs() { return (int)(sizeof(g_cfgs) / sizeof(g_cfgs[0])); }
int gemm_config_bn(int cfg) { return cfg >= 0 && cfg < gemm_num_configs() ? g_cfgs[cfg].bn : 0; }
const char* gemm_config_name(int cfg) { return cfg >= 0 && cfg < gemm_num_configs() ? g_cfgs[cfg].name : "?"; }

template <typename T, int BM, int BN, int WM, int WN, int NS>
static bool gemm_go(const GemmArgs& a, hipStream_t s, bool query, std::string& err) {
  using G = GemmGeom<T, BM, BN, WM, WN, NS>;
  auto kern = srv_gemm_kernel<T, BM, BN, WM, WN, NS>;
  static LdsAttrMemo memo;
  if (G::BIG && !(sizeof(T) == 2 && !a.deconv && !a.scale && (!a.out_f32 || a.ctc_part))) { err = "a big tile has the register epilogues only"; return false; }
  if (a.ctc_part && sizeof(T) != 2) { err = "CTC partials: f16 build"; return false; }
  if (a.cat_n) { err = "a concatenated input: the halo form only"; return false; }
  if (a.ctc_part && !(a.out_f32 && !a.deconv && !a.res_up && !a.scale && a.act == SACT_NONE && a.ctc_slots >= (a.Ncols + 63) / 64)) { err = "CTC partials: a plain f32-output linear"; return false; }
  if (G::LDS > 64 * 1024 && !raise_dynamic_lds((const void*)kern, (int)G::LDS, memo)) { err = "dynamic LDS attribute refused"; return false; }
  if (query) return true;
  const long nb = ((a.M + BM - 1) / BM) * ((a.Ncols + BN - 1) / BN);
  if (nb <= 0 || nb > 0x7fffffffL) { err = "grid"; return false; }
  // a launch with fewer K tiles than ring slots uses that many slots only (the register epilogue's blocks fit one slot): the thin
  // 1x1 convs of the detector's first stages have ONE tile, and what hides their DMA latency is workgroups per CU
  unsigned lds = G::LDS;
  if (sizeof(T) == 2 && !a.deconv && !a.out_f32 && !a.scale && a.nkt < NS) lds = (unsigned)a.nkt * G::STG;
  hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(G::NT), lds, s, a);
  return true;
}
template <typename T, int BM, int BN, int WM, int WN, bool OF32>
static bool pgemm_go(const GemmArgs& a, hipStream_t s, bool query, std::string& err) {
  using G = PGeom<T, BM, BN, WM, WN, 3, OF32>;
  auto kern = srv_pgemm_kernel<T, BM, BN, WM, WN, 3, OF32>;
  static LdsAttrMemo memo;
  if (!a.x1) { err = "the persistent form takes 1x1 problems"; return false; }
  if (a.ctc_part || a.cat_n) { err = "the persistent form has neither the CTC epilogue nor a concatenated input"; return false; }
  if (a.res_up && (OF32 || a.res_bytes >= 0xfffffff0ull)) { err = "residual"; return false; }
  if (a.y_bytes >= 0xfffffff0ull) { err = "output beyond the 4 GB a buffer descriptor spans"; return false; }
  if (G::LDS > 64 * 1024 && !raise_dynamic_lds((const void*)kern, (int)G::LDS, memo)) { err = "dynamic LDS attribute refused"; return false; }
  if (query) return true;
  const long ntiles = ((a.M + BM - 1) / BM) * ((a.Ncols + BN - 1) / BN);
  if (ntiles <= 0 || ntiles > 0x7fffffffL) { err = "grid"; return false; }
  static const int ncu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, dev) == hipSuccess) n = pr.multiProcessorCount; }
    return n > 0 ? n : 256;
  }();
  const int per_cu = (int)(160 * 1024 / G::LDS) > 0 ? (int)(160 * 1024 / G::LDS) : 1;
  const long g0 = std::min<long>(ntiles, (long)ncu * per_cu);
  const int tpb = (int)((ntiles + g0 - 1) / g0);
  const unsigned grid = (unsigned)((ntiles + tpb - 1) / tpb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NT), G::LDS, s, a, tpb);
  return true;
}
static bool gemm_dispatch(const GemmArgs& a, bool half, int cfg, hipStream_t s, bool query, std::string& err) {
  // shape contract of the kernel
  if (a.x_bytes >= 0xfffffff0ull || a.w_bytes >= 0xfffffff0ull) { err = "tensor beyond the 4 GB a buffer descriptor spans"; return false; }
  if (a.Npad % 256 || a.Ncols > a.Npad || a.nkt < 1) { err = "weight image shape"; return false; }
  switch (cfg * 2 + (half ? 1 : 0)) {
#define X(id, BM, BN, WM, WN, NS)                                                   \
    case id * 2 + 1: return gemm_go<_Float16, BM, BN, WM, WN, NS>(a, s, query, err); \
    case id * 2: return gemm_go<float, BM, BN, WM, WN, NS>(a, s, query, err);
    SRV_CFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, FM, FN, FWM, FWN)                                                                          \
    case id * 2 + 1: return a.out_f32 ? pgemm_go<_Float16, BM, BN, WM, WN, true>(a, s, query, err) : pgemm_go<_Float16, BM, BN, WM, WN, false>(a, s, query, err); \
    case id * 2: return pgemm_go<float, FM, FN, FWM, FWN, false>(a, s, query, err);
    SRV_PCFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, NS)                                                   \
    case id * 2 + 1: return gemm_go<_Float16, BM, BN, WM, WN, NS>(a, s, query, err); \
    case id * 2: return gemm_go<float, 128, 128, 2, 2, 2>(a, s, query, err);
    SRV_BCFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, NS)                                                   \
    case id * 2 + 1: return gemm_go<_Float16, BM, BN, WM, WN, NS>(a, s, query, err); \
    case id * 2: return gemm_go<float, BM, BN, WM, WN, NS>(a, s, query, err);
    SRV_CFGS2(X)
#undef X
#define X(id, BN)                                                  \
    case id * 2 + 1: return conv3_go<BN>(a, true, s, query, err); \
    case id * 2: err = "the halo form is f16 only"; return false;
    SRV_HCFGS(X)
#undef X
#define X(id, BM, BN, WM, WN, NS)                                                   \
    case id * 2 + 1: return gemm_go<_Float16, BM, BN, WM, WN, NS>(a, s, query, err); \
    case id * 2: return gemm_go<float, 128, 128, 2, 2, 2>(a, s, query, err);
    SRV_BCFGS2(X)
#undef X
  }
  err = "no such tile configuration";
  return false;
}
bool gemm_config_ok(const GemmArgs& a, bool half, int cfg) {
  std::string e;
  return gemm_dispatch(a, half, cfg, nullptr, true, e);
}
bool launch_gemm(const GemmArgs& a, bool half, int cfg, hipStream_t s, std::string& err) { return gemm_dispatch(a, half, cfg, s, false, err); }

static bool mlp_go(const MlpArgs& a, unsigned long long x_bytes, unsigned long long w1_bytes, unsigned long long w2_bytes, long M, int C, hipStream_t s, bool query,
                   std::string& err);
bool launch_mlp(const void* x, unsigned long long x_bytes, const void* w1, unsigned long long w1_bytes, int w1_npad, const void* w2,
                unsigned long long w2_bytes, int w2_npad, const float* b1, const float* b2, void* y, long M, int C, hipStream_t s, bool query,
                std::string& err) {
  MlpArgs a{x, x_bytes, w1, w1_bytes, w1_npad, w2, w2_bytes, w2_npad, b1, b2, y, M};
  return mlp_go(a, x_bytes, w1_bytes, w2_bytes, M, C, s, query, err);
}
bool launch_mlp_ln(const void* x, unsigned long long x_bytes, const void* w1, unsigned long long w1_bytes, int w1_npad, const void* w2,
                   unsigned long long w2_bytes, int w2_npad, const MlpLn& ln, const float* b2, void* y, long M, int C, hipStream_t s, bool query,
                   std::string& err) {
  MlpArgs a{x, x_bytes, w1, w1_bytes, w1_npad, w2, w2_bytes, w2_npad, nullptr, b2, y, M};
  a.ln_g = ln.g; a.ln_b = ln.b; a.ln_s = ln.s; a.ln_c = ln.c; a.ln_eps = ln.eps;
  if (!query && !(ln.g && ln.b && ln.s && ln.c)) { err = "fused MLP: absorbed LayerNorm without its vectors"; return false; }
  return mlp_go(a, x_bytes, w1_bytes, w2_bytes, M, C, s, query, err);
}
static bool mlp_go(const MlpArgs& a, unsigned long long x_bytes, unsigned long long w1_bytes, unsigned long long w2_bytes, long M, int C, hipStream_t s, bool query,
                   std::string& err) {
  if (x_bytes >= 0xfffffff0ull || w1_bytes >= 0xfffffff0ull || w2_bytes >= 0xfffffff0ull) { err = "tensor beyond the 4 GB a buffer descriptor spans"; return false; }
  const unsigned nb = (unsigned)((M + 127) / 128);
  static LdsAttrMemo m192, m256, m512;
  switch (C) {
#define SRV_MLP_CASE(CC, memo)                                                                                              \
    case CC:                                                                                                                \
      if (a.ln_g) {                                                                                                         \
        static LdsAttrMemo memo_ln;                                                                                         \
        if (!raise_dynamic_lds((const void*)srv_mlp_kernel<CC, true>, (int)MlpGeom<CC>::LDS, memo_ln)) { err = "dynamic LDS attribute refused"; return false; } \
        if (!query) hipLaunchKernelGGL((srv_mlp_kernel<CC, true>), dim3(nb), dim3(512), MlpGeom<CC>::LDS, s, a);            \
        return true;                                                                                                        \
      }                                                                                                                     \
      if (!raise_dynamic_lds((const void*)srv_mlp_kernel<CC, false>, (int)MlpGeom<CC>::LDS, memo)) { err = "dynamic LDS attribute refused"; return false; } \
      if (!query) hipLaunchKernelGGL((srv_mlp_kernel<CC, false>), dim3(nb), dim3(512), MlpGeom<CC>::LDS, s, a);             \
      return true;
    SRV_MLP_CASE(192, m192)
    SRV_MLP_CASE(256, m256)
    SRV_MLP_CASE(512, m512)
#undef SRV_MLP_CASE
  }
  err = "fused MLP: channel count not instantiated";
  return false;
}

// =================================================================================================== streaming kernels
template <typename T>
__global__ void __launch_bounds__(256) pack_input_kernel(const float* __restrict__ x, T* __restrict__ y, long pixels) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= pixels) return;
  float v[8] = {x[3 * p], x[3 * p + 1], x[3 * p + 2], 0.f, 0.f, 0.f, 0.f, 0.f};
  st8(y + 8 * p, v);
}
void launch_pack_input(const float* x, void* y, long pixels, bool half, hipStream_t s) {
  const unsigned nb = (unsigned)((pixels + 255) / 256);
  if (half) hipLaunchKernelGGL(pack_input_kernel<_Float16>, dim3(nb), dim3(256), 0, s, x, (_Float16*)y, pixels);
  else hipLaunchKernelGGL(pack_input_kernel<float>, dim3(nb), dim3(256), 0, s, x, (float*)y, pixels);
}

template <typename T>
__global__ void __launch_bounds__(256) pool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int Cs, int OH, int OW, int kh,
                                                   int kw, int sh, int sw, int ph, int pw, int is_max) {
  const int C8 = Cs >> 3;
  const long total = (long)N * OH * OW * C8;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % C8);
  long p = t / C8;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int n = (int)(p / OH);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = is_max ? -INFINITY : 0.f;
  int cnt = 0;
  for (int dy = 0; dy < kh; ++dy)
    for (int dx = 0; dx < kw; ++dx) {
      const int iy = oy * sh - ph + dy, ix = ox * sw - pw + dx;
      if (iy < 0 || ix < 0 || iy >= H || ix >= W) continue;
      float v[8];
      ld8(x + (((long)n * H + iy) * W + ix) * Cs + 8 * c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = is_max ? fmaxf(acc[e], v[e]) : acc[e] + v[e];
      ++cnt;
    }
  if (!is_max) {
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = acc[e] / (float)cnt;
  }
  st8(y + (((long)n * OH + oy) * OW + ox) * Cs + 8 * c, acc);
}
void launch_pool(const void* x, void* y, int N, int H, int W, int Cs, int OH, int OW, int kh, int kw, int sh, int sw, int ph, int pw, bool is_max,
                 bool half, hipStream_t s) {
  const long total = (long)N * OH * OW * (Cs >> 3);
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (half) hipLaunchKernelGGL(pool_kernel<_Float16>, dim3(nb), dim3(256), 0, s, (const _Float16*)x, (_Float16*)y, N, H, W, Cs, OH, OW, kh, kw, sh, sw, ph, pw, is_max ? 1 : 0);
  else hipLaunchKernelGGL(pool_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, (float*)y, N, H, W, Cs, OH, OW, kh, kw, sh, sw, ph, pw, is_max ? 1 : 0);
}

struct CatArgs { const void* src[4]; int up[4]; int nsrc, cs; };
template <typename T>
__global__ void __launch_bounds__(256) concat_up_kernel(const CatArgs a, T* __restrict__ y, int N, int OH, int OW) {
  const int C8 = (a.cs * a.nsrc) >> 3, c8s = a.cs >> 3;
  const long total = (long)N * OH * OW * C8;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int c = (int)(t % C8);
  long p = t / C8;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int n = (int)(p / OH);
  const int j = c / c8s, cc = c - j * c8s;
  const int u = a.up[j], sh_ = OH / u, sw_ = OW / u;
  const T* src = (const T*)a.src[j] + (((long)n * sh_ + oy / u) * sw_ + ox / u) * a.cs + 8 * cc;
  float v[8];
  ld8(src, v);
  st8(y + t * 8, v);
}
void launch_concat_up(const void* const src[4], const int up[4], int nsrc, int cs, void* y, int N, int OH, int OW, bool half, hipStream_t s) {
  CatArgs a;
  for (int j = 0; j < 4; ++j) { a.src[j] = j < nsrc ? src[j] : nullptr; a.up[j] = j < nsrc ? up[j] : 1; }
  a.nsrc = nsrc; a.cs = cs;
  const long total = (long)N * OH * OW * ((cs * nsrc) >> 3);
  const unsigned nb = (unsigned)((total + 255) / 256);
  if (half) hipLaunchKernelGGL(concat_up_kernel<_Float16>, dim3(nb), dim3(256), 0, s, a, (_Float16*)y, N, OH, OW);
  else hipLaunchKernelGGL(concat_up_kernel<float>, dim3(nb), dim3(256), 0, s, a, (float*)y, N, OH, OW);
}

template <typename T>
__global__ void __launch_bounds__(256) addpos_kernel(const T* __restrict__ x, const float* __restrict__ pos, T* __restrict__ y, long pixels, int hw, int Cs) {
  const int C8 = Cs >> 3;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= pixels * C8) return;
  const int c = (int)(t % C8);
  const long p = t / C8;
  const int q = (int)(p % hw);
  float v[8], g[8];
  ld8(x + t * 8, v);
  ld8(pos + (long)q * Cs + 8 * c, g);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = v[e] + g[e];
  st8(y + t * 8, v);
}
void launch_addpos(const void* x, const float* pos, void* y, long pixels, int hw, int Cs, bool half, hipStream_t s) {
  const unsigned nb = (unsigned)((pixels * (Cs >> 3) + 255) / 256);
  if (half) hipLaunchKernelGGL(addpos_kernel<_Float16>, dim3(nb), dim3(256), 0, s, (const _Float16*)x, pos, (_Float16*)y, pixels, hw, Cs);
  else hipLaunchKernelGGL(addpos_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, pos, (float*)y, pixels, hw, Cs);
}

// ---- layer norm.  f16: a wave per row, lane l owns granules l, l + 64, ... (f32 arithmetic, wave reductions);
// f32 twin: a thread per row, the oracle's sequential sums
// LPR = lanes per row (C / 8 granules, rounded up to a power of two: 32 for C = 192 / 256, 64 for C = 512): a wave normalises
// 64 / LPR rows at once (round 6: one row per wave left 40 of 64 lanes idle at C = 192 - 2.4 TB/s)
template <int LPR>
__global__ void __launch_bounds__(256) layernorm_h_kernel(const _Float16* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                          _Float16* __restrict__ y, long rows, int C, float eps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, l = lane % LPR;
  const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + sub;
  const int G = C >> 3;  // granules per row (<= 2 * LPR)
  const bool live = row < rows;
  float v[2][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gi = l + LPR * i;
    if (live && gi < G) {
      ld8(x + row * C + 8 * gi, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (live && l + LPR * i < G) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q = fmaf(d, d, q); }
    }
#pragma unroll
  for (int o = LPR / 2; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gi = l + LPR * i;
    if (live && gi < G) {
      float gg[8], bb[8], o[8];
      ld8(g + 8 * gi, gg);
      ld8(b + 8 * gi, bb);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
      st8(y + row * C + 8 * gi, o);
    }
  }
}
__global__ void __launch_bounds__(64) layernorm_f_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                         float* __restrict__ y, long rows, int C, float eps) {
  const long row = (long)blockIdx.x * 64 + threadIdx.x;
  if (row >= rows) return;
  const float* src = x + row * C;
  float* dst = y + row * C;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = s + src[c];
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) { const float xm = src[c] - mean; v = fmaf(xm, xm, v); }
  const float var = v / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int c = 0; c < C; ++c) {
    const float xm = src[c] - mean;
    float t = xm * rstd;
    t = t * g[c];
    dst[c] = t + b[c];
  }
}
void launch_layernorm(const void* x, const float* g, const float* b, void* y, long rows, int C, float eps, bool half, hipStream_t s) {
  if (half) {
    const int G = C >> 3;
    if (G <= 32) hipLaunchKernelGGL(layernorm_h_kernel<16>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, (const _Float16*)x, g, b, (_Float16*)y, rows, C, eps);
    else if (G <= 64) hipLaunchKernelGGL(layernorm_h_kernel<32>, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, s, (const _Float16*)x, g, b, (_Float16*)y, rows, C, eps);
    else hipLaunchKernelGGL(layernorm_h_kernel<64>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const _Float16*)x, g, b, (_Float16*)y, rows, C, eps);
  }
  else hipLaunchKernelGGL(layernorm_f_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, (const float*)x, g, b, (float*)y, rows, C, eps);
}

// ---- the DB head's last layer (64 -> 1, 2x2 stride 2, + bias, sigmoid): a thread per input pixel, four dot products
template <typename T>
__global__ void __launch_bounds__(256) deconv_map_kernel(const T* __restrict__ x, const float* __restrict__ w4, float bias, float* __restrict__ prob, int N,
                                                         int H, int W, int Cs) {
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long)N * H * W) return;
  const int ix = (int)(p % W);
  const long q = p / W;
  const int iy = (int)(q % H);
  const int n = (int)(q / H);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < Cs; c += 8) {
    float v[8];
    ld8(x + p * Cs + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = fmaf(v[e], w4[t * Cs + c + e], acc[t]);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float yv = acc[t] + bias;
    prob[((long)n * 2 * H + 2 * iy + (t >> 1)) * (2 * W) + 2 * ix + (t & 1)] = srv_act(SACT_SIGMOID, yv);
  }
}
// f16 build, 64 stored channels (the DB head's): EIGHT lanes per pixel, one 16-byte chunk each - a wave reads eight whole 128-byte
// pixel rows per instruction (the thread-per-pixel form above reads 16 bytes of 64 different lines: 1.07 TB/s on 944 MB, the
// detector's slowest launch per byte) - four partial sums per lane, folded over the eight lanes by three DPP row shifts; lanes 0..3
// of a pixel write its four map values.  (Summation order differs from the f32 twin's ascending chain: tolerance-tested mode.)
__global__ void __launch_bounds__(256) deconv_map_h64_kernel(const _Float16* __restrict__ x, const float* __restrict__ w4, float bias,
                                                             float* __restrict__ prob, int N, int H, int W) {
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  const long p = gi >> 3;
  const int ch = (int)(gi & 7);
  const bool ok = p < (long)N * H * W;
  float v[8];
  ld8(x + (ok ? p : 0) * 64 + 8 * ch, v);
  float acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float wv[8];
    ld8(w4 + t * 64 + 8 * ch, wv);
    float a0 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) a0 = fmaf(v[e], wv[e], a0);
    a0 += __shfl_xor(a0, 1);
    a0 += __shfl_xor(a0, 2);
    a0 += __shfl_xor(a0, 4);
    acc[t] = a0;
  }
  if (!ok || ch >= 4) return;
  const float yv = (ch == 0 ? acc[0] : ch == 1 ? acc[1] : ch == 2 ? acc[2] : acc[3]) + bias;
  const int ix = (int)(p % W);
  const long q = p / W;
  const int iy = (int)(q % H);
  const int n = (int)(q / H);
  prob[((long)n * 2 * H + 2 * iy + (ch >> 1)) * (2 * W) + 2 * ix + (ch & 1)] = srv_act_h(SACT_SIGMOID, yv);
}
void launch_deconv_to_map(const void* x, const float* w4, float bias, float* prob, int N, int H, int W, int Cs, bool half, hipStream_t s) {
  const unsigned nb = (unsigned)(((long)N * H * W + 255) / 256);
  if (half && Cs == 64) {
    const unsigned nb8 = (unsigned)(((long)N * H * W * 8 + 255) / 256);
    hipLaunchKernelGGL(deconv_map_h64_kernel, dim3(nb8), dim3(256), 0, s, (const _Float16*)x, w4, bias, prob, N, H, W);
    return;
  }
  if (half) hipLaunchKernelGGL(deconv_map_kernel<_Float16>, dim3(nb), dim3(256), 0, s, (const _Float16*)x, w4, bias, prob, N, H, W, Cs);
  else hipLaunchKernelGGL(deconv_map_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, w4, bias, prob, N, H, W, Cs);
}

// ---- the DB head's two transposed convs as ONE kernel (f16 build): 64 -> 64 (2x2 stride 2, + bias (batch norm folded), ReLU) -> 1
// (2x2 stride 2, + bias, sigmoid).  As two launches the 480 x 480 x 64 tensor between them is written and read back: 1.9 GB of the
// 2.2 GB the pair moves at batch 32 (0.63 + 0.35 ms); here an input pixel's 4 x 64 mid values never leave the registers:
//   product 1  D[col n = tap * 64 + c][pixel] = W1 rows x pixel rows (K = 64): eight 32 x 32 blocks per 32 pixels, W1's 32 KB image resident in LDS;
//   its accumulators - bias, ReLU, one rounding to f16 (what the tensor would have held) - ARE the B fragments of
//   product 2  D[row = second tap][pixel] += W2^T (4 of 32 rows used) x mid (K = 64 per first tap), two matrix instructions per block:
//   after the eight swaps of the register epilogue a lane's chunk c is k step c of that operand (as P feeds P V in the attention kernel).
// A wave owns 32 consecutive pixels of a 128-pixel tile; pixel rows arrive as whole lines through 4 KB of wave-private LDS; the lanes of
// half 0 end up with the pixel's 4 x 4 map values and store them as four 16-byte pieces (32 lanes = 512 contiguous bytes per map row).
__global__ void __launch_bounds__(256) head_tail_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w1img, const float* __restrict__ b1,
                                                        const float* __restrict__ w4, float bias2, float* __restrict__ prob, int N, int H, int W,
                                                        long tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const sW = smem;               // [256 rows][8 granules] as stored (granule g of row n at slot g ^ ((n >> 1) & 7))
  float* const sB = (float*)(smem + 32768);     // [256]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const scr = smem + 33792 + wave * 4096;
  const int r = lane & 31, h = lane >> 5;
  for (int i = tid; i < 2048; i += 256) ((h8v*)sW)[i] = ((const h8v*)w1img)[i];
  sB[tid] = b1[tid];
  // W2^T fragments: row r = second tap (4 real rows), element j of (first-tap half block i2, k step c) = mid channel 32 i2 + 16 c + 8 h + j
  h8v w2a[2][2];
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) w2a[i2][c][j] = r < 4 ? (_Float16)w4[r * 64 + 32 * i2 + 16 * c + 8 * h + j] : (_Float16)0.f;
  __syncthreads();
  const int swz = (r >> 1) & 7;
  const long M = (long)N * H * W;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long p0 = t * 128 + wave * 32;
    // ---- 32 pixel rows in as lines: lane -> (row 8 i + lane / 8, chunk lane % 8)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (lane >> 3), kk = lane & 7;
      const long p = p0 + row;
      *(h8v*)(scr + row * 128 + ((kk ^ ((row >> 1) & 7)) << 4)) = p < M ? *(const h8v*)(x + p * 64 + 8 * kk) : h8v{0, 0, 0, 0, 0, 0, 0, 0};
    }
    h8v fb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) fb[s] = *(const h8v*)(scr + r * 128 + (((2 * s + h) ^ swz) << 4));
    f16x acc2[4];
#pragma unroll
    for (int tp = 0; tp < 4; ++tp)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[tp][q] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      f16x acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const unsigned char* wr = sW + (nt * 32 + r) * 128;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const h8v*)(wr + (((2 * s + h) ^ swz) << 4)), fb[s], acc, 0, 0, 0);
      {
        float a0 = acc[0], a1 = acc[1], a2 = acc[2], a3 = acc[3], a4 = acc[4], a5 = acc[5], a6 = acc[6], a7 = acc[7];
        float b0 = acc[8], b1_ = acc[9], b2 = acc[10], b3 = acc[11], b4 = acc[12], b5 = acc[13], b6 = acc[14], b7 = acc[15];
        asm volatile(
            "s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\tv_permlane32_swap_b32 %3, %7\n\t"
            "v_permlane32_swap_b32 %8, %12\n\tv_permlane32_swap_b32 %9, %13\n\tv_permlane32_swap_b32 %10, %14\n\tv_permlane32_swap_b32 %11, %15\n\ts_nop 1"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1_), "+v"(b2), "+v"(b3),
              "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        acc[0] = a0; acc[1] = a1; acc[2] = a2; acc[3] = a3; acc[4] = a4; acc[5] = a5; acc[6] = a6; acc[7] = a7;
        acc[8] = b0; acc[9] = b1_; acc[10] = b2; acc[11] = b3; acc[12] = b4; acc[13] = b5; acc[14] = b6; acc[15] = b7;
      }
      // chunk c of this lane = columns nt * 32 + 16 c + 8 h .. + 7 = mid channels (nt & 1) * 32 + 16 c + 8 h .. of first tap nt >> 1
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float* bb = sB + nt * 32 + 16 * c + 8 * h;
        const f4v blo = *(const f4v*)bb, bhi = *(const f4v*)(bb + 4);
        h8v pc;
#pragma unroll
        for (int e = 0; e < 8; ++e) pc[e] = (_Float16)fminf(fmaxf(acc[8 * c + e] + (e < 4 ? blo[e] : bhi[e - 4]), 0.f), 65504.0f);
        acc2[nt >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2a[nt & 1][c], pc, acc2[nt >> 1], 0, 0, 0);
      }
    }
    // ---- register e of a half-0 lane of acc2[first tap (dy, dx)] = second tap e = (dy2, dx2) of pixel r: map rows 4 y + 2 dy + dy2
    const long p = p0 + r;
    if (h == 0 && p < M) {
      const int ix = (int)(p % W);
      const long q = p / W;
      const int iy = (int)(q % H);
      const long n = q / H;
      float* const o = prob + ((n * 4 * H + 4 * iy) * (4L * W) + 4 * ix);
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        const int dy = a4 >> 1, dy2 = a4 & 1;
        f4v v;
        v[0] = srv_act_h(SACT_SIGMOID, acc2[2 * dy][2 * dy2] + bias2);
        v[1] = srv_act_h(SACT_SIGMOID, acc2[2 * dy][2 * dy2 + 1] + bias2);
        v[2] = srv_act_h(SACT_SIGMOID, acc2[2 * dy + 1][2 * dy2] + bias2);
        v[3] = srv_act_h(SACT_SIGMOID, acc2[2 * dy + 1][2 * dy2 + 1] + bias2);
        *(f4v*)(o + (long)a4 * 4 * W) = v;
      }
    }
  }
}
bool launch_head_tail(const void* x, const void* w1img, const float* b1, const float* w4, float bias2, float* prob, int N, int H, int W, hipStream_t s,
                      bool query, std::string& err) {
  constexpr int LDS = 33792 + 4 * 4096;
  static LdsAttrMemo memo;
  if (LDS > 64 * 1024 && !raise_dynamic_lds((const void*)head_tail_kernel, LDS, memo)) { err = "dynamic LDS attribute refused"; return false; }
  if (query) return true;
  const long tiles = ((long)N * H * W + 127) / 128;
  const unsigned grid = (unsigned)std::min<long>(tiles, 1024);
  hipLaunchKernelGGL(head_tail_kernel, dim3(grid), dim3(256), LDS, s, (const _Float16*)x, (const _Float16*)w1img, b1, w4, bias2, prob, N, H, W, tiles);
  return true;
}

// ---- CTC tail of a head launch in partial mode: a thread per row folds the row's column-tile partials (ascending columns: the first maximum wins)
__global__ void __launch_bounds__(256) ctc_reduce_kernel(const float* __restrict__ part, long rows, int slots, int step, int* __restrict__ amax,
                                                         float* __restrict__ pmax) {
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  constexpr float L2E = 1.44269504088896341f;
  float mx = -INFINITY, sum = 0.f;
  int mi = 0x7fffffff;
  for (int sl = 0; sl < slots; sl += step) {
    const f4v p4 = *(const f4v*)(part + ((size_t)row * (size_t)slots + (size_t)sl) * 4);
    const float om = p4[0], os = p4[1];
    const int oi = __float_as_int(p4[2]);
    const float M2 = fmaxf(mx, om);
    const float sa = mx == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f((mx - M2) * L2E);
    const float sb = om == -INFINITY ? 0.f : os * __builtin_amdgcn_exp2f((om - M2) * L2E);
    if (om > mx || (om == mx && oi < mi)) mi = oi;
    mx = M2;
    sum = sa + sb;
  }
  amax[row] = mi;
  pmax[row] = 1.0f / sum;
}
void launch_ctc_reduce(const float* part, long rows, int slots, int step, int* amax, float* pmax, hipStream_t s) {
  hipLaunchKernelGGL(ctc_reduce_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, rows, slots, step, amax, pmax);
}

// ---- CTC tail: a wave per row; arg max = first maximum of the logits, its probability 1 / sum exp(x - max)
__global__ void __launch_bounds__(256) argmax_softmax_kernel(const float* __restrict__ logits, long rows, int C, int ld, int* __restrict__ amax,
                                                             float* __restrict__ pmax) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* x = logits + row * ld;
  float m = -INFINITY;
  int mi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = x[c];
    if (v > m) { m = v; mi = c; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float om = __shfl_xor(m, o);
    const int oi = __shfl_xor(mi, o);
    if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
  }
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += ocr_expf(x[c] - m);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) { amax[row] = mi; pmax[row] = 1.0f / s; }
}
// f16 build: ONE pass over the row (the form above reads it twice, four bytes per lane: 1.17 ms for the 2.17 GB of a step's logits,
// more than the head's matrix product), 16 bytes per lane, each lane an online softmax of its own - running maximum m, sum of
// exp(x - m) rescaled when m moves, v_exp_f32 - the 64 (m, sum, index) triples folded at the end; the first maximum wins ties.
__global__ void __launch_bounds__(256) argmax_softmax_fast_kernel(const float* __restrict__ logits, long rows, int C, int ld, int* __restrict__ amax,
                                                                  float* __restrict__ pmax) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* x = logits + row * ld;
  constexpr float L2E = 1.44269504088896341f;
  float m = -INFINITY, sum = 0.f;
  int mi = 0x7fffffff;
  for (int c = 4 * lane; c < C; c += 256) {
    f4v v = *(const f4v*)(x + c);  // (ld is a multiple of 4 and >= C rounded up to 4: the launcher checks)
    float lm = m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (c + e >= C) v[e] = -INFINITY;
      if (v[e] > lm) { lm = v[e]; mi = c + e; }
    }
    float acc = m == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f((m - lm) * L2E);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc += __builtin_amdgcn_exp2f((v[e] - lm) * L2E);
    sum = acc;
    m = lm;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float om = __shfl_xor(m, o), os = __shfl_xor(sum, o);
    const int oi = __shfl_xor(mi, o);
    const float M = fmaxf(m, om);
    const float sa = m == -INFINITY ? 0.f : sum * __builtin_amdgcn_exp2f((m - M) * L2E);
    const float sb = om == -INFINITY ? 0.f : os * __builtin_amdgcn_exp2f((om - M) * L2E);
    if (om > m || (om == m && oi < mi)) mi = oi;
    m = M;
    sum = sa + sb;
  }
  if (lane == 0) { amax[row] = mi; pmax[row] = 1.0f / sum; }
}
void launch_argmax_softmax(const float* logits, long rows, int C, int ld, int* amax, float* pmax, bool half, hipStream_t s) {
  if (half && ld % 4 == 0 && ld >= ((C + 3) & ~3)) {
    hipLaunchKernelGGL(argmax_softmax_fast_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, rows, C, ld, amax, pmax);
    return;
  }
  launch_argmax_softmax(logits, rows, C, ld, amax, pmax, s);
}
void launch_argmax_softmax(const float* logits, long rows, int C, int ld, int* amax, float* pmax, hipStream_t s) {
  hipLaunchKernelGGL(argmax_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, rows, C, ld, amax, pmax);
}

template <typename T>
__global__ void __launch_bounds__(256) to_f32_kernel(const T* __restrict__ x, float* __restrict__ y, long pixels, int Cs, int C) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= pixels * C) return;
  const long p = t / C;
  const int c = (int)(t - p * C);
  y[t] = (float)x[p * Cs + c];
}
void launch_to_f32(const void* x, float* y, long pixels, int Cs, int C, bool half, hipStream_t s) {
  const unsigned nb = (unsigned)((pixels * C + 255) / 256);
  if (half) hipLaunchKernelGGL(to_f32_kernel<_Float16>, dim3(nb), dim3(256), 0, s, (const _Float16*)x, y, pixels, Cs, C);
  else hipLaunchKernelGGL(to_f32_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, y, pixels, Cs, C);
}

// =================================================================================================== attention
// f32 twin: a thread per (line, head, query), the oracle's loops (three passes over the keys: maximum, sum, weighted values -
// a score is recomputed, by the same fma chain, instead of kept)
__global__ void __launch_bounds__(64) attn_f_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int T, int heads, float scale, int gw,
                                                    int lh, int lw) {
  constexpr int HD = 32;
  const long id = (long)blockIdx.x * 64 + threadIdx.x;
  if (id >= (long)N * heads * T) return;
  const int t = (int)(id % T);
  const int hh = (int)((id / T) % heads);
  const int n = (int)(id / ((long)T * heads));
  const int D = heads * HD;
  const float* base = qkv + (long)n * T * 3 * D;
  float q[HD];
  for (int d = 0; d < HD; ++d) q[d] = base[(long)t * 3 * D + hh * HD + d] * scale;
  const int qy = t / gw, qx = t % gw;
  auto allowed = [&](int u) {
    if (lh <= 0) return true;
    const int dy = u / gw - qy, dx = u % gw - qx;
    return dy >= -(lh / 2) && dy <= lh / 2 && dx >= -(lw / 2) && dx <= lw / 2;
  };
  auto score = [&](int u) {
    const float* kr = base + (long)u * 3 * D + D + hh * HD;
    float acc = 0.f;
    for (int d = 0; d < HD; ++d) acc = fmaf(q[d], kr[d], acc);
    return acc;
  };
  float m = -INFINITY;
  for (int u = 0; u < T; ++u) if (allowed(u)) m = fmaxf(m, score(u));
  float sum = 0.f;
  for (int u = 0; u < T; ++u) if (allowed(u)) sum = sum + ocr_expf(score(u) - m);
  float acc[HD];
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  for (int u = 0; u < T; ++u) {
    if (!allowed(u)) continue;
    const float e = ocr_expf(score(u) - m);
    const float p = e / sum;
    const float* vr = base + (long)u * 3 * D + 2 * D + hh * HD;
    for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, vr[d], acc[d]);
  }
  float* dst = out + ((long)n * T + t) * D + hh * HD;
  for (int d = 0; d < HD; ++d) dst[d] = acc[d];
}

// f16: one workgroup per (line, head); K and V of the pair in LDS ([T + 32][32] halfs each; K rows XOR-swizzled by
// (row >> 2) & 3 for conflict-free ds_read_b128, V rows linear for ds_read_b64_tr_b16), a wave per tile of 32 queries of one grid row.
// NH = heads per workgroup.  A head's K (V) row of a token is 64 bytes of the token's 3 D wide qkv row: fetched head by head, every
// 128-byte line of the tensor is read by two workgroups (PMC: FETCH = 2.0 x the qkv tensor per launch - 1.51 GB in 0.40 ms, the
// launch's real bound at stage 3).  NH = 2: a workgroup takes a PAIR of heads, its loads are whole lines, half of its waves work on each.
template <int NWV, int NH>
__global__ void __launch_bounds__(64 * NWV, NWV >= 16 ? 1 : 16 / NWV) attn_h_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out, int T, int heads,
                                                                                   float scale_log2e, int gh, int gw, int lh, int lw) {
  constexpr int HD = 32, WPH = NWV / NH;  // waves per head
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = wave_all / WPH, wave = wave_all - hl * WPH;  // this wave's head of the workgroup, its index among that head's waves
  const int r = lane & 31, h = lane >> 5;
  const int hpb = heads / NH;
  const int n = blockIdx.x / hpb, hh = (blockIdx.x % hpb) * NH + hl;
  const int D = heads * HD;
  const _Float16* base = qkv + (long)n * T * 3 * D + hh * HD;
  const size_t img = (size_t)(T + 32) * 64;
  unsigned char* sK = smem + (size_t)hl * 2 * img;
  unsigned char* sV = sK + img;
  // ---- K, V -> LDS (16-byte chunks; the 32 rows past T hold zeros: a masked key's probability is 0, and 0 x garbage must not be NaN)
  {
    const _Float16* base0 = qkv + (long)n * T * 3 * D + (blockIdx.x % hpb) * NH * HD;
    for (int i = tid; i < (T + 32) * 4 * NH; i += 64 * NWV) {
      const int t = i / (4 * NH), c8 = i % (4 * NH), hd2 = c8 >> 2, c = c8 & 3;  // (lanes of a token: consecutive 16-byte chunks of NH x 64 bytes)
      h8v kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
      if (t < T) {
        kv = *(const h8v*)(base0 + (long)t * 3 * D + D + 8 * c8);
        vv = *(const h8v*)(base0 + (long)t * 3 * D + 2 * D + 8 * c8);
      }
      unsigned char* k0 = smem + (size_t)hd2 * 2 * img;
      *(h8v*)(k0 + t * 64 + ((c ^ ((t >> 2) & 3)) << 4)) = kv;
      *(h8v*)(k0 + img + t * 64 + (c << 4)) = vv;
    }
  }
  __syncthreads();
  const int hwy = lh >> 1, hwx = lw >> 1;
  // Query tiles.  Global mixers and wide windows: 32 consecutive positions of one grid row.  Local mixers with a window of at most
  // 17 columns (SVTR's 7 x 11): TWO rows x 16 columns - the keys such a tile can see in one grid row are 16 + 2 hwx <= 32 columns, ONE
  // key tile per row (at an unaligned start), where a 32-column query tile needs up to three: 6.7 key tiles per 32 queries instead of 14.4
  // on the 12 x 80 grid, 5.3 instead of 11.7 on 6 x 80
  const bool pairs = lh > 0 && hwx <= 8;
  const int segs = pairs ? (gw + 15) >> 4 : (gw + 31) >> 5;
  const int nq = (pairs ? (gh + 1) >> 1 : gh) * segs;
  // transposed-read lane geometry (T10): group G = lane >> 4 -> d block 16 (G & 1), key sub-block 4 (G >> 1); lane 4 q + p of the group
  // supplies the address of row q, columns 4 p .. 4 p + 3
  const int trq = (lane & 15) >> 2, trp = lane & 3;
  const unsigned tr_off = (unsigned)((4 * h + trq) * 64 + (16 * ((lane >> 4) & 1) + 4 * trp) * 2);
  for (int qt = wave; qt < nq; qt += WPH) {
    const int qrow = qt / segs, qseg = qt - qrow * segs;
    const int qy0 = pairs ? 2 * qrow : qrow, qxs = qseg << (pairs ? 4 : 5);
    const int qy = pairs ? qy0 + (r >> 4) : qy0, qx = qxs + (pairs ? r & 15 : r);
    const bool qok = qx < gw && qy < gh;
    const int tq = qok ? qy * gw + qx : qy0 * gw + qxs;  // (a lane without a query computes some valid token's and stores nothing)
    // Q fragments (B operand of S^T = K Q^T): lane (query r, half h), step s: d = 16 s + 8 h .. + 7; pre-scaled by scale * log2(e)
    h8v qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const h8v t8 = *(const h8v*)(base + (long)tq * 3 * D + 16 * s + 8 * h);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[s][e] = (_Float16)((float)t8[e] * scale_log2e);
    }
    f16x o;
#pragma unroll
    for (int e = 0; e < 16; ++e) o[e] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    const int ky0 = lh > 0 ? max(qy0 - hwy, 0) : 0, ky1 = lh > 0 ? min(qy0 + (pairs ? 1 : 0) + hwy, gh - 1) : gh - 1;
    // Key tiles of a grid row: columns kx_lo, kx_lo + 32, .. below kx_end.  pairs: the ONE tile that starts 8 columns left of the
    // queries (columns qxs - 8 .. qxs + 23 hold every window), pulled inside the grid row; a wide local window: the aligned tiles it
    // touches; a global mixer: all of them
    const int kx_lo = pairs ? max(min(qxs - 8, gw - 32), 0) : (lh > 0 ? (max(qxs - hwx, 0) >> 5) << 5 : 0);
    const int kx_end = pairs ? kx_lo + 1 : (lh > 0 ? min(qxs + 31 + hwx, gw - 1) + 1 : gw);
    const int ntl = (ky1 - ky0 + 1) * ((kx_end - kx_lo + 31) >> 5);
    // the column part of the mask as an additive 0 / -inf that the S accumulators START from (register i of lane (q, h) = key
    // 8 (i >> 2) + 4 h + (i & 3) of the tile): with one tile per row it is the same for every row - computed once per query tile,
    // free afterwards.  (Round 6: the mask was 96 VALU instructions per tile, a third of the kernel's issue slots.)
    float cb[16];
    auto col_bias = [&](int kxs) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kx = kxs + 8 * (i >> 2) + 4 * h + (i & 3);
        cb[i] = kx < gw && (lh <= 0 || (unsigned)(kx - qx + hwx) <= (unsigned)(2 * hwx)) ? 0.f : -INFINITY;
      }
    };
    if (pairs) col_bias(kx_lo);
    auto load_k = [&](int ky, int kxs, h8v (&kf)[2]) {
      const int tk = ky * gw + kxs + r;  // (< T + 32: rows past the grid row's end are other tokens or the zero rows; masked)
      const unsigned char* kr = sK + tk * 64;
      const int ksw = (tk >> 2) & 3;
#pragma unroll
      for (int s = 0; s < 2; ++s) kf[s] = *(const h8v*)(kr + (((2 * s + h) ^ ksw) << 4));
    };
    int nky = ky0, nkx = kx_lo;  // the tile whose K fragments are in flight
    h8v kf[2];
    load_k(nky, nkx, kf);
    for (int j = 0; j < ntl; ++j) {
      const int ky = nky, kxs = nkx;
      const int tk0 = ky * gw + kxs;
      nkx += 32;
      if (nkx >= kx_end) { nkx = kx_lo; ++nky; }
      // ---- S^T = K Q^T + mask: A = K rows (key r), 2 steps over d
      f16x sacc;
      const bool colmask = lh > 0 || kxs + 32 > gw;  // (wave-uniform: a full tile of a global mixer needs no mask)
      if (colmask && !pairs) col_bias(kxs);
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[e] = colmask ? cb[e] : 0.f;
      if (pairs && (ky < qy0 + 1 - hwy || ky > qy0 + hwy)) {  // (wave-uniform: the window's first / last row belongs to ONE of the tile's two query rows)
        const bool rok = (unsigned)(ky - qy + hwy) <= (unsigned)(2 * hwy);
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[e] = rok ? sacc[e] : -INFINITY;
      }
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0], sacc, 0, 0, 0);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[1], sacc, 0, 0, 0);
      // ---- this tile's V (transposed reads) and the next tile's K leave for LDS now, behind the matrix instructions; both are back
      // before the softmax below is through
      const unsigned vb = (unsigned)(size_t)(sV + (size_t)tk0 * 64 + tr_off);
      h4v v0, v1, v2, v3;
      asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\tds_read_b64_tr_b16 %2, %4 offset:1024\n\tds_read_b64_tr_b16 %3, %4 offset:1536"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(vb) : "memory");
      if (j + 1 < ntl) load_k(nky, nkx, kf);
      // ---- running maximum, probabilities
      float mt = fmaxf(sacc[0], sacc[1]);
#pragma unroll
      for (int i = 2; i < 16; ++i) mt = fmaxf(mt, sacc[i]);
      {
        float ma = mt, mb = mt;  // the other half's lanes hold the same queries: lanes 32.. of ma <-> lanes ..31 of mb
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(ma), "+v"(mb));
        mt = fmaxf(ma, mb);
      }
      const float mnew = fmaxf(mrun, mt);
      const float muse = mnew == -INFINITY ? 0.f : mnew;
      const float alpha = __builtin_amdgcn_exp2f(mrun - muse);
      mrun = mnew;
      float ls = 0.f;
      h8v pf[2];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f2v p2;
        p2[0] = __builtin_amdgcn_exp2f(sacc[i] - muse);
        p2[1] = __builtin_amdgcn_exp2f(sacc[i + 1] - muse);
        ls += p2[0];
        ls += p2[1];
        const h2v ph = __builtin_convertvector(p2, h2v);  // (v_cvt_pk_f16_f32)
        pf[i >> 3][i & 7] = ph[0];
        pf[i >> 3][(i & 7) + 1] = ph[1];
      }
      lrun = lrun * alpha + ls;
      if (__any(alpha != 1.0f)) {  // (wave-uniform: once the maxima have settled the sixteen multiplies are by one)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] *= alpha;
      }
      // ---- O^T += V^T P^T: A = V^T (row d = r; element j of step s = key 16 s + 8 (j >> 2) + 4 h + (j & 3)), B = the P registers as they are
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory");
      {
        const h8v vf0 = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const h8v vf1 = {v2[0], v2[1], v2[2], v2[3], v3[0], v3[1], v3[2], v3[3]};
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf0, pf[0], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf1, pf[1], o, 0, 0, 0);
      }
    }
    // ---- normalise and store: register i of lane (q, h) = d 8 (i >> 2) + 4 h + (i & 3)
    const float ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / ltot;
    if (qok) {
      _Float16* dst = out + ((long)n * T + tq) * D + hh * HD + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h4v t4;
#pragma unroll
        for (int e = 0; e < 4; ++e) t4[e] = (_Float16)__builtin_amdgcn_fmed3f(o[4 * g + e] * inv, -65504.0f, 65504.0f);
        *(h4v*)(dst + 8 * g) = t4;
      }
    }
  }
}

bool launch_attention(const void* qkv, void* out, int N, int T, int heads, int hd, float scale, int gh, int gw, int lh, int lw, bool half,
                      hipStream_t s, std::string& err) {
  if (hd != 32) { err = "attention: head dimension must be 32"; return false; }
  if (gh * gw != T) { err = "attention: token grid does not match the token count"; return false; }
  if (!half) {
    const long total = (long)N * heads * T;
    hipLaunchKernelGGL(attn_f_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, (const float*)qkv, (float*)out, N, T, heads, scale, gw, lh, lw);
    return true;
  }
  const size_t lds = (size_t)(T + 32) * 64 * 2;
  if (lds > 160 * 1024) { err = "attention: K and V of a line do not fit LDS"; return false; }
  const float sl = scale * 1.44269504088896341f;
  // a Global mixer has no use for the grid: its tokens as ONE row of T (T = 240: 8 x 8 tile pairs instead of 9 x 9 row-aligned
  // ones whose third column tile is half empty, T = 480: 15 x 15 instead of 18 x 18)
  if (lh <= 0) { gh = 1; gw = T; }
  const bool pairs = lh > 0 && lw / 2 <= 8;  // (the kernel's query tiling)
  const int nq = pairs ? ((gh + 1) / 2) * ((gw + 15) / 16) : gh * ((gw + 31) / 32);
  // heads in pairs (whole-line loads) where two heads' K and V fit beside another workgroup's: T = 240 (70 KB, 8 waves, two workgroups
  // per CU) and T = 480 (131 KB: one workgroup of 16 waves); a 960-token line (123 KB per head) keeps one head per workgroup
  static LdsAttrMemo memo8, memo4, memo8p, memo16p;
  static const bool pair_on = [] { const char* e = getenv("OCR_SRV_ATTN_PAIR"); return !(e && e[0] == '0'); }();
  if (pair_on && heads % 2 == 0 && 2 * lds <= 160 * 1024) {
    const size_t lds2 = 2 * lds;
    if (2 * lds2 <= 160 * 1024) {
      if (lds2 > 64 * 1024 && !raise_dynamic_lds((const void*)attn_h_kernel<8, 2>, (int)lds2, memo8p)) { err = "attention: dynamic LDS attribute refused"; return false; }
      hipLaunchKernelGGL((attn_h_kernel<8, 2>), dim3((unsigned)(N * heads / 2)), dim3(512), lds2, s, (const _Float16*)qkv, (_Float16*)out, T, heads, sl, gh, gw, lh, lw);
    } else {
      if (!raise_dynamic_lds((const void*)attn_h_kernel<16, 2>, (int)lds2, memo16p)) { err = "attention: dynamic LDS attribute refused"; return false; }
      hipLaunchKernelGGL((attn_h_kernel<16, 2>), dim3((unsigned)(N * heads / 2)), dim3(1024), lds2, s, (const _Float16*)qkv, (_Float16*)out, T, heads, sl, gh, gw, lh, lw);
    }
    return true;
  }
  if (nq >= 12) {
    if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)attn_h_kernel<8, 1>, (int)lds, memo8)) { err = "attention: dynamic LDS attribute refused"; return false; }
    hipLaunchKernelGGL((attn_h_kernel<8, 1>), dim3((unsigned)(N * heads)), dim3(512), lds, s, (const _Float16*)qkv, (_Float16*)out, T, heads, sl, gh, gw, lh, lw);
  } else {
    if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)attn_h_kernel<4, 1>, (int)lds, memo4)) { err = "attention: dynamic LDS attribute refused"; return false; }
    hipLaunchKernelGGL((attn_h_kernel<4, 1>), dim3((unsigned)(N * heads)), dim3(256), lds, s, (const _Float16*)qkv, (_Float16*)out, T, heads, sl, gh, gw, lh, lw);
  }
  return true;
}

}  // namespace srv
}  // namespace ocr
