// Network kernels for gfx950 (MI355X).  f32 storage, f32 MFMA (`v_mfma_f32_32x32x2_f32`), every
// contraction a single ascending-k fmaf chain — see ocr_common.h for the layout argument and
// DESIGN.md §4 for the arithmetic contract these kernels implement bit for bit.
//
// Replaces the `Predictor::Run()` calls of the reference
// (/root/reference/src/ocr_det.cpp:122, ocr_cls.cpp:74, ocr_rec.cpp:81).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "conv_device.h"
#include "dw_epilogue.h"
#include "kernels_net.h"

// This file is compiled TWICE (build.py).  As it stands: the f32 contract - f32 tensors, v_mfma_f32_32x32x2_f32.  With
// -DOCR_TU_H16 (object kernels_net_h16.o): precision "fp16" - the C8I activation tensors are stored as f16 (ld4 / st4 of
// conv_device.h with kH16), the matrix products are v_mfma_f32_32x32x8_f16 on those halfs and f16 weight fragments, every
// accumulation, reduction and epilogue stays f32.  The twin's kernels live in the inline namespace ocr::h16, its launchers
// are `<name>_h16` (kernels_net.h); a launcher of this build hands a call with the f16 flag set to its twin.
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

// =====================================================================================
// Dense conv / linear / 2x2-s2 deconv as implicit GEMM on the f32 matrix cores.
//   rows  m = (n, y, x) output pixels (stride-1 conv, zero padding), 32 per wave
//   cols  j = output channels, NT tiles of 32 per wave
//   k     = ((ky*KW)+kx)*Cin + ci ascending: one v_mfma_f32_32x32x2_f32 consumes (2s, 2s+1)
// A comes straight from HBM/L2 as one 16-byte load per lane per 8 input channels (C8I layout),
// B from the host-built fragment image (one 16-byte load per lane per n-tile per 8 channels).
// No LDS, no barriers: four independent waves per workgroup.
// =====================================================================================
template <int NT, int MODE, bool TAP1, bool GATE = false, bool HALF = kH16>
__global__ void __launch_bounds__(256, (NT <= 3 ? 4 : 3)) conv_mfma_kernel(const ConvArgs a, const Epilogue ep) {
  using WV = typename WFrag<HALF>::T;  // a lane's weight fragment of one octet: float4, or four halfs (precision "fp16")
  using AV = typename WFrag<HALF>::T;  // its pixel operand: float4 from an f32 tensor, or four halfs straight from an f16 tensor
  constexpr int ES = HALF ? 2 : 4;     // bytes per stored activation (precision "fp16" stores the C8I tensors as f16)
  CONV_PROBE(0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 31, h = lane >> 5;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);  // logical block: N-group fastest, then M-tile
  const unsigned groups = (unsigned)a.NTtot / NT;
  const long m0 = ((long)(lb / groups) * 4 + wave) * 32;
  const int nt0 = (int)(lb % groups) * NT;
  __shared__ float s_par[OCR_MAX_EP * 2 * NT * 32];
  __shared__ float s_xp[MODE == OUT_C8I ? 4 * OCR_XP_FLOATS : 1];  // the waves' store tiles (conv_finish, conv_device.h)
  conv_stage_params<NT>(a, ep, nt0, s_par);
  if (m0 >= a.M) return;
  const long m = m0 + p;
  const bool mvalid = m < a.M;

  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // K loop flattened over (tap, 8-channel group), software-pipelined over two register sets: the loads
  // of step kk+1 are in flight while the 4*NT MFMAs of step kk issue.  The loop body is straight-line
  // (no branch between a load and the MFMAs that overlap it, the step past the end re-reads the last
  // one), so the waitcnt the compiler places before each MFMA group counts the loads issued after the
  // ones it needs instead of draining everything.
  // What a step costs besides its MFMAs is not hidden by the other waves of the SIMD (probe with all
  // memory traffic removed: the generic walk below holds the kernel at 75% of the MFMA rate), so the
  // single-tap case - 1x1 conv, linear, deconv: most of the FLOPs - gets a walk of a few instructions.
  const WV* __restrict__ wf = (const WV*)a.wfrag;
  const int KK = a.KH * a.KW * a.C8;
  const WV* p_w = wf + (long)nt0 * 64 + lane;
  const long wstride = (long)a.NTtot * 64;
  const char* zpage = (const char*)(a.zeros + 4 * h);  // (zero bits are 0.0 in either format)
  // ---- single tap: step j reads the lane's row at channel 8j and fragment block j
  const char* xrow = mvalid ? (const char*)a.in + (m * a.Cs_in + 4 * h) * ES : zpage;
  // GATE: the SE gate of the lane's image (f32), same four channels as the pixel load (single-tap only)
  const float* grow = a.zeros + 4 * h;
  if constexpr (GATE) {
    if (mvalid) {
      const int gn = a.rin.w ? rag_sample_of_pixel(a.rin, a.N, a.H, m, (long)(lb / groups) * 128) : (int)((unsigned)m / (unsigned)a.gate_hw);
      grow = a.gate + (long)gn * a.Cs_in + 4 * h;
    }
  }
  int p_step = 0;
  // ---- general: incremental (tap, c8) walk with selects
  int p_c8 = 0, p_ky = 0, p_kx = 0;
  int n = 0, y = 0, x = 0;
  int lw = a.W, lh = a.H;  // width / height of the lane's sample (ragged batch: its own)
  const char* lane_base = nullptr;
  long tap_off = 0;  // + c8*8, elements
  if constexpr (!TAP1) {
    if (a.rin.w) {  // stride-1 "same" convs only (host): the output level is the input level
      rag_decompose(a.rin, a.N, a.H, mvalid ? m : a.M - 1, (long)(lb / groups) * 128, n, y, x, lw);
      lh = rag_h(a.rin, n, a.H);
      lane_base = (const char*)a.in + (4 * h + (rag_pix0(a.rin, n, a.H) + (long)(y - a.PH) * lw + (x - a.PW)) * a.Cs_in) * ES;
    } else {
      decompose(mvalid ? m : a.M - 1, a.OH * a.OW, a.OW, n, y, x);
      // this lane's pixel at tap (0,0), channel 4h; a tap adds the uniform offset (ky*W + kx)*Cs_in
      lane_base = (const char*)a.in + (4 * h + (((long)n * a.H + (y - a.PH)) * a.W + (x - a.PW)) * a.Cs_in) * ES;
    }
  }
  auto load_step = [&](AV& av, WV (&bv)[NT], float4& gv) {
    if constexpr (GATE) gv = *(const float4*)(grow + p_step * 8);
    if constexpr (TAP1) {
#ifdef OCR_PROBE_NOX
      { av = AV{}; av.x = (decltype(av.x))(size_t)xrow; }
#else
      av = *(const AV*)(xrow + (long)p_step * 8 * ES);
#endif
    } else {
      const int iy = y - a.PH + p_ky, ix = x - a.PW + p_kx;
      const bool valid = mvalid && (unsigned)iy < (unsigned)lh && (unsigned)ix < (unsigned)lw;
      // padding / out-of-range rows read a zero page: no select ever touches a loaded value
      const char* src = valid ? lane_base + tap_off * ES : zpage;
      av = *(const AV*)src;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#ifdef OCR_PROBE_NOW
      { bv[t] = WV{}; bv[t].x = (decltype(bv[t].x))(size_t)p_w; }
#else
      bv[t] = p_w[t * 64];  // the fragment image is padded to whole NT groups
#endif
    const bool more = p_step + 1 < KK;                 // past the end: stay on the last step (loaded, unused)
    p_step += more;
    p_w += more ? wstride : 0;
    if constexpr (!TAP1) {
      p_c8 += more;
      const bool wc = p_c8 == a.C8;
      p_c8 = wc ? 0 : p_c8;
      p_kx += wc;
      const bool wx = p_kx == a.KW;
      p_kx = wx ? 0 : p_kx;
      p_ky += wx;
      tap_off = ((long)p_ky * lw + p_kx) * a.Cs_in + p_c8 * 8;
    }
  };
  // One VALU instruction between the groups of NT MFMAs (the gate multiply of the next group, else a v_nop): with
  // nothing but MFMAs in the step the waves of a SIMD take turns badly.  tools/micro/conv_time.hip, 983040 x 480 -> 480
  // alone: 108.3 TFLOP/s without, 128.3 with (s_nop: no change; a VALU op after EVERY MFMA: 120); 240 -> 240: 92.6 ->
  // 105.0; 480 -> 120: 94.4 -> 110.2; NT = 1 (every MFMA followed by one) loses 4 %, so only for NT >= 2.
  auto mfma_step = [&](const AV& av0, const WV (&bv)[NT], const float4& gv) {
    if constexpr (HALF) {  // one f16 matrix instruction per octet and column tile; the operand comes as stored
      ocr_h4 ah = ocr_as_h4(av0);
      // x * gate as two packed f16 multiplies on the gate rounded to f16 (v_pk_mul_f16): the product is rounded to f16 either
      // way - via f32 it cost ten VALU instructions per octet and pixel tile, more than the octet's matrix instructions
      if constexpr (GATE) ah = ah * ocr_to_h4(gv);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(ocr_as_h4(bv[t]), ah, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#define OCR_C_SWEEP(C)                                                                                                    \
  {                                                                                                                       \
    float avc = av0.C;                                                                                                    \
    if constexpr (GATE) avc = av0.C * gv.C; /* x * gate: one rounding, as ew_kernel did */                                \
    else if constexpr (NT >= 2) asm volatile("v_nop");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].C, avc, acc[t], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                                    \
  }
    OCR_C_SWEEP(x) OCR_C_SWEEP(y) OCR_C_SWEEP(z) OCR_C_SWEEP(w)
#undef OCR_C_SWEEP
    }
  };
  AV a0, a1;
  WV b0[NT], b1[NT];
  float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
  load_step(a0, b0, g0);
  CONV_PROBE(1);
  int kk = 0;
  for (; kk + 2 <= KK; kk += 2) {
    load_step(a1, b1, g1);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMAs they overlap
    mfma_step(a0, b0, g0);
    __builtin_amdgcn_sched_barrier(0);
    load_step(a0, b0, g0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(a1, b1, g1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kk < KK) mfma_step(a0, b0, g0);
  CONV_PROBE(2);

  // ---- epilogue: lane owns output column j (one channel), 16 rows ----
  conv_finish<NT, MODE, HALF>(a, ep, acc, nt0, m, h, s_par, NT * 32, MODE == OUT_C8I ? s_xp + wave * OCR_XP_FLOATS : nullptr);
  CONV_PROBE(3);
}

// The single-tap case (1x1 conv / linear with a C8I output) with MT pixel tiles per wave: a wave owns MT consecutive
// 32-pixel tiles x NT column tiles, so a weight fragment fetched from L2 feeds MT MFMAs instead of one and a workgroup
// stages its parameters once per MT*128 pixels.  tools/micro/conv_time.hip (rec op 30's shape, 983040 x 480 -> 480,
// alone): MT x NT = 1x3 (the kernel above) 99.8 TFLOP/s gated / 108.6 plain, 1x5 110.5 / 115.6 - the direct kernel is
// bound by the bytes it pulls through L1 per MFMA (1x3: 5 KB per 12 MFMAs; 2x3: 7 KB per 24), not by occupancy
// (3 or 4 waves per SIMD measure the same).  Every output's chain is the one above: bit-identical.
#ifdef OCR_CONV_CLKRATE  // development probe (tools/micro/conv_time.hip): shader cycles and 100 MHz ticks over the workgroups' lives
__device__ unsigned long long ocr_conv_clkrate[2];
#endif
template <int NT, int MT, bool GATE, bool HALF = kH16>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_mfma_mt_kernel(const ConvArgs a, const Epilogue ep) {
  using WV = typename WFrag<HALF>::T;
  using AV = typename WFrag<HALF>::T;
  constexpr int ES = HALF ? 2 : 4;
#ifdef OCR_CONV_CLKRATE
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 31, h = lane >> 5;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);  // logical block: N-group fastest, then M-tile
  const unsigned groups = (unsigned)a.NTtot / NT;
  const long mblk = (long)(lb / groups) * (128 * MT);
  const long m0 = mblk + (long)wave * (32 * MT);
  const int nt0 = (int)(lb % groups) * NT;
  __shared__ float s_par[OCR_MAX_EP * 2 * NT * 32];
  conv_stage_params<NT>(a, ep, nt0, s_par);
  if (m0 >= a.M) return;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

  const WV* __restrict__ wf = (const WV*)a.wfrag;
  const int KK = a.C8;
  const WV* p_w = wf + (long)nt0 * 64 + lane;
  const long wstride = (long)a.NTtot * 64;
  const char* zpage = (const char*)(a.zeros + 4 * h);
  const char* xrow[MT];
  const float* grow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const long m = m0 + 32 * i + p;
    const bool mvalid = m < a.M;
    xrow[i] = mvalid ? (const char*)a.in + (m * a.Cs_in + 4 * h) * ES : zpage;
    grow[i] = a.zeros + 4 * h;
    if constexpr (GATE) {
      if (mvalid) {
        const int gn = a.rin.w ? rag_sample_of_pixel(a.rin, a.N, a.H, m, mblk) : (int)((unsigned)m / (unsigned)a.gate_hw);
        grow[i] = a.gate + (long)gn * a.Cs_in + 4 * h;
      }
    }
  }
  int p_step = 0;
  struct Set { AV av[MT]; float4 gv[MT]; WV bv[NT]; };
  auto load_step = [&](Set& s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (GATE) s.gv[i] = *(const float4*)(grow[i] + p_step * 8);
      s.av[i] = *(const AV*)(xrow[i] + (long)p_step * 8 * ES);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) s.bv[t] = p_w[t * 64];  // the fragment image is padded to whole NT groups
    const bool more = p_step + 1 < KK;  // past the end: stay on the last step (loaded, unused)
    p_step += more;
    p_w += more ? wstride : 0;
  };
  auto mfma_step = [&](const Set& s) __attribute__((always_inline)) {
    if constexpr (HALF) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        ocr_h4 ah = ocr_as_h4(s.av[i]);
        if constexpr (GATE) ah = ah * ocr_to_h4(s.gv[i]);  // (packed f16 multiplies: conv_mfma_kernel)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x8f16(ocr_as_h4(s.bv[t]), ah, acc[i][t], 0, 0, 0);
      }
    } else {
    float4 av[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      av[i] = s.av[i];
      if constexpr (GATE) { av[i].x = s.av[i].x * s.gv[i].x; av[i].y = s.av[i].y * s.gv[i].y; av[i].z = s.av[i].z * s.gv[i].z; av[i].w = s.av[i].w * s.gv[i].w; }  // x * gate: one rounding
    }
    // One VALU instruction after each pixel tile's NT MFMAs: with nothing but MFMAs in the step the two waves of a SIMD
    // take turns badly (conv_time, 983040 x 480 -> 480: 117.8 TFLOP/s without, 129.7 with a v_nop per NT MFMAs, 120.3 with
    // one per MFMA, no change with s_nop); the gated kernel has its multiplies there (124.5; hoisted in front of the
    // MFMAs 108.6), and more of them do not help it.
#define OCR_MT_SWEEP(C)                                                                                      \
  _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                           \
    _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                           \
      acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(s.bv[t].C, av[i].C, acc[i][t], 0, 0, 0);              \
    if constexpr (!GATE) { __builtin_amdgcn_sched_barrier(0); asm volatile("v_nop"); __builtin_amdgcn_sched_barrier(0); } \
  }
    OCR_MT_SWEEP(x) OCR_MT_SWEEP(y) OCR_MT_SWEEP(z) OCR_MT_SWEEP(w)
#undef OCR_MT_SWEEP
    }
  };
  Set s0, s1;
  if constexpr (!GATE) {
#pragma unroll
    for (int i = 0; i < MT; ++i) s0.gv[i] = s1.gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  load_step(s0);
  int kk = 0;
  for (; kk + 2 <= KK; kk += 2) {
    load_step(s1);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMAs they overlap
    mfma_step(s0);
    __builtin_amdgcn_sched_barrier(0);
    load_step(s0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kk < KK) mfma_step(s0);
  // (spelled out: left to "#pragma unroll" the compiler keeps the loop for the larger NT and indexes acc through scratch)
  conv_finish<NT, OUT_C8I, HALF>(a, ep, acc[0], nt0, m0 + p, h, s_par);
  if constexpr (MT > 1) conv_finish<NT, OUT_C8I, HALF>(a, ep, acc[1], nt0, m0 + 32 + p, h, s_par);
  if constexpr (MT > 2) conv_finish<NT, OUT_C8I, HALF>(a, ep, acc[2], nt0, m0 + 64 + p, h, s_par);
#ifdef OCR_CONV_CLKRATE
  if (threadIdx.x == 0) {
    atomicAdd(&ocr_conv_clkrate[0], __builtin_readcyclecounter() - clk_c0);
    atomicAdd(&ocr_conv_clkrate[1], wall_clock64() - clk_r0);
  }
#endif
}

#ifdef OCR_TU_H16
// The same kernel on gfx950's v_mfma_f32_32x32x16_f16 (precision "fp16", an even number of octets): one matrix instruction
// consumes TWO octets at twice the rate of 32x32x8.  A lane's operand is a whole octet - lane (p, h) of step s holds the eight
// physical channels of octet 2s + h: ONE 16-byte load from the f16 tensor - and the weights come from the fragment image in
// that arrangement (`frag16x:`, net.hip: [octet pair][column tile][64 lanes][8 halfs]).  f32 accumulation as before; the
// products of a step are summed in the instruction's own order, which is not the 32x32x8 kernel's (same tolerances hold).
typedef _Float16 ocr_h8 __attribute__((ext_vector_type(8)));
typedef float ocr_f8v __attribute__((ext_vector_type(8)));
template <int NT, int MT, bool GATE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) conv_mfma_mt16_kernel(const ConvArgs a, const Epilogue ep) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 31, h = lane >> 5;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);  // logical block: N-group fastest, then M-tile
  const unsigned groups = (unsigned)a.NTtot / NT;
  const long mblk = (long)(lb / groups) * (128 * MT);
  const long m0 = mblk + (long)wave * (32 * MT);
  const int nt0 = (int)(lb % groups) * NT;
  __shared__ float s_par[OCR_MAX_EP * 2 * NT * 32];
  conv_stage_params<NT>(a, ep, nt0, s_par);
  if (m0 >= a.M) return;

  floatx16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.0f;

  const uint4* __restrict__ wf = (const uint4*)a.wfrag_x16;
  const int KK = a.C8 >> 1;  // octet pairs
  const uint4* p_w = wf + (long)nt0 * 64 + lane;
  const long wstride = (long)a.NTtot * 64;
  const char* zpage = (const char*)a.zeros;
  const char* xrow[MT];
  const float* grow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const long m = m0 + 32 * i + p;
    const bool mvalid = m < a.M;
    xrow[i] = mvalid ? (const char*)a.in + (m * a.Cs_in + 8 * h) * 2 : zpage;
    grow[i] = a.zeros;
    if constexpr (GATE) {
      if (mvalid) {
        const int gn = a.rin.w ? rag_sample_of_pixel(a.rin, a.N, a.H, m, mblk) : (int)((unsigned)m / (unsigned)a.gate_hw);
        grow[i] = a.gate + (long)gn * a.Cs_in + 8 * h;
      }
    }
  }
  int p_step = 0;
  struct Set { uint4 av[MT]; float4 glo[MT], ghi[MT]; uint4 bv[NT]; };
  auto load_step = [&](Set& s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if constexpr (GATE) { s.glo[i] = *(const float4*)(grow[i] + p_step * 16); s.ghi[i] = *(const float4*)(grow[i] + p_step * 16 + 4); }
      s.av[i] = *(const uint4*)(xrow[i] + (long)p_step * 32);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) s.bv[t] = p_w[t * 64];
    const bool more = p_step + 1 < KK;  // past the end: stay on the last step (loaded, unused)
    p_step += more;
    p_w += more ? wstride : 0;
  };
  auto mfma_step = [&](const Set& s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      ocr_h8 ah = __builtin_bit_cast(ocr_h8, s.av[i]);
      if constexpr (GATE) {  // packed f16 multiplies by the gate rounded to f16 (conv_mfma_kernel)
        const ocr_f8v g = {s.glo[i].x, s.glo[i].y, s.glo[i].z, s.glo[i].w, s.ghi[i].x, s.ghi[i].y, s.ghi[i].z, s.ghi[i].w};
        ah = ah * __builtin_convertvector(g, ocr_h8);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ocr_h8, s.bv[t]), ah, acc[i][t], 0, 0, 0);
    }
  };
  Set s0, s1;
  load_step(s0);
  int kk = 0;
  for (; kk + 2 <= KK; kk += 2) {
    load_step(s1);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMAs they overlap
    mfma_step(s0);
    __builtin_amdgcn_sched_barrier(0);
    load_step(s0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kk < KK) mfma_step(s0);
  conv_finish<NT, OUT_C8I, true>(a, ep, acc[0], nt0, m0 + p, h, s_par);
  if constexpr (MT > 1) conv_finish<NT, OUT_C8I, true>(a, ep, acc[1], nt0, m0 + 32 + p, h, s_par);
}
#endif  // OCR_TU_H16

// Two pixel tiles per wave (the kernel above) for the shapes it was measured on: single tap, C8I output, nt = 3 or 4.
bool OCR_L(launch_conv_mfma_mt2)(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s) {
  OCR_H16_TWIN(a.half, launch_conv_mfma_mt2_h16(a, ep, nt, s))
  const bool tap1 = a.KH == 1 && a.KW == 1 && a.PH == 0 && a.PW == 0 && a.OH == a.H && a.OW == a.W;
  if (!tap1 || a.out_mode != OUT_C8I || a.NTtot % nt) return false;
  // (round 5: the K loop's loads TWO steps ahead through three register sets - 175 / 198 registers - measured 1-2 % slower
  // on all three shapes, conv_time: 3.50 -> 3.53, gated 3.67 -> 3.74, 240 -> 480 gated 2.11 -> 2.14 ms: load latency is not
  // what the 20-35 % below the matrix peak are made of.)
  // (round 5: the CTC head - linear 120 -> 6625 with the softmax partials, 208 column tiles - on this kernel: 1.42 ms against
  // 1.31 on conv_mfma_kernel, same box: its epilogue's exps want the four waves per SIMD of the one-tile kernel.  Not kept.)
  // (round 5: THREE pixel tiles per wave - a fragment feeding three MFMAs, 216 registers - measured 1.7 % faster alone
  // (conv_time, 983 040 x 480 -> 480: 3.524 -> 3.464 ms) and 1-3 % SLOWER in the step on the same box, alternating runs
  // (op 30: 3.97 -> 4.08 ms): not kept.  Five column tiles x two pixel tiles spill 13 registers.)
  dim3 grid((unsigned)(((a.M + 255) / 256) * (a.NTtot / nt)));
#ifdef OCR_TU_H16
  // an even number of octets and the paired fragment image at hand: the 32x32x16 form (OCR_MFMA_X16=0: the 32x32x8 one, A/B)
  const bool x16 = a.wfrag_x16 && (a.C8 & 1) == 0 && rt_options().mfma_x16;
#define OCR_MT_LAUNCH(NT_, GATE_)                                                                            \
  {                                                                                                          \
    if (x16) hipLaunchKernelGGL((conv_mfma_mt16_kernel<NT_, 2, GATE_>), grid, dim3(256), 0, s, a, ep);       \
    else hipLaunchKernelGGL((conv_mfma_mt_kernel<NT_, 2, GATE_>), grid, dim3(256), 0, s, a, ep);             \
  }
#else
#define OCR_MT_LAUNCH(NT_, GATE_) { hipLaunchKernelGGL((conv_mfma_mt_kernel<NT_, 2, GATE_>), grid, dim3(256), 0, s, a, ep); }
#endif
  if (nt == 3) {
    if (a.gate) OCR_MT_LAUNCH(3, true) else OCR_MT_LAUNCH(3, false)
  } else if (nt == 4) {
    if (a.gate) OCR_MT_LAUNCH(4, true) else OCR_MT_LAUNCH(4, false)
  } else {
    return false;
  }
#undef OCR_MT_LAUNCH
  return true;
}

bool OCR_L(launch_conv_mfma)(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s) {
  OCR_H16_TWIN(a.half, launch_conv_mfma_h16(a, ep, nt, s))
  constexpr bool HALF = kH16;
  dim3 grid((unsigned)(((a.M + 127) / 128) * (a.NTtot / nt)));
#define OCR_LAUNCH_MODE(MODE, TAP1)                                                                             \
  switch (nt) {                                                                                                 \
    case 1: hipLaunchKernelGGL((conv_mfma_kernel<1, MODE, TAP1, false, HALF>), grid, dim3(256), 0, s, a, ep); break;         \
    case 2: hipLaunchKernelGGL((conv_mfma_kernel<2, MODE, TAP1, false, HALF>), grid, dim3(256), 0, s, a, ep); break;         \
    case 3: hipLaunchKernelGGL((conv_mfma_kernel<3, MODE, TAP1, false, HALF>), grid, dim3(256), 0, s, a, ep); break;         \
    default: hipLaunchKernelGGL((conv_mfma_kernel<4, MODE, TAP1, false, HALF>), grid, dim3(256), 0, s, a, ep); break;        \
  }
  const bool tap1 = a.KH == 1 && a.KW == 1 && a.PH == 0 && a.PW == 0 && a.OH == a.H && a.OW == a.W;
  if (a.gate) {
    if (!(tap1 && a.out_mode == OUT_C8I)) return false;  // a gated input needs a 1x1 conv with a C8I output
    switch (nt) {
      case 1: hipLaunchKernelGGL((conv_mfma_kernel<1, OUT_C8I, true, true, HALF>), grid, dim3(256), 0, s, a, ep); break;
      case 2: hipLaunchKernelGGL((conv_mfma_kernel<2, OUT_C8I, true, true, HALF>), grid, dim3(256), 0, s, a, ep); break;
      case 3: hipLaunchKernelGGL((conv_mfma_kernel<3, OUT_C8I, true, true, HALF>), grid, dim3(256), 0, s, a, ep); break;
      default: hipLaunchKernelGGL((conv_mfma_kernel<4, OUT_C8I, true, true, HALF>), grid, dim3(256), 0, s, a, ep); break;
    }
    return true;
  }
  if (a.out_mode == OUT_HEAD && tap1) { OCR_LAUNCH_MODE(OUT_HEAD, true) }
  else if (a.out_mode == OUT_PLAIN && tap1) { OCR_LAUNCH_MODE(OUT_PLAIN, true) }
  else if (a.out_mode == OUT_DECONV && tap1) { OCR_LAUNCH_MODE(OUT_DECONV, true) }
  else if (a.out_mode == OUT_C8I && tap1) { OCR_LAUNCH_MODE(OUT_C8I, true) }
  else if (a.out_mode == OUT_C8I) { OCR_LAUNCH_MODE(OUT_C8I, false) }
  else return false;  // a multi-tap conv with a plain / deconv output is not instantiated
#undef OCR_LAUNCH_MODE
  return true;
}


#ifndef OCR_TU_H16  // (f32 contract only: precision "fp16" keeps the direct and the LDS-tile kernels)
// =====================================================================================
// The same implicit GEMM with operands staged through LDS (the default for K >= 64).
//   workgroup tile : 128 output pixels (4 waves x 32) x NT*32 output channels
//   K chunk        : BK input channels of one tap; A tile 128 x BK (rows padded by 16 B: the
//                    16-lane groups of ds_read_b128 then hit 16 distinct 4-bank slots),
//                    B tile = the BK/8 x NT fragment KBs of the chunk, shared by the 4 waves
//   global -> LDS  : every thread moves 16-byte pieces; a pixel's BK channels are one contiguous
//                    BK*4-byte run (a full 128-B line for BK = 32), B pieces are linear
//   pipeline       : chunk c+1 travels global -> registers while chunk c is multiplied out of LDS,
//                    then registers -> the other LDS buffer; one barrier per chunk
// vs conv_mfma_kernel (fragment-shaped loads straight from L1/L2): B leaves L2 once per workgroup
// instead of once per wave, A arrives in whole lines, and the matrix pipe is fed from LDS
// (rocprof r1e: the direct kernel kept the MFMA pipe 44-54 % busy, TA-bound for NT = 1).
// The MFMA sequence per output is unchanged: same ascending-k chain, bit-identical results.
// =====================================================================================
template <int NT, int BK>
__global__ void __launch_bounds__(256, 2) conv_lds_kernel(const ConvArgs a, const Epilogue ep) {
  constexpr int A_STRIDE = BK + 4;            // floats per staged row
  constexpr int A_TILE = 128 * A_STRIDE;      // floats
  constexpr int B_TILE = (BK / 8) * NT * 256; // floats
  constexpr int A_PER_THR = BK / 8;           // float4 pieces of A per thread per chunk
  constexpr int SEG = BK / 4;                 // 16-B pieces per row
  constexpr int B_PIECES = (BK / 8) * NT * 64;  // float4 pieces of B per chunk
  constexpr int B_PER_THR = (B_PIECES + 255) / 256;
  __shared__ float4 smem4[2 * (A_TILE + B_TILE) / 4];
  float* smem = (float*)smem4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned groups = (unsigned)a.NTtot / NT;
  const long m_base = (long)(lb / groups) * 128;
  const int nt0 = (int)(lb % groups) * NT;
  const int hw = a.OH * a.OW;
  const long m0 = m_base + wave * 32;

  // per-row addressing state of the 128 staged rows, computed once into LDS (a per-thread array of
  // it ends up in scratch): {first pixel of the row's image or -1, the image's width | height << 16, y, x}
  __shared__ int4 rowinfo[128];
  if (tid < 128) {
    const long m = m_base + tid;
    int4 ri = make_int4(-1, 1 | (1 << 16), 0, 0);
    if (m < a.M) {
      int n, y, x, w = a.W, hh = a.H;
      long pix0;
      if (a.rin.w) {  // ragged batch (stride-1 "same" convs only: output level = input level)
        rag_decompose(a.rin, a.N, a.H, m, m_base, n, y, x, w);
        hh = rag_h(a.rin, n, a.H);
        pix0 = rag_pix0(a.rin, n, a.H);
      } else {
        decompose(m, hw, a.OW, n, y, x);
        pix0 = (long)n * a.H * a.W;
      }
      ri = make_int4((int)pix0, w | (hh << 16), y, x);  // (pixel counts of a launch stay below 2^31, sides below 2^15: host check)
    }
    rowinfo[tid] = ri;
  }
  __shared__ float s_par[OCR_MAX_EP * 2 * NT * 32];
  conv_stage_params<NT>(a, ep, nt0, s_par);  // ends with the barrier that also publishes rowinfo
  const int cpt = a.Cs_in / BK;  // chunks per tap
  const int nchunks = a.KH * a.KW * cpt;
  const float4* __restrict__ wf = (const float4*)a.wfrag;

  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  // One loop body, one copy of each phase (c = -1 is the pipeline fill): keeps the staging
  // registers ra/rb in VGPRs (two call sites of a lambda pushed them to scratch).
  for (int c = -1; c < nchunks; ++c) {
    // staging registers as named scalars (arrays of them were demoted to scratch by the compiler)
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool more = c + 1 < nchunks;
    if (more) {  // ---- global -> registers for chunk c+1
      const int cn = c + 1;
      const int tap = cn / cpt, cc = cn - tap * cpt;
      const int ky = tap / a.KW, kx = tap - ky * a.KW;
      auto load_a = [&](int i) __attribute__((always_inline)) -> float4 {
        const int idx = tid + i * 256, row = idx / SEG, seg = idx - row * SEG;
        const int4 ri = rowinfo[row];
        const int iy = ri.z - a.PH + ky, ix = ri.w - a.PW + kx;
        // branch-free validity and address (no short-circuit control flow around the load)
        const int rw = ri.y & 0xffff, rh = ri.y >> 16;
        const bool v = (ri.x >= 0) & (iy >= 0) & (iy < rh) & (ix >= 0) & (ix < rw);
        // (a folded same-resolution concat, ConvArgs::cat_*: the chunk's channels live in source cc * BK / cat_cs - uniform)
        const float* abase = a.in;
        int cs_row = a.Cs_in, coff = cc * BK;
        if (a.cat_n) {
          const int j = coff / a.cat_cs;
          abase = j == 0 ? a.cat_src[0] : j == 1 ? a.cat_src[1] : j == 2 ? a.cat_src[2] : a.cat_src[3];
          coff -= j * a.cat_cs;
          cs_row = a.cat_cs;
        }
        const long off = ((long)ri.x + (long)(v ? iy : 0) * rw + (v ? ix : 0)) * cs_row + coff + seg * 4;
        const float* src = v ? abase + off : a.zeros;
        return *(const float4*)src;
      };
      ra0 = load_a(0);
      if constexpr (A_PER_THR > 1) ra1 = load_a(1);
      if constexpr (A_PER_THR > 2) { ra2 = load_a(2); ra3 = load_a(3); }
      const float4* wsrc = wf + ((long)cn * (BK / 8) * a.NTtot + nt0) * 64;
      auto load_b = [&](int i) __attribute__((always_inline)) -> float4 {
        const int j = tid + i * 256;
        const int jj = j < B_PIECES ? j : 0;
        const int sstep = jj / (NT * 64), rem = jj - sstep * (NT * 64);
        return wsrc[(long)sstep * a.NTtot * 64 + rem];
      };
      rb0 = load_b(0);
      if constexpr (B_PER_THR > 1) rb1 = load_b(1);
      if constexpr (B_PER_THR > 2) rb2 = load_b(2);
      if constexpr (B_PER_THR > 3) rb3 = load_b(3);
    }
    if (c >= 0) {  // ---- multiply chunk c out of LDS
      const int buf = c & 1;
      const float* sA = smem + buf * (A_TILE + B_TILE) + (wave * 32 + p) * A_STRIDE + 4 * h;
      const float4* sB = (const float4*)(smem + buf * (A_TILE + B_TILE) + A_TILE) + lane;
#pragma unroll
      for (int sstep = 0; sstep < BK / 8; ++sstep) {
        const float4 av = *(const float4*)(sA + sstep * 8);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 bv = sB[(sstep * NT + t) * 64];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.x, av.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.y, av.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.z, av.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv.w, av.w, acc[t], 0, 0, 0);
        }
      }
    }
    if (more) {  // ---- registers -> the other LDS buffer
      float* sAw = smem + ((c + 1) & 1) * (A_TILE + B_TILE);
      float4* sBw = (float4*)(sAw + A_TILE);
      auto store_a = [&](int i, const float4& v) __attribute__((always_inline)) {
        const int idx = tid + i * 256, row = idx / SEG, seg = idx - row * SEG;
        *(float4*)(sAw + row * A_STRIDE + seg * 4) = v;
      };
      store_a(0, ra0);
      if constexpr (A_PER_THR > 1) store_a(1, ra1);
      if constexpr (A_PER_THR > 2) { store_a(2, ra2); store_a(3, ra3); }
      auto store_b = [&](int i, const float4& v) __attribute__((always_inline)) {
        const int j = tid + i * 256;
        if (j < B_PIECES) sBw[j] = v;
      };
      store_b(0, rb0);
      if constexpr (B_PER_THR > 1) store_b(1, rb1);
      if constexpr (B_PER_THR > 2) store_b(2, rb2);
      if constexpr (B_PER_THR > 3) store_b(3, rb3);
    }
    __syncthreads();
  }
  const long m = m0 + p;
  if (m >= a.M) return;
  conv_finish<NT, OUT_C8I>(a, ep, acc, nt0, m, h, s_par);
}

#endif  // OCR_TU_H16
// =====================================================================================
// 3x3 stride-1 pad-1 conv with the whole input tile resident in LDS (the DB neck/head 96 -> 24 convs:
// 5.56 of det's 10.36 GFLOP).  conv_lds_kernel above re-stages the 128-pixel A tile for every tap and
// channel chunk (27 chunks, a barrier each, the input fetched 9 times through L1/L2: rocprof 2.8x the
// algorithmic HBM bytes, MFMA pipe 54 % busy).  Here a workgroup owns an 8 x 16 pixel tile of one image:
//   fill   : the 10 x 18 halo region x all Cin channels goes global -> LDS once (coalesced 16-byte pieces,
//            zero outside the image), rows padded by 16 B so the 16-lane groups of ds_read_b128 are
//            conflict-free; ONE barrier
//   K loop : 9 taps x C8 octets straight out of LDS, no further barrier; a wave = 2 tile rows x 16 columns
//            (32 pixels) x NT column tiles; weight fragments come from L1/L2 a whole tap ahead (two
//            register sets - occupancy is LDS-bound at 2 workgroups per CU, registers are free)
// Same ascending (tap, channel) MFMA chain per output: bit-identical to the other conv kernels.
// =====================================================================================
template <int C8, int NT, bool HALF = kH16>
__global__ void __launch_bounds__(256, 2) conv3x3_tile_kernel(const ConvArgs a, const Epilogue ep, const int tiles_x,
                                                              const int tiles_y) {
  using WV = typename WFrag<HALF>::T;
  constexpr int TH = 8, TW = 16, RH = TH + 2, RW = TW + 2;
  constexpr int CS = C8 * 8, STRIDE = CS + 4;  // floats per staged pixel
  constexpr int Q = CS / 4;                    // 16-byte pieces per pixel
  extern __shared__ float4 s_tile4[];
  float* s_tile = (float*)s_tile4;             // [RH*RW][STRIDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned groups = (unsigned)a.NTtot / NT;
  const int nt0 = (int)(lb % groups) * NT;
  unsigned tile = lb / groups;
  const int tx = (int)(tile % tiles_x);
  tile /= tiles_x;
  const int ty = (int)(tile % tiles_y), n = (int)(tile / tiles_y);
  const int y0 = ty * TH, x0 = tx * TW;
  // ---- fill: (RH*RW) pixels x Q pieces, consecutive threads take consecutive pieces of a pixel
  {
    const long img = (long)n * a.H * a.W * CS;  // element index of the image (f16 tensors: converted up on the way into LDS)
    constexpr int PIECES = RH * RW * Q, PER_THR = (PIECES + 255) / 256;
    float4 r[PER_THR];
#pragma unroll
    for (int i = 0; i < PER_THR; ++i) {
      const int idx = tid + i * 256;
      const int px = idx / Q, q = idx - px * Q;
      const int py = px / RW, pxx = px - py * RW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
      const bool v = idx < PIECES && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      r[i] = v ? ld4<HALF>(a.in, img + ((long)iy * a.W + ix) * CS + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < PER_THR; ++i) {
      const int idx = tid + i * 256;
      const int px = idx / Q, q = idx - px * Q;
      if (idx < PIECES) *(float4*)(s_tile + px * STRIDE + q * 4) = r[i];
    }
  }
  __shared__ float s_par[OCR_MAX_EP * 2 * NT * 32];
  conv_stage_params<NT>(a, ep, nt0, s_par);  // ends with the barrier that also publishes the tile

  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const int ly = wave * 2 + (p >> 4), lx = p & 15;  // this lane's pixel inside the tile
  const float* sA = s_tile + (ly * RW + lx) * STRIDE + 4 * h;
  const WV* __restrict__ wf = (const WV*)a.wfrag + (long)nt0 * 64 + lane;
  const long wstride = (long)a.NTtot * 64;
  auto load_tap = [&](WV (&bv)[C8][NT], int tap) {
    const WV* q = wf + (long)(tap < 9 ? tap : 8) * C8 * wstride;
#pragma unroll
    for (int c = 0; c < C8; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[c][t] = q[c * wstride + t * 64];
  };
  auto mul_tap = [&](const WV (&bv)[C8][NT], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const float* src = sA + (ky * RW + kx) * STRIDE;
#pragma unroll
    for (int c = 0; c < C8; ++c) {
      const float4 av = *(const float4*)(src + c * 8);
      if constexpr (HALF) {
        const ocr_h4 ah = ocr_to_h4(av);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x8f16(ocr_as_h4(bv[c][t]), ah, acc[t], 0, 0, 0);
      } else
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[c][t].x, av.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[c][t].y, av.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[c][t].z, av.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[c][t].w, av.w, acc[t], 0, 0, 0);
      }
    }
  };
  WV bA[C8][NT], bB[C8][NT];
  load_tap(bA, 0);
#pragma unroll 1
  for (int tap = 0; tap < 9; tap += 2) {
    load_tap(bB, tap + 1);
    __builtin_amdgcn_sched_barrier(0);
    mul_tap(bA, tap);
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < 9) {
      load_tap(bA, tap + 2);
      __builtin_amdgcn_sched_barrier(0);
      mul_tap(bB, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int oy = y0 + ly, ox = x0 + lx;
  const long m = (oy < a.OH && ox < a.OW) ? ((long)n * a.OH + oy) * a.OW + ox : a.M;
  conv_finish<NT, OUT_C8I, HALF>(a, ep, acc, nt0, m, h, s_par);
}

#ifdef OCR_TU_H16
// The LDS-tile conv on v_mfma_f32_32x32x16_f16 (precision "fp16"): the tile is staged AS f16 (half the LDS of the f32 form,
// four workgroups per CU instead of two, no conversion on either side), a lane's operand of an octet pair is one
// ds_read_b128 - lane (p, h) takes the eight channels of octet 2c + h of its pixel - and the weights come from the paired
// fragment image ([tap][octet pair][column tile][64 lanes][8 halfs], `frag16x:`).  Pixel stride in LDS = CS + 8 halfs
// (an odd number of 16-byte units).
template <int C8, int NT>
__global__ void __launch_bounds__(256, 2) conv3x3_tile16_kernel(const ConvArgs a, const Epilogue ep, const int tiles_x, const int tiles_y) {
  static_assert(C8 % 2 == 0, "octet pairs");
  constexpr int TH = 8, TW = 16, RH = TH + 2, RW = TW + 2;
  constexpr int CS = C8 * 8, STRIDE = CS + 8;  // halfs per staged pixel
  constexpr int Q = CS / 8, C16 = C8 / 2;      // 16-byte pieces (octets) per pixel; octet pairs
  extern __shared__ float4 s_tile4[];
  _Float16* s_tile = (_Float16*)s_tile4;       // [RH*RW][STRIDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
  const unsigned groups = (unsigned)a.NTtot / NT;
  const int nt0 = (int)(lb % groups) * NT;
  unsigned tile = lb / groups;
  int tx, ty, n;
  int IH = a.H, IW = a.W;  // the tile's image (ragged batch of images: its own size, conv3x3_c24_kernel's decode)
  long pix0;
  if (a.rtiles) {
    n = rag_line(a.rtiles, a.N, tile, 1);
    IH = rag_h(a.rin, n, a.H); IW = rag_w(a.rin, n);
    pix0 = rag_pix0(a.rin, n, a.H);
    const unsigned t = tile - (unsigned)a.rtiles[n], txn = (unsigned)(IW + TW - 1) / TW;
    ty = (int)(t / txn);
    tx = (int)(t - (unsigned)ty * txn);
  } else {
    tx = (int)(tile % tiles_x);
    tile /= tiles_x;
    ty = (int)(tile % tiles_y);
    n = (int)(tile / tiles_y);
    pix0 = (long)n * a.H * a.W;
  }
  const int y0 = ty * TH, x0 = tx * TW;
  if (a.cat_n) {  // a folded concat: source by source, as conv3x3_c24_kernel's fill (8-channel pieces: three per source pixel)
    constexpr int CCS = 24, QS = CCS / 8, PER_SRC = RH * RW * QS, PER_THR = (PER_SRC + 255) / 256;
    static_assert(CS == 4 * CCS, "four 24-channel sources");
    uint4 r[4][PER_THR];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lu = 31 - __clz(a.cat_up[j]);
      const int sw = IW >> lu;
      const int base = a.rtiles ? (a.rin.cw[n] >> (2 * (a.rin.shift + lu))) : n * (IH >> lu) * sw;
      const _Float16* sp = (const _Float16*)a.cat_src[j];
#pragma unroll
      for (int i = 0; i < PER_THR; ++i) {
        const int p = tid + i * 256;
        const int px = p / QS, ql = p - px * QS;
        const int py = px / RW, pxx = px - py * RW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
        const bool v = p < PER_SRC && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
        r[j][i] = v ? *(const uint4*)(sp + (base + (iy >> lu) * sw + (ix >> lu)) * CCS + 8 * ql) : uint4{0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < PER_THR; ++i) {
        const int p = tid + i * 256;
        const int px = p / QS, ql = p - px * QS;
        if (p < PER_SRC) *(uint4*)(s_tile + px * STRIDE + j * CCS + 8 * ql) = r[j][i];
      }
  } else {
    const _Float16* img = (const _Float16*)a.in + pix0 * CS;
    constexpr int PIECES = RH * RW * Q, PER_THR = (PIECES + 255) / 256;
    uint4 r[PER_THR];
#pragma unroll
    for (int i = 0; i < PER_THR; ++i) {
      const int idx = tid + i * 256;
      const int px = idx / Q, q = idx - px * Q;
      const int py = px / RW, pxx = px - py * RW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
      const bool v = idx < PIECES && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
      r[i] = v ? *(const uint4*)(img + ((long)iy * IW + ix) * CS + q * 8) : uint4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < PER_THR; ++i) {
      const int idx = tid + i * 256;
      const int px = idx / Q, q = idx - px * Q;
      if (idx < PIECES) *(uint4*)(s_tile + px * STRIDE + q * 8) = r[i];
    }
  }
  __shared__ float s_par[OCR_MAX_EP * 2 * NT * 32];
  conv_stage_params<NT>(a, ep, nt0, s_par);  // ends with the barrier that also publishes the tile

  floatx16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  const int ly = wave * 2 + (p >> 4), lx = p & 15;  // this lane's pixel inside the tile
  const _Float16* sA = s_tile + (ly * RW + lx) * STRIDE + 8 * h;
  const uint4* __restrict__ wf = (const uint4*)a.wfrag_x16 + (long)nt0 * 64 + lane;
  const long wstride = (long)a.NTtot * 64;
  auto load_tap = [&](uint4 (&bv)[C16][NT], int tap) {
    const uint4* q = wf + (long)(tap < 9 ? tap : 8) * C16 * wstride;
#pragma unroll
    for (int c = 0; c < C16; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) bv[c][t] = q[c * wstride + t * 64];
  };
  auto mul_tap = [&](const uint4 (&bv)[C16][NT], int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const _Float16* src = sA + (ky * RW + kx) * STRIDE;
#pragma unroll
    for (int c = 0; c < C16; ++c) {
      const ocr_h8 ah = __builtin_bit_cast(ocr_h8, *(const uint4*)(src + c * 16));
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ocr_h8, bv[c][t]), ah, acc[t], 0, 0, 0);
    }
  };
  uint4 bA[C16][NT], bB[C16][NT];
  load_tap(bA, 0);
#pragma unroll 1
  for (int tap = 0; tap < 9; tap += 2) {
    load_tap(bB, tap + 1);
    __builtin_amdgcn_sched_barrier(0);
    mul_tap(bA, tap);
    __builtin_amdgcn_sched_barrier(0);
    if (tap + 1 < 9) {
      load_tap(bA, tap + 2);
      __builtin_amdgcn_sched_barrier(0);
      mul_tap(bB, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int oy = y0 + ly, ox = x0 + lx;
  const long m = (oy < IH && ox < IW) ? pix0 + (long)oy * IW + ox : a.M;  // (stride 1, pad 1: the output has the input's size)
  conv_finish<NT, OUT_C8I, true>(a, ep, acc, nt0, m, h, s_par);
}
#endif  // OCR_TU_H16

#ifndef OCR_TU_H16  // (f32 contract only)
// =====================================================================================
// The same conv for 24 output channels on 4x4x1 matrix blocks (the five DB neck / head 96 -> 24 convs).
// conv3x3_tile_kernel<12, 1> multiplies 32-column tiles: with 24 channels a quarter of every v_mfma_f32_32x32x2_f32 is
// spent on zero columns (83 TFLOP/s useful = 0.71 of a pipe that can give 0.75).  v_mfma_f32_4x4x1_16B_f32 is sixteen
// independent 4x4 outer products per instruction (measured 128 TFLOP/s back to back, tools/micro/mfma_4x4.hip, against
// 150 for 32x32x2): a block = 4 pixels x 4 output channels, so 24 channels are six blocks wide and nothing is wasted.
//   lane l of a wave  : B operand = pixel l of the wave's 64 pixels (one input channel per instruction, K = 1);
//                       result    = its pixel's channels 4g .. 4g+3 in four registers -> one 16-byte store
//   A operand         : ONE register per k for all 24 channels - lane 4g + m holds the weight of channel 4g + m, and
//                       instruction g broadcasts block g's four lanes to all sixteen blocks (CBSZ = 4, ABID = g)
//   workgroup         : the 8 x 16 pixel tile of conv3x3_tile_kernel (10 x 18 x 96 halo region in LDS, one barrier),
//                       4 waves = 2 pixel halves x 2 channel halves (3 blocks of 4 channels each)
//   k order           : tap-major, then LOGICAL input channel ascending - an octet of the C8I layout is read as
//                       two float4 (physical 0..3 = logical 0,2,4,6; physical 4..7 = logical 1,3,5,7) and issued
//                       lo.x, hi.x, lo.y, hi.y, ...: the contract's chain, one fma per instruction
//   weights           : image [tap][octet][32 lanes][8 logical steps] (lanes 24..31 zero), two 16-byte loads per lane
//                       and octet, a whole octet ahead
// =====================================================================================
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int C8, int TH>
__global__ void __launch_bounds__(TH * 32, 2) conv3x3_c24_kernel(const ConvArgs a, const Epilogue ep, const float* __restrict__ wimg,
                                                             const int tiles_x, const int tiles_y) {
  constexpr int TW = 16, RH = TH + 2, RW = TW + 2, NTHR = TH * 32;  // 64 pixels x 2 channel halves per 4 tile rows
  constexpr int CS = C8 * 8, STRIDE = CS + 4;  // floats per staged pixel
  constexpr int Q = CS / 4;                    // 16-byte pieces per pixel
  extern __shared__ float4 s_tile4[];
  float* s_tile = (float*)s_tile4;             // [RH*RW][STRIDE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lb = xcd_swizzle(blockIdx.x, gridDim.x);
  unsigned tile = lb;
  int tx, ty, n;
  int IH = a.H, IW = a.W;  // the tile's image (ragged batch of images: its own size)
  long pix0;
  if (a.rtiles) {
    n = rag_line(a.rtiles, a.N, tile, 1);
    IH = rag_h(a.rin, n, a.H); IW = rag_w(a.rin, n);
    pix0 = rag_pix0(a.rin, n, a.H);
    const unsigned t = tile - (unsigned)a.rtiles[n], txn = (unsigned)(IW + TW - 1) / TW;
    ty = (int)(t / txn);
    tx = (int)(t - (unsigned)ty * txn);
  } else {
    tx = (int)(tile % tiles_x);
    tile /= tiles_x;
    ty = (int)(tile % tiles_y);
    n = (int)(tile / tiles_y);
    pix0 = (long)n * a.H * a.W;
  }
  const int y0 = ty * TH, x0 = tx * TW;
  if (a.cat_n) {
    // a folded concat (ConvArgs::cat_*): the fill gathers the tile's channels from the four 24-channel sources, SOURCE BY
    // SOURCE - consecutive threads take consecutive quads of consecutive pixels of ONE source, whose rows are contiguous in
    // memory (piece by piece of a pixel across the sources, every load instruction touched four tensors: +0.43 ms)
    constexpr int CCS = 24, QS = CCS / 4, PER_SRC = RH * RW * QS, PER_THR = (PER_SRC + NTHR - 1) / NTHR;
    static_assert(CS == 4 * CCS, "four 24-channel sources");
    float4 r[4][PER_THR];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // (the source as a compile-time constant: its pointer, factor and row width are scalars)
      const int lu = 31 - __clz(a.cat_up[j]);
      const int sw = IW >> lu;
      const int base = a.rtiles ? (a.rin.cw[n] >> (2 * (a.rin.shift + lu))) : n * (IH >> lu) * sw;  // (source pixels < 2^31: checked at bind)
      const float* sp = a.cat_src[j];
#pragma unroll
      for (int i = 0; i < PER_THR; ++i) {
        const int p = tid + i * NTHR;
        const int px = p / QS, ql = p - px * QS;
        const int py = px / RW, pxx = px - py * RW;
        const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
        const bool v = p < PER_SRC && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
        r[j][i] = v ? *(const float4*)(sp + (base + (iy >> lu) * sw + (ix >> lu)) * CCS + 4 * ql) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < PER_THR; ++i) {
        const int p = tid + i * NTHR;
        const int px = p / QS, ql = p - px * QS;
        if (p < PER_SRC) *(float4*)(s_tile + px * STRIDE + j * CCS + 4 * ql) = r[j][i];
      }
  } else {  // fill: (RH*RW) pixels x Q pieces, consecutive threads take consecutive pieces of a pixel
    const float* img = a.in + pix0 * CS;
    constexpr int PIECES = RH * RW * Q, PER_THR = (PIECES + NTHR - 1) / NTHR;
    float4 r[PER_THR];
#pragma unroll
    for (int i = 0; i < PER_THR; ++i) {
      const int idx = tid + i * NTHR;
      const int px = idx / Q, q = idx - px * Q;
      const int py = px / RW, pxx = px - py * RW;
      const int iy = y0 - 1 + py, ix = x0 - 1 + pxx;
      const bool v = idx < PIECES && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
      r[i] = v ? *(const float4*)(img + ((long)iy * IW + ix) * CS + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < PER_THR; ++i) {
      const int idx = tid + i * NTHR;
      const int px = idx / Q, q = idx - px * Q;
      if (idx < PIECES) *(float4*)(s_tile + px * STRIDE + q * 4) = r[i];
    }
  }
  __syncthreads();
  const int ph = wave & 1, chh = __builtin_amdgcn_readfirstlane(wave >> 1);  // the channel half as a scalar
  const int ly = ph * 4 + (lane >> 4), lx = lane & 15;  // this lane's pixel inside the tile
  const float* sA = s_tile + (ly * RW + lx) * STRIDE;
  const float4* wl = (const float4*)wimg + (lane & 31) * 2;  // [tap][octet][32][8]
  floatx4 acc[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) acc[g] = floatx4{0.f, 0.f, 0.f, 0.f};
  // operands of an octet: the pixel's eight channels from LDS one octet ahead of their 24 instructions, the weights - 110 KB per
  // conv, an L2 round trip per fetch once four waves per SIMD-pair stream them through a 16 KB L1 - TWO octets ahead (round 5:
  // a ring of three weight pairs; one octet = 24 instructions x ~20 clocks is shorter than that round trip)
  struct Wt { float4 w0, w1; };
  struct Px { float4 lo, hi; };
  auto load_w = [&](Wt& o, const float4* w_tap, int c) __attribute__((always_inline)) { o.w0 = w_tap[c * 64]; o.w1 = w_tap[c * 64 + 1]; };
  auto load_p = [&](Px& o, const float* src_tap, int c) __attribute__((always_inline)) {
    o.lo = *(const float4*)(src_tap + c * 8);
    o.hi = *(const float4*)(src_tap + c * 8 + 4);
  };
  auto mul_oct = [&](const Wt& w, const Px& o) __attribute__((always_inline)) {
    const float xs[8] = {o.lo.x, o.hi.x, o.lo.y, o.hi.y, o.lo.z, o.hi.z, o.lo.w, o.hi.w};  // logical channels 8c .. 8c+7
    const float ws[8] = {w.w0.x, w.w0.y, w.w0.z, w.w0.w, w.w1.x, w.w1.y, w.w1.z, w.w1.w};
    if (chh == 0) {  // wave-uniform (a scalar branch): blocks 0..2 or 3..5 of the weight register
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ws[s], xs[s], acc[0], 4, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ws[s], xs[s], acc[1], 4, 1, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ws[s], xs[s], acc[2], 4, 2, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(ws[s], xs[s], acc[0], 4, 3, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(ws[s], xs[s], acc[1], 4, 4, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(ws[s], xs[s], acc[2], 4, 5, 0);
      }
    }
  };
  static_assert(C8 % 6 == 0, "octets are walked in groups of six (two pixel sets x three weight sets)");
  Wt wr[3];
  Px pr[2];
  load_w(wr[0], wl, 0);
  load_w(wr[1], wl, 1);
  load_p(pr[0], sA, 0);
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {   // the 12 octets of a tap unrolled: every operand address is base + constant
    const int ky = tap / 3, kx = tap - ky * 3;
    const float* src = sA + (ky * RW + kx) * STRIDE;
    const float4* wt = wl + (long)tap * C8 * 64;
    const int nt = tap < 8 ? tap + 1 : 8, nky = nt / 3, nkx = nt - nky * 3;  // past the end: the last tap again (unused)
    const float* nsrc = sA + (nky * RW + nkx) * STRIDE;
    const float4* nwt = wl + (long)nt * C8 * 64;
#pragma unroll
    for (int c = 0; c < C8; ++c) {
      // weights of octet c + 2, pixels of octet c + 1 (into the sets octet c - 1 has left), then octet c's instructions
      if (c + 2 < C8) load_w(wr[(c + 2) % 3], wt, c + 2);
      else load_w(wr[(c + 2) % 3], nwt, c + 2 - C8);
      if (c + 1 < C8) load_p(pr[(c + 1) % 2], src, c + 1);
      else load_p(pr[(c + 1) % 2], nsrc, 0);
#ifndef OCR_C24_NOSCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
      mul_oct(wr[c % 3], pr[c % 2]);
#ifndef OCR_C24_NOSCHED
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  }
  const int oy = y0 + ly, ox = x0 + lx;
  if (oy >= IH || ox >= IW) return;  // (stride 1, pad 1: the output has the input's size)
  const long m = pix0 + (long)oy * IW + ox;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int pc = chh * 12 + g * 4;
    float4 v = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
    v = apply_epilogue4(ep, v, pc, n, oy, ox, m * a.Cs_out + pc, a.Cs_out);
    *(float4*)(a.out + m * a.Cs_out + pc) = v;
  }
}

// weights image of conv3x3_c24_kernel from the logical filter w[co][ci][3][3] (co = 24, ci = 96); host side
std::vector<float> conv3x3_c24_image(const float* w, int co, int ci) {
  const int C8 = ci / 8;
  std::vector<float> img((size_t)9 * C8 * 32 * 8, 0.f);
  for (int tap = 0; tap < 9; ++tap)
    for (int c = 0; c < C8; ++c)
      for (int pco = 0; pco < 24; ++pco)
        for (int s = 0; s < 8; ++s) {
          const int lco = c8i_logical(pco), lci = c * 8 + s;  // logical
          if (lco < co) img[((((size_t)tap * C8 + c) * 32) + pco) * 8 + s] = w[((size_t)lco * ci + lci) * 9 + tap];
        }
  return img;
}

bool launch_conv3x3_c24(const ConvArgs& a, const Epilogue& ep, const float* wimg, hipStream_t s, bool query) {
  if (!rt_options().conv_c24) return false;  // OCR_CONV_C24=0: the 32-column tile kernel (A/B; results are identical)
  if (!wimg || !(a.KH == 3 && a.KW == 3 && a.PH == 1 && a.PW == 1 && a.OH == a.H && a.OW == a.W && a.out_mode == OUT_C8I)) return false;
  if (a.Cs_in != 96 || a.Cs_out != 24 || a.Cout != 24) return false;
  if (a.cat_n && (a.cat_n != 4 || a.cat_cs != 24)) return false;  // (the folded fill is compiled for four 24-channel sources)
  for (int i = 0; i < ep.n; ++i) if (ep.st[i].kind == EP_ADDUP) return false;
  const int tiles_x = (a.OW + 15) / 16;   // (4-row tiles, 3 workgroups per CU: 64 instead of 90 TFLOP/s - measured, removed)
  const int tiles_y = (a.OH + 7) / 8;
  if (a.rin.w && !a.rtiles) return false;  // a ragged batch needs the tile table (images) - lines never come here
  for (int i = 0; i < ep.n; ++i) if (a.rtiles && ep.st[i].kind == EP_MULC) return false;
  const dim3 grid((unsigned)(a.rtiles ? (long)a.rtiles_total : (long)a.N * tiles_y * tiles_x));
  const unsigned lds = 10 * 18 * (96 + 4) * sizeof(float);  // 72 000 B: two workgroups per CU
  static LdsAttrMemo attr_state;
  if (!raise_dynamic_lds((const void*)conv3x3_c24_kernel<12, 8>, (int)lds, attr_state)) return false;
  if (query && rt_refuse_launch() == "conv3x3_c24@bind") return false;  // fault injection (tests): the bind-time probe is refused as a failed LDS attribute would refuse it
  if (query) return true;  // asked at bind time, on the device that will run it (net.hip folds the DB neck's concat only then)
  hipLaunchKernelGGL((conv3x3_c24_kernel<12, 8>), grid, dim3(256), lds, s, a, ep, wimg, tiles_x, tiles_y);
  return true;
}

#endif  // OCR_TU_H16
// true if the launch was taken (3x3, stride 1, pad 1, 96 input channels, C8I output)
bool OCR_L(launch_conv3x3_tile)(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s, bool query) {
  OCR_H16_TWIN(a.half, launch_conv3x3_tile_h16(a, ep, nt, s, query))
  if (!(a.KH == 3 && a.KW == 3 && a.PH == 1 && a.PW == 1 && a.OH == a.H && a.OW == a.W && a.out_mode == OUT_C8I)) return false;
  if (a.Cs_in != 96 || nt != 1) return false;
  // ragged batches: the f16-staged form takes a batch of IMAGES through the tile table (conv3x3_c24_kernel's scheme); the f32
  // form here decodes none (the chunked kernel decodes per row)
  if (a.rin.w && !(kH16 && a.rtiles && a.rin.h && a.wfrag_x16 && rt_options().mfma_x16)) return false;
  for (int i = 0; i < ep.n; ++i) if (a.rtiles && ep.st[i].kind == EP_MULC) return false;
  const int tiles_x = (a.OW + 15) / 16, tiles_y = (a.OH + 7) / 8;
  const dim3 grid((unsigned)((a.rtiles ? (long)a.rtiles_total : (long)a.N * tiles_y * tiles_x) * (a.NTtot / nt)));
  const unsigned lds = 10 * 18 * (96 + 4) * sizeof(float);  // 72 000 B: two workgroups per CU
  // more than 64 KB of dynamic LDS has to be allowed per device (a worker pool drives several from one process)
  static LdsAttrMemo attr_state;
  if (a.cat_n && (a.cat_n != 4 || a.cat_cs != 24)) return false;  // (the folded fill is compiled for four 24-channel sources)
#ifdef OCR_TU_H16
  if (a.wfrag_x16 && rt_options().mfma_x16) {  // the f16-staged 32x32x16 form (52 bytes under the 64 KB that need no attribute)
    const unsigned lds16 = 10 * 18 * (96 + 8) * sizeof(_Float16);  // 37 440 B: four workgroups per CU
    if (query) return true;
    hipLaunchKernelGGL((conv3x3_tile16_kernel<12, 1>), grid, dim3(256), lds16, s, a, ep, tiles_x, tiles_y);
    return true;
  }
#endif
  if (a.cat_n) return false;  // (a folded concat: the 4x4x1 / f16-staged forms only)
  if (!raise_dynamic_lds((const void*)conv3x3_tile_kernel<12, 1>, (int)lds, attr_state)) return false;  // the general kernel takes the launch
  if (query) return true;
  hipLaunchKernelGGL((conv3x3_tile_kernel<12, 1>), grid, dim3(256), lds, s, a, ep, tiles_x, tiles_y);
  return true;
}

#ifndef OCR_TU_H16
template <int NT>
static void launch_conv_lds_nt(const ConvArgs& a, const Epilogue& ep, dim3 grid, hipStream_t s) {
  if (a.Cs_in % 32 == 0) hipLaunchKernelGGL((conv_lds_kernel<NT, 32>), grid, dim3(256), 0, s, a, ep);
  else if (a.Cs_in % 16 == 0) hipLaunchKernelGGL((conv_lds_kernel<NT, 16>), grid, dim3(256), 0, s, a, ep);
  else hipLaunchKernelGGL((conv_lds_kernel<NT, 8>), grid, dim3(256), 0, s, a, ep);
}
void launch_conv_lds(const ConvArgs& a, const Epilogue& ep, int nt, hipStream_t s) {
  dim3 grid((unsigned)(((a.M + 127) / 128) * (a.NTtot / nt)));
  switch (nt) {
    case 1: launch_conv_lds_nt<1>(a, ep, grid, s); break;
    case 2: launch_conv_lds_nt<2>(a, ep, grid, s); break;
    case 3: launch_conv_lds_nt<3>(a, ep, grid, s); break;
    default: launch_conv_lds_nt<4>(a, ep, grid, s); break;
  }
}
#endif  // OCR_TU_H16

// =====================================================================================
// Stem: dense conv with Cin = 3 on the plain NHWC3 f32 image (any stride/pad), VALU.
// One thread = one output pixel, all CS output channels; weights are wave-uniform (scalar loads).
// =====================================================================================
template <int CS, bool H16 = kH16>
__global__ void __launch_bounds__(256) stem_conv_kernel(const StemArgs a, const Epilogue ep) {
  const long m_wg = (long)xcd_swizzle(blockIdx.x, gridDim.x) * 256;
  if (m_wg + (threadIdx.x & ~63) >= a.M) return;  // (a whole wave past the end)
  const long m_lane = m_wg + threadIdx.x;
  const long m = m_lane < a.M ? m_lane : a.M - 1;  // lanes past the end compute the last pixel again: every lane of the wave takes part in the stores below
  int n, y, x;
  int iw = a.W, ih = a.H;    // input size of the pixel's image (ragged batch: the sample's own)
  long ipix0;                // first input pixel of that image
  if (a.rout.w) {
    int ow;
    rag_decompose(a.rout, a.N, a.OH, m, m_wg, n, y, x, ow);
    iw = rag_w(a.rin, n);
    ih = rag_h(a.rin, n, a.H);
    ipix0 = rag_pix0(a.rin, n, a.H);
  } else {
    decompose(m, a.OH * a.OW, a.OW, n, y, x);
    ipix0 = (long)n * a.H * a.W;
  }
  // (two adjacent output channels per v_pk_fma_f32 - each lane of it is the same IEEE fma as the scalar instruction, the weights
  // are wave-uniform scalar pairs; as 27 x CS scalar fmas the kernel spent a third of its time issuing them)
  ocr_f2 acc[CS / 2];
#pragma unroll
  for (int c = 0; c < CS / 2; ++c) acc[c] = ocr_f2{0.f, 0.f};
  // one 12-byte load per tap (the pixel's three floats; 4-byte alignment is all dwordx3 needs), masked afterwards
  // with a lane mask: three dword loads under a select each made the kernel bound by the number of load instructions
  struct P3 { float c[3]; };
  auto tap_load = [&](int ky, int kx, P3& px, unsigned& keep) {
    const int iy = y * a.SH - a.PH + ky, ix = x * a.SW - a.PW + kx;
    const bool v = iy >= 0 && iy < ih && ix >= 0 && ix < iw;
    const float* src = a.in + (ipix0 + (long)(v ? iy : 0) * iw + (v ? ix : 0)) * 3;
    __builtin_memcpy(&px, __builtin_assume_aligned(src, 4), 12);
    keep = v ? 0xffffffffu : 0u;
  };
  auto tap_fma = [&](int ky, int kx, const P3& px, unsigned keep) {
    float in3[3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) in3[ci] = __uint_as_float(__float_as_uint(px.c[ci]) & keep);
    const float* w = a.w + (long)(ky * a.KW + kx) * 3 * CS;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int c = 0; c < CS; c += 2)
        acc[c >> 1] = __builtin_elementwise_fma(ocr_f2{in3[ci], in3[ci]}, ocr_f2{w[ci * CS + c], w[ci * CS + c + 1]}, acc[c >> 1]);
  };
  if (a.KH == 3 && a.KW == 3) {  // every stem on this path: the nine taps' loads in flight before the first FMA
    P3 px[9];
    unsigned keep[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap_load(t / 3, t % 3, px[t], keep[t]);
#pragma unroll
    for (int t = 0; t < 9; ++t) tap_fma(t / 3, t % 3, px[t], keep[t]);
  } else {
    for (int ky = 0; ky < a.KH; ++ky)
      for (int kx = 0; kx < a.KW; ++kx) {
        P3 px;
        unsigned keep;
        tap_load(ky, kx, px, keep);
        tap_fma(ky, kx, px, keep);
      }
  }
  // Stores: a lane's pixel is CS * 4 bytes, so lane-per-pixel stores put 16 bytes into each of 64 different lines per
  // instruction and the lines reach HBM partly merged (round 5 PMC: 1.25 GB written for a 0.94 GB tensor).  The wave's 64
  // pixels are consecutive in memory: through a per-wave LDS tile the same quads leave as whole lines - instruction i
  // writes pixels 64 i / Q ... of the wave, Q = CS / 4 lanes per pixel, 1 KB contiguous.
  constexpr int Q = CS / 4, PS = CS + 4;  // quads per pixel; padded pixel stride in the tile (floats)
  static_assert(64 % Q == 0, "whole-line stores: instruction i writes pixels (64 / Q) i + lane / Q - Q = CS / 4 must divide the wave (CS = 8, 16)");
  __shared__ float s_tile[4][64 * PS];
  float* const tile = s_tile[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < CS; c += 4) {
    float4 v = make_float4(acc[c >> 1].x, acc[c >> 1].y, acc[(c >> 1) + 1].x, acc[(c >> 1) + 1].y);
    v = apply_epilogue4<H16>(ep, v, c, n, y, x, m * CS + c, CS);
    *(float4*)(tile + lane * PS + c) = v;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const long m_wave = m_lane - lane;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int pl = (64 / Q) * i + lane / Q, q = lane % Q;  // pixel of the wave, quad of the pixel
    if (m_wave + pl < a.M) st4<H16>(a.out, (m_wave + pl) * CS + 4 * q, *(const float4*)(tile + pl * PS + 4 * q));
  }
}

void OCR_L(launch_stem)(const StemArgs& a, const Epilogue& ep, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_stem_h16(a, ep, s))
  dim3 grid((unsigned)((a.M + 255) / 256));
  if (a.Cs_out == 16) hipLaunchKernelGGL(stem_conv_kernel<16>, grid, dim3(256), 0, s, a, ep);
  else hipLaunchKernelGGL(stem_conv_kernel<8>, grid, dim3(256), 0, s, a, ep);
}

// =====================================================================================
// Depthwise conv, VALU.  One thread = a TO x R patch of output pixels (TO along x, R rows) x 4 physical
// channels.  The thread walks the (R-1)*SH + K input rows the patch needs from top to bottom; each row
// segment is loaded once into registers and feeds every output row it is a tap of, so an output costs
// ((R-1)*SH+K) * ((TO-1)*SW+K) / (TO*R) input loads instead of K*K (5x5, stride 1, 8x2: 4.5 vs 25; the
// 4x1 patch this replaces: 10), plus K*K / TO weight loads.  The kernel is bound by the L1/TA rate of
// those loads, not by HBM.  For every output the taps still arrive in (ky, kx) ascending order (rows
// top to bottom, kx inner): the arithmetic contract is untouched.
// =====================================================================================
// The K*K x Cs weights are copied to LDS once per workgroup (a workgroup's 256 threads span every channel
// quad): a thread's 50 weight fetches per patch become LDS reads and leave the L1/TA path to the pixels.
template <int K, int SW, int TO, int R, bool ROWSUM = false, bool RAG = false, bool H16 = kH16>
__global__ void __launch_bounds__(256, 2) dw_conv_kernel(const DwArgs a, const Epilogue ep) {
  constexpr int NIN = (TO - 1) * SW + K;
  extern __shared__ float4 s_dw_w[];  // [K*K][Cs/4]
  {
    const int nw4 = K * K * (a.Cs >> 2);
    const float4* gw = (const float4*)a.w;
    for (int i = threadIdx.x; i < nw4; i += 256) s_dw_w[i] = gw[i];
    __syncthreads();
  }
  const long t_wg = (long)xcd_swizzle(blockIdx.x, gridDim.x) * 256;
  const long t = t_wg + threadIdx.x;
  const int c4n = a.Cs >> 2;
  const int bands = (a.OH + R - 1) / R;
  constexpr bool rag = RAG;  // ragged batch (own instantiation: the uniform one keeps its sizes in scalar registers)
  // ROWSUM: a thread owns a band (R output rows) of one image x 4 channels and walks its strips left to right, so the
  // pool's row sums (x ascending from 0, the contract's order) accumulate in registers beside the conv
  // ragged batch: work items are counted sample by sample (a.rwork: prefix sums of the samples' bands (ROWSUM) or
  // bands x strips)
  const long npatch = rag ? (long)a.rwork_total : (ROWSUM ? (long)a.N * bands : (long)a.N * bands * ((a.OW + TO - 1) / TO));
  if (t >= npatch * c4n) return;
  const long sidx = t / c4n;
  const int pc = (int)(t - sidx * c4n) * 4;
  int n, y0, sx_first, sx_end;
  int IW = a.W, OW = a.OW, IH = a.H, OHn = a.OH;  // sizes of the patch's image (ragged batch: the sample's own)
  long ipix0, opix0, orow0;                        // its first input / output pixel, its first row of the row-sum buffer
  if constexpr (!rag) {
    const int strips = (a.OW + TO - 1) / TO;
    sx_first = ROWSUM ? 0 : (int)(sidx % strips);
    const long nb = ROWSUM ? sidx : sidx / strips;
    y0 = (int)(nb % bands) * R;
    n = (int)(nb / bands);
    sx_end = ROWSUM ? strips : sx_first + 1;
    ipix0 = (long)n * a.H * a.W; opix0 = (long)n * a.OH * a.OW; orow0 = (long)n * a.OH;
  } else {
    n = rag_line_near(a.rwork, a.N, sidx, 1, t_wg / c4n);
    IW = rag_w(a.rin, n); OW = rag_w(a.rout, n); IH = rag_h(a.rin, n, a.H); OHn = rag_h(a.rout, n, a.OH);
    const int strips = (OW + TO - 1) / TO;
    const int rem = (int)(sidx - a.rwork[n]);
    const int band = ROWSUM ? rem : rem / strips;
    y0 = band * R;
    sx_first = ROWSUM ? 0 : rem - band * strips;
    sx_end = ROWSUM ? strips : sx_first + 1;
    ipix0 = rag_pix0(a.rin, n, a.H); opix0 = rag_pix0(a.rout, n, a.OH); orow0 = rag_row0(a.rout, n, a.OH);
  }
  using F4 = DwF4;
  F4 rsum[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { rsum[r].lo = ocr_f2{0.f, 0.f}; rsum[r].hi = ocr_f2{0.f, 0.f}; }
#pragma unroll 1
  for (int sx = sx_first; sx < sx_end; ++sx) {
  const int x0 = sx * TO;
  // Accumulators and arithmetic are written as explicit 2-vectors (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32:
  // two f32 lanes per VALU slot, each lane the same IEEE operation as the scalar instruction), not left to the
  // SLP vectoriser, whose pairing changes with unrelated edits of the epilogue.
  F4 acc[R][TO];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int o = 0; o < TO; ++o) { acc[r][o].lo = ocr_f2{0.f, 0.f}; acc[r][o].hi = ocr_f2{0.f, 0.f}; }
  const int ixb = x0 * SW - a.PW, iyb = y0 * a.SH - a.PH;
  const int nrows = (R - 1) * a.SH + K;
  // Two row buffers in ping-pong: row j+1 is in flight while row j is consumed (the kernel runs at 2
  // waves/SIMD, so the bytes in flight per wave are what hides the memory latency).  Rows beyond the
  // image are zero rows; the row after the last one is loaded (clamped) and never used.
  using RV = typename WFrag<H16>::T;  // a loaded quad as it travels (f16 storage: converted in use_row, not next to its load)
  auto load_row = [&](RV (&in)[NIN], int j) {
    const int iy = iyb + j;
    const bool rv = j < nrows && iy >= 0 && iy < IH;
    const long row = (ipix0 + (long)(rv ? iy : 0) * IW) * a.Cs + pc;  // element index of the row's first quad
#pragma unroll
    for (int q = 0; q < NIN; ++q) {
      const int ix = ixb + q;
      in[q] = (rv && ix >= 0 && ix < IW) ? ld4_raw<H16>(a.in, row + (long)ix * a.Cs) : RV{};
    }
  };
  auto use_row = [&](const RV (&raw)[NIN], int j) {
    float4 in[NIN];
#pragma unroll
    for (int q = 0; q < NIN; ++q) in[q] = up4(raw[q]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ky = j - r * a.SH;  // uniform
      if (ky < 0 || ky >= K) continue;
      const float4* wrow = s_dw_w + (ky * K) * (a.Cs >> 2) + (pc >> 2);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 wv = wrow[kx * (a.Cs >> 2)];
        const ocr_f2 wlo = {wv.x, wv.y}, whi = {wv.z, wv.w};
#pragma unroll
        for (int o = 0; o < TO; ++o) {
          const float4 v = in[o * SW + kx];
          acc[r][o].lo = __builtin_elementwise_fma(ocr_f2{v.x, v.y}, wlo, acc[r][o].lo);
          acc[r][o].hi = __builtin_elementwise_fma(ocr_f2{v.z, v.w}, whi, acc[r][o].hi);
        }
      }
    }
  };
  RV rowA[NIN], rowB[NIN];
  load_row(rowA, 0);
#pragma unroll 1
  for (int j = 0; j < nrows; j += 2) {
    load_row(rowB, j + 1);
    __builtin_amdgcn_sched_barrier(0);
    use_row(rowA, j);
    __builtin_amdgcn_sched_barrier(0);
    load_row(rowA, j + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < nrows) use_row(rowB, j + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // epilogue: stage loop outside, the patch inside (one copy of each stage's code)
  const long obase = (opix0 + (long)y0 * OW + x0) * a.Cs + pc;
  const long orow = (long)OW * a.Cs;
  dw_patch_epilogue<R, TO, H16>(acc, ep, pc, n, a.Cs, obase, orow, y0, x0, OHn, OW);
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int o = 0; o < TO; ++o)
      if (y0 + r < OHn && x0 + o < OW) {
        st4<H16>(a.out, obase + r * orow + (long)o * a.Cs, make_float4(acc[r][o].lo.x, acc[r][o].lo.y, acc[r][o].hi.x, acc[r][o].hi.y));
        if constexpr (ROWSUM) { rsum[r].lo = rsum[r].lo + acc[r][o].lo; rsum[r].hi = rsum[r].hi + acc[r][o].hi; }  // s = s + v, x ascending
      }
  }  // strips
  if constexpr (ROWSUM) {
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (y0 + r < OHn)
        *(float4*)(a.rowsum + (orow0 + y0 + r) * a.Cs + pc) = make_float4(rsum[r].lo.x, rsum[r].lo.y, rsum[r].hi.x, rsum[r].hi.y);
  }
}

template <int TO, int R, bool RAG>
static void launch_dw_patch(const DwArgs& a, const Epilogue& ep, hipStream_t s) {
  const unsigned lds = (unsigned)(a.K * a.K * a.Cs * sizeof(float));  // <= 48 KB for 5x5 x 480 channels
  if (a.rowsum) {  // a thread per band: all strips of its rows
    const long bands = (RAG ? (long)a.rwork_total : (long)a.N * ((a.OH + R - 1) / R)) * (a.Cs >> 2);
    dim3 g((unsigned)((bands + 255) / 256));
    if (a.K == 3 && a.SW == 1) hipLaunchKernelGGL((dw_conv_kernel<3, 1, TO, R, true, RAG>), g, dim3(256), lds, s, a, ep);
    else if (a.K == 3) hipLaunchKernelGGL((dw_conv_kernel<3, 2, TO, R, true, RAG>), g, dim3(256), lds, s, a, ep);
    else if (a.SW == 1) hipLaunchKernelGGL((dw_conv_kernel<5, 1, TO, R, true, RAG>), g, dim3(256), lds, s, a, ep);
    else hipLaunchKernelGGL((dw_conv_kernel<5, 2, TO, R, true, RAG>), g, dim3(256), lds, s, a, ep);
    return;
  }
  const long total = (RAG ? (long)a.rwork_total : (long)a.N * ((a.OW + TO - 1) / TO) * ((a.OH + R - 1) / R)) * (a.Cs >> 2);
  dim3 grid((unsigned)((total + 255) / 256));
  if (a.K == 3 && a.SW == 1) hipLaunchKernelGGL((dw_conv_kernel<3, 1, TO, R, false, RAG>), grid, dim3(256), lds, s, a, ep);
  else if (a.K == 3) hipLaunchKernelGGL((dw_conv_kernel<3, 2, TO, R, false, RAG>), grid, dim3(256), lds, s, a, ep);
  else if (a.SW == 1) hipLaunchKernelGGL((dw_conv_kernel<5, 1, TO, R, false, RAG>), grid, dim3(256), lds, s, a, ep);
  else hipLaunchKernelGGL((dw_conv_kernel<5, 2, TO, R, false, RAG>), grid, dim3(256), lds, s, a, ep);
}
// output pixels per thread along x (ragged batch: OW = the narrowest line; the host builds DwArgs::rwork for this value)
#ifndef OCR_TU_H16
int dw_patch_to(int OW, int SW, int OH, int K) {
  (void)OH; (void)K;
  return (OW < 8 || SW == 2) ? 4 : 8;  // stride 2 needs 2*TO+K-2 pixels per row buffer: 8 wide does not fit the registers
}
// Output rows per thread.  Measured and NOT kept (round 4): full-height patches (R = 3 / 6) for the 5x5 layers on the
// recognizer's 3- and 6-row maps - every input row read once instead of 2.2x - are slower (rec op 26 with its row sums:
// 1.09 -> 1.88 ms, a third as many threads each walking 20 strips; ops 21 / 31 / 33 unchanged within 3 %): these
// kernels are bound by loads in flight per wave, not by the bytes the halo rows add.
int dw_patch_r(int OH, int K) {
  (void)K;
  return OH < 2 ? 1 : 2;
}
#endif  // OCR_TU_H16
void OCR_L(launch_dw)(const DwArgs& a, const Epilogue& ep, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_dw_h16(a, ep, s))
  const int r = dw_patch_r(a.OH, a.K);
  const int to = dw_patch_to(a.OW, a.SW, a.OH, a.K);
  if (a.rout.w) {  // ragged batch (the recognizer)
    if (to == 8 && r == 2) launch_dw_patch<8, 2, true>(a, ep, s);
    else if (to == 8) launch_dw_patch<8, 1, true>(a, ep, s);
    else if (r == 2) launch_dw_patch<4, 2, true>(a, ep, s);
    else launch_dw_patch<4, 1, true>(a, ep, s);
    return;
  }
  if (to == 8 && r == 2) launch_dw_patch<8, 2, false>(a, ep, s);
  else if (to == 8) launch_dw_patch<8, 1, false>(a, ep, s);
  else if (r == 2) launch_dw_patch<4, 2, false>(a, ep, s);
  else launch_dw_patch<4, 1, false>(a, ep, s);
}

// =====================================================================================
// Elementwise chain (SE gate multiply, residual add, FPN upsample-add).
// =====================================================================================
template <bool H16>
__global__ void __launch_bounds__(256) ew_kernel(const float* __restrict__ in, float* __restrict__ out, long M, int H,
                                                 int W, int Cs, const Epilogue ep, int N, const RagLevel rag) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int c4n = Cs >> 2;
  if (t >= M * c4n) return;
  const long m = t / c4n;
  const int pc = (int)(t - m * c4n) * 4;
  int n, y, x;
  long up_base = -1;
  int up_w = 0;
  if (rag.w) {  // ragged batch: the sample of this pixel (per-sample stages: the SE gate)
    int w;
    rag_decompose(rag, N, H, m, (long)blockIdx.x * 256 / c4n, n, y, x, w);
    if (rag.h) {  // images: an upsampled operand (FPN top-down add) lives on the level log2(up) coarser
      for (int s = 0; s < ep.n; ++s)
        if (ep.st[s].kind == EP_ADDUP) {
          const int lu = 31 - __clz(ep.st[s].a0);
          up_base = (long)(rag.cw[n] >> (2 * (rag.shift + lu)));
          up_w = rag.w[n] >> (rag.shift + lu);
        }
    }
  } else
  decompose(m, H * W, W, n, y, x);
  const long idx = m * Cs + pc;
  float4 v = ld4<H16>(in, idx);
  v = apply_epilogue4<H16>(ep, v, pc, n, y, x, idx, Cs, up_base, up_w);
  st4<H16>(out, idx, v);
}

void OCR_L(launch_ew)(const float* in, float* out, long M, int H, int W, int Cs, const Epilogue& ep, hipStream_t s, int N, RagLevel rag, bool h16) {
  OCR_H16_TWIN(h16, launch_ew_h16(in, out, M, H, W, Cs, ep, s, N, rag, h16))
  const long total = M * (Cs >> 2);
  hipLaunchKernelGGL(ew_kernel<kH16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, M, H, W, Cs, ep, N, rag);
}

// =====================================================================================
// Global average pool in the contract's order: row-sequential sums, then column-sequential.
// =====================================================================================
template <bool H16>
__global__ void __launch_bounds__(256) gap_rows_kernel(const float* __restrict__ in, float* __restrict__ part, int N,
                                                       int H, int W, int Cs, const RagLevel rag, long rows) {
  // one thread = 4 physical channels of one image row: four independent sequential sums, 16-byte loads
  const int c4n = Cs >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * c4n) return;
  const int pc = (int)(t % c4n) * 4;
  const long ny = t / c4n;
  long src = ny * W * Cs + pc;  // element index of the row's first quad
  if (rag.w) {  // ragged batch: row y of sample n, the sample's own width
    const int n = rag_sample_of_row(rag, N, H, ny, (long)blockIdx.x * 256 / c4n);
    const int y = (int)(ny - rag_row0(rag, n, H));
    W = rag_w(rag, n);
    src = (rag_pix0(rag, n, H) + (long)y * W) * Cs + pc;
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  // eight loads in flight, then their eight sequential adds (the sum's order is the contract's: left to right)
  constexpr int U = 8;
  int x = 0;
  for (; x + U <= W; x += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld4<H16>(in, src + (long)(x + u) * Cs);
#pragma unroll
    for (int u = 0; u < U; ++u) { s.x = s.x + v[u].x; s.y = s.y + v[u].y; s.z = s.z + v[u].z; s.w = s.w + v[u].w; }
  }
  for (; x < W; ++x) {
    const float4 v = ld4<H16>(in, src + (long)x * Cs);
    s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
  }
  *(float4*)(part + ny * Cs + pc) = s;
}
__global__ void __launch_bounds__(256) gap_cols_kernel(const float* __restrict__ part, float* __restrict__ out, int N,
                                                       int H, int Cs, float cnt, const RagLevel rag) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= N * Cs) return;
  const int pc = t % Cs, n = t / Cs;
  const float* src = part + (long)n * H * Cs + pc;
  if (rag.w) {  // ragged batch: the sample's own rows and pixel count
    src = part + rag_row0(rag, n, H) * Cs + pc;
    H = rag_h(rag, n, H);
    cnt = (float)(H * rag_w(rag, n));
  }
  float s = 0.f;
  for (int y = 0; y < H; ++y) s = s + src[(long)y * Cs];
  out[t] = s / cnt;
}
// =====================================================================================
// conv 1x1 (Cin <= 24) -> row sums, nothing written but the sums (kernels_net.h, ConvRowsumArgs).  One thread = one
// image row x 4 physical output channels, as gap_rows_kernel; per pixel the contract's chain for each of its channels
// (k ascending in LOGICAL input-channel order from 0, one fmaf per k - what the matrix pipe does with the fragment
// image), then s = s + v.  The 24 threads of a row read the same input pixels (L1 broadcasts).
// =====================================================================================
template <int CIN, int CS_IN, int U, bool RAG>
__global__ void __launch_bounds__(256) conv_rowsum_kernel(const ConvRowsumArgs a) {
  const int c4n = a.Cs_out >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.rows * c4n) return;
  const int pc = (int)(t % c4n) * 4;
  const long ny = t / c4n;
  ocr_f2 wlo[CIN], whi[CIN];  // (the padded k of the matrix kernels multiply zeros: fma(0, 0, acc) = acc, left out here)
#pragma unroll
  for (int k = 0; k < CIN; ++k) {
    const float4 w = *(const float4*)(a.w + (long)k * a.Cs_out + pc);
    wlo[k] = ocr_f2{w.x, w.y}; whi[k] = ocr_f2{w.z, w.w};
  }
  long src = ny * a.W * CS_IN;  // element index of the row's first pixel
  int W = a.W;  // (uniform batch: a scalar; the ragged form is its own instantiation - with a per-lane width in the same
                // code the uniform launch ran at half speed)
  if constexpr (RAG) {  // ragged batch: row y of sample n, the sample's own width
    const int n = rag_sample_of_row(a.rag, a.N, a.H, ny, (long)blockIdx.x * 256 / c4n);
    const int y = (int)(ny - rag_row0(a.rag, n, a.H));
    W = rag_w(a.rag, n);
    src = (rag_pix0(a.rag, n, a.H) + (long)y * W) * CS_IN;
  }
  ocr_f2 slo = {0.f, 0.f}, shi = {0.f, 0.f};
  auto pixel = [&](const float4 (&xin)[CS_IN / 4], ocr_f2& lo, ocr_f2& hi) __attribute__((always_inline)) {
    lo = ocr_f2{0.f, 0.f}; hi = ocr_f2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < CIN; ++k) {
      const int ph = c8i_phys(k), c = ph & 3;  // (constants after unrolling)
      const float4 q = xin[ph >> 2];
      const float xv = c == 0 ? q.x : c == 1 ? q.y : c == 2 ? q.z : q.w;
      lo = __builtin_elementwise_fma(ocr_f2{xv, xv}, wlo[k], lo);
      hi = __builtin_elementwise_fma(ocr_f2{xv, xv}, whi[k], hi);
    }
  };
  int x = 0;
  for (; x + U <= W; x += U) {
    float4 xin[U][CS_IN / 4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < CS_IN / 4; ++q) xin[u][q] = ld4<kH16>(a.in, src + (long)(x + u) * CS_IN + 4 * q);
    ocr_f2 lo[U], hi[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pixel(xin[u], lo[u], hi[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) { slo = slo + lo[u]; shi = shi + hi[u]; }  // left to right: the pool's order
  }
  for (; x < W; ++x) {
    float4 xin[CS_IN / 4];
#pragma unroll
    for (int q = 0; q < CS_IN / 4; ++q) xin[q] = ld4<kH16>(a.in, src + (long)x * CS_IN + 4 * q);
    ocr_f2 lo, hi;
    pixel(xin, lo, hi);
    slo = slo + lo; shi = shi + hi;
  }
  *(float4*)(a.part + ny * a.Cs_out + pc) = make_float4(slo.x, slo.y, shi.x, shi.y);
}
template <bool RAG>
static bool launch_conv_rowsum_r(const ConvRowsumArgs& a, hipStream_t s) {
  const long threads = a.rows * (a.Cs_out >> 2);
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (a.Cs_in == 16 && a.Cin == 12) hipLaunchKernelGGL((conv_rowsum_kernel<12, 16, 4, RAG>), grid, dim3(256), 0, s, a);
  else if (a.Cs_in == 24 && a.Cin == 18) hipLaunchKernelGGL((conv_rowsum_kernel<18, 24, 2, RAG>), grid, dim3(256), 0, s, a);
  else if (a.Cs_in == 16) hipLaunchKernelGGL((conv_rowsum_kernel<16, 16, 4, RAG>), grid, dim3(256), 0, s, a);
  else if (a.Cs_in == 24) hipLaunchKernelGGL((conv_rowsum_kernel<24, 24, 2, RAG>), grid, dim3(256), 0, s, a);
  else return false;
  return true;
}
bool OCR_L(launch_conv_rowsum)(const ConvRowsumArgs& a, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_conv_rowsum_h16(a, s))
  return a.rag.w ? launch_conv_rowsum_r<true>(a, s) : launch_conv_rowsum_r<false>(a, s);
}
#ifndef OCR_TU_H16
void launch_gap_cols(const float* part, float* out, int N, int H, int W, int Cs, hipStream_t s, RagLevel rag) {
  hipLaunchKernelGGL(gap_cols_kernel, dim3((unsigned)((N * Cs + 255) / 256)), dim3(256), 0, s, part, out, N, H, Cs, (float)(H * W), rag);
}
#endif
void OCR_L(launch_gap)(const float* in, float* part, float* out, int N, int H, int W, int Cs, hipStream_t s, RagLevel rag, long rows, bool h16) {
  OCR_H16_TWIN(h16, launch_gap_h16(in, part, out, N, H, W, Cs, s, rag, rows, h16))
  if (rows <= 0) rows = (long)N * H;  // (ragged batch of images: the sum of the images' heights)
  const long t1 = rows * (Cs >> 2);
  hipLaunchKernelGGL(gap_rows_kernel<kH16>, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, in, part, N, H, W, Cs, rag, rows);
  hipLaunchKernelGGL(gap_cols_kernel, dim3((unsigned)((N * Cs + 255) / 256)), dim3(256), 0, s, part, out, N, H, Cs,
                     (float)(H * W), rag);
}

#ifndef OCR_TU_H16  // (per-image vectors: f32 in both precisions)
// =====================================================================================
// Squeeze-excite FCs: gate = hsig(W2 * relu(W1 * m + b1) + b2); one workgroup per sample.
// Weights are in logical order; m / gate are [N][Cs] physical.
// =====================================================================================
__global__ void __launch_bounds__(256) sefc_kernel(const SeArgs a) {
  extern __shared__ float sm[];
  float* m = sm;            // [C] logical
  float* hbuf = sm + a.C;   // [R]
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < a.C; c += 256) m[c] = a.in[(long)n * a.Cs + c8i_phys(c)];
  __syncthreads();
  for (int j = threadIdx.x; j < a.R; j += 256) {
    const float* w = a.w1 + (long)j * a.C;
    float acc = 0.f;
    for (int c = 0; c < a.C; ++c) acc = fmaf(m[c], w[c], acc);
    acc = acc + a.b1[j];
    hbuf[j] = fmaxf(acc, 0.0f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < a.Cs; c += 256) {
    const int lc = c8i_logical(c);
    float g = 0.f;
    if (lc < a.C) {
      const float* w = a.w2 + (long)lc * a.R;
      float acc = 0.f;
      for (int j = 0; j < a.R; ++j) acc = fmaf(hbuf[j], w[j], acc);
      acc = acc + a.b2[lc];
      float t = acc * a.slope;
      t = t + a.offset;
      g = fminf(fmaxf(t, 0.0f), 1.0f);
    }
    a.out[(long)n * a.Cs + c] = g;
  }
}
void launch_sefc(const SeArgs& a, int N, hipStream_t s) {
  hipLaunchKernelGGL(sefc_kernel, dim3(N), dim3(256), (a.C + a.R) * sizeof(float), s, a);
}
#endif  // OCR_TU_H16

// =====================================================================================
// Concat with per-source nearest upsampling (FPN fuse; rec neck concat).
// =====================================================================================
__global__ void __launch_bounds__(256) concat_kernel(const ConcatArgs a) {
  const long t = (long)xcd_swizzle(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  const int c4n = a.Cs >> 2;
  if (t >= a.M * c4n) return;
  const long m = t / c4n;
  const int pc = (int)(t - m * c4n) * 4;
  int n, y, x;
  int j = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.nsrc && pc >= a.coff[i]) j = i;
  const int up = a.up[j];
  long spix;
  if (a.rout.h) {  // ragged batch of images: source j lives log2(up) levels coarser than the output
    int w;
    rag_decompose(a.rout, a.N, a.H, m, (long)xcd_swizzle(blockIdx.x, gridDim.x) * 256 / c4n, n, y, x, w);
    const int lu = 31 - __clz(up);
    spix = (long)(a.rout.cw[n] >> (2 * (a.rout.shift + lu))) + (long)(y / up) * (a.rout.w[n] >> (a.rout.shift + lu)) + x / up;
  } else {
    decompose(m, a.H * a.W, a.W, n, y, x);
    const int sh = a.H / up, sw = a.W / up;
    spix = ((long)n * sh + y / up) * sw + x / up;
  }
  if constexpr (kH16) {  // (a copy: the halfs travel as they are)
    *(uint2*)((_Float16*)a.out + m * a.Cs + pc) = *(const uint2*)((const _Float16*)a.src[j] + spix * a.scs[j] + (pc - a.coff[j]));
  } else {
    const float4 v = *(const float4*)(a.src[j] + spix * a.scs[j] + (pc - a.coff[j]));
    *(float4*)(a.out + m * a.Cs + pc) = v;
  }
}
void OCR_L(launch_concat)(const ConcatArgs& a, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_concat_h16(a, s))
  const long total = a.M * (a.Cs >> 2);
  hipLaunchKernelGGL(concat_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// =====================================================================================
// Pooling without padding (rec avg k(3,2)s(3,2) incl. the H=2 truncation quirk; cls max 2x2).
// =====================================================================================
__global__ void __launch_bounds__(256) pool_kernel(const PoolArgs a) {
  const long t_wg = (long)xcd_swizzle(blockIdx.x, gridDim.x) * 256;
  const long t = t_wg + threadIdx.x;
  const int c4n = a.Cs >> 2;
  if (t >= a.M * c4n) return;
  const long m = t / c4n;
  const int pc = (int)(t - m * c4n) * 4;
  int n, y, x;
  int IW = a.W;
  long ipix0;
  if (a.rout.w) {  // ragged batch: the line's own widths
    int ow;
    rag_decompose(a.rout, a.N, a.OH, m, t_wg / c4n, n, y, x, ow);
    IW = a.rin.w[n];
    ipix0 = (long)a.rin.cw[n] * a.H;
  } else {
    decompose(m, a.OH * a.OW, a.OW, n, y, x);
    ipix0 = (long)n * a.H * a.W;
  }
  float4 acc = a.is_max ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0.f, 0.f, 0.f, 0.f);
  int cnt = 0;
  for (int dy = 0; dy < a.KH; ++dy)
    for (int dx = 0; dx < a.KW; ++dx) {
      const int iy = y * a.SH + dy, ix = x * a.SW + dx;
      if (iy >= a.H || ix >= IW) continue;
      const float4 v = ld4<kH16>(a.in, (ipix0 + (long)iy * IW + ix) * a.Cs + pc);
      if (a.is_max) {
        acc.x = fmaxf(acc.x, v.x); acc.y = fmaxf(acc.y, v.y); acc.z = fmaxf(acc.z, v.z); acc.w = fmaxf(acc.w, v.w);
      } else {
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w;
      }
      ++cnt;
    }
  if (!a.is_max) {
    const float d = (float)cnt;
    acc.x = acc.x / d; acc.y = acc.y / d; acc.z = acc.z / d; acc.w = acc.w / d;
  }
  st4<kH16>(a.out, m * a.Cs + pc, acc);
}
void OCR_L(launch_pool)(const PoolArgs& a, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_pool_h16(a, s))
  const long total = a.M * (a.Cs >> 2);
  hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

// =====================================================================================
// LayerNorm over C logical channels, one thread per token (sequential sums: the contract's order).
// =====================================================================================
__global__ void __launch_bounds__(64) ln_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int C,
                                                int Cs, float eps, const float* __restrict__ g,
                                                const float* __restrict__ b) {
  const long r = (long)blockIdx.x * 64 + threadIdx.x;
  if (r >= rows) return;
  const long src = r * Cs;  // element index of the token's first channel
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = s + ld1<kH16>(in, src + c8i_phys(c));
  const float mean = s / (float)C;
  float v = 0.f;
  for (int c = 0; c < C; ++c) {
    const float xm = ld1<kH16>(in, src + c8i_phys(c)) - mean;
    v = fmaf(xm, xm, v);
  }
  const float var = v / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int c = 0; c < C; ++c) {
    const int pc = c8i_phys(c);
    const float xm = ld1<kH16>(in, src + pc) - mean;
    float t = xm * rstd;
    t = t * g[c];
    st1<kH16>(out, src + pc, t + b[c]);
  }
  for (int pc = 0; pc < Cs; ++pc)
    if (c8i_logical(pc) >= C) st1<kH16>(out, src + pc, 0.f);
}
// Same arithmetic, memory access restructured for the channel counts without pad channels (C == Cs,
// C % 8 == 0; 120 on the rec path): a workgroup moves 64 whole rows - one contiguous span of global
// memory - through LDS with full-line loads and stores, and each lane walks its own row out of
// registers.  The per-row chains (ascending logical channel) are exactly those of ln_kernel; only who
// fetches which byte changed.  ln_kernel's one-row-per-lane global reads were a dependent chain of
// strided loads: ~120 us for ANY row count (rocprof r1: 105-250 us per call).
template <int C>
__global__ void __launch_bounds__(64) ln_tile_kernel(const float* __restrict__ in, float* __restrict__ out, long rows,
                                                     float eps, const float* __restrict__ g,
                                                     const float* __restrict__ b) {
  constexpr int LS = C + 1;  // odd row stride: a column walk across lanes touches every bank once
  __shared__ float s_x[64 * LS];
  const int lane = threadIdx.x;
  const long r0 = (long)blockIdx.x * 64;
  const int nrows = (int)(rows - r0 < 64 ? rows - r0 : 64);
  const int nvec = nrows * (C / 4);
#pragma unroll 6
  for (int i = lane; i < nvec; i += 64) {
    const float4 v = ld4<kH16>(in, r0 * C + 4L * i);
    const int row = i / (C / 4), col = (i - row * (C / 4)) * 4;
    float* d = s_x + row * LS + col;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  if (lane < nrows) {
    float x[C];  // physical (octet-interleaved) order
    float* row = s_x + lane * LS;
#pragma unroll
    for (int pc = 0; pc < C; ++pc) x[pc] = row[pc];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) s = s + x[c8i_phys(c)];
    const float mean = s / (float)C;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float xm = x[c8i_phys(c)] - mean;
      v = fmaf(xm, xm, v);
    }
    const float var = v / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int pc = c8i_phys(c);
      const float xm = x[pc] - mean;
      float t = xm * rstd;
      t = t * g[c];
      row[pc] = t + b[c];
    }
  }
  __syncthreads();
#pragma unroll 6
  for (int i = lane; i < nvec; i += 64) {
    const int row = i / (C / 4), col = (i - row * (C / 4)) * 4;
    const float* d = s_x + row * LS + col;
    st4<kH16>(out, r0 * C + 4L * i, make_float4(d[0], d[1], d[2], d[3]));
  }
}

void OCR_L(launch_ln)(const float* in, float* out, long rows, int C, int Cs, float eps, const float* g, const float* b,
                      hipStream_t s, bool h16) {
  OCR_H16_TWIN(h16, launch_ln_h16(in, out, rows, C, Cs, eps, g, b, s, h16))
  if (C == 120 && Cs == 120) {
    hipLaunchKernelGGL(ln_tile_kernel<120>, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, in, out, rows, eps, g, b);
    return;
  }
  hipLaunchKernelGGL(ln_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, s, in, out, rows, C, Cs, eps, g, b);
}

// =====================================================================================
// Multi-head self-attention of the SVTR neck (T <= a few hundred, head dim 15).
// One thread = one (n, head, query); three passes over the keys recompute q.k so nothing is
// staged (max, sum of exp, P.V) — identical values each pass, so the contract's order holds.
// =====================================================================================
template <int HD>
__global__ void __launch_bounds__(64) attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int T,
                                                  int heads, int Cs_in, int Cs_out, float scale) {
  const long gid = (long)blockIdx.x * 64 + threadIdx.x;
  if (gid >= (long)N * heads * T) return;
  const int t = (int)(gid % T);
  const int hh = (int)((gid / T) % heads);
  const int n = (int)(gid / ((long)T * heads));
  const int D = heads * HD;
  const long base = (long)n * T * Cs_in;  // element index of the line's first token
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] = ld1<kH16>(qkv, base + (long)t * Cs_in + c8i_phys(hh * HD + d)) * scale;
  int kp[HD], vp[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    kp[d] = c8i_phys(D + hh * HD + d);
    vp[d] = c8i_phys(2 * D + hh * HD + d);
  }
  float mx = -INFINITY;
  for (int u = 0; u < T; ++u) {
    const long kr = base + (long)u * Cs_in;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc = fmaf(q[d], ld1<kH16>(qkv, kr + kp[d]), acc);
    mx = fmaxf(mx, acc);
  }
  float sum = 0.f;
  for (int u = 0; u < T; ++u) {
    const long kr = base + (long)u * Cs_in;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc = fmaf(q[d], ld1<kH16>(qkv, kr + kp[d]), acc);
    sum = sum + ocr_expf(acc - mx);
  }
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int u = 0; u < T; ++u) {
    const long kr = base + (long)u * Cs_in;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc = fmaf(q[d], ld1<kH16>(qkv, kr + kp[d]), acc);
    const float e = ocr_expf(acc - mx);
    const float pw = e / sum;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = fmaf(pw, ld1<kH16>(qkv, kr + vp[d]), o[d]);
  }
  const long dst = ((long)n * T + t) * Cs_out;
#pragma unroll
  for (int d = 0; d < HD; ++d) st1<kH16>(out, dst + c8i_phys(hh * HD + d), o[d]);
}
// Same chains with the operands staged: one wave per (line n, head).  K and V rows of the head sit in
// LDS (16-float rows, every lane reads the same row: broadcast), the scaled q row in registers, and the
// score of (query, key) is computed once and parked in LDS ([key][lane]: conflict-free) instead of being
// recomputed in each of the three passes.  Values and their order are those of attn_kernel.
// attn_kernel's per-thread global walks cost 0.3-1.1 ms per call regardless of size (rocprof r1).
template <int HD>
__global__ void __launch_bounds__(64) attn_lds_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int T,
                                                      int heads, int Cs_in, int Cs_out, float scale, const RagLevel rag) {
  static_assert(HD <= 16, "head rows are padded to 16 floats");
  extern __shared__ float s_att[];
  const int lane = threadIdx.x;
  const int hh = blockIdx.x % heads, n = blockIdx.x / heads;
  long row0 = (long)n * T;            // the line's first token
  if (rag.w) { T = rag.w[n]; row0 = rag.cw[n]; }  // ragged batch: the line's own length (LDS is sized for the longest)
  float* s_k = s_att;                 // [T][16]
  float* s_v = s_k + (long)T * 16;    // [T][16]
  float* s_q = s_v + (long)T * 16;    // [64][17]
  float* s_e = s_q + 64 * 17;         // [T][64]
  const int D = heads * HD;
  const long base = row0 * Cs_in;  // element index of the line's first token
  for (int i = lane; i < T * 32; i += 64) {
    const int u = i >> 5, which = (i >> 4) & 1, d = i & 15;
    float val = 0.f;
    if (d < HD) val = ld1<kH16>(qkv, base + (long)u * Cs_in + c8i_phys((1 + which) * D + hh * HD + d));
    (which ? s_v : s_k)[u * 16 + d] = val;
  }
  for (int t0 = 0; t0 < T; t0 += 64) {
    __syncthreads();  // K/V staged (first chunk); previous chunk's q rows consumed (later chunks)
    for (int i = lane; i < 64 * 16; i += 64) {
      const int tq = i >> 4, d = i & 15;
      if (d < HD && t0 + tq < T) s_q[tq * 17 + d] = ld1<kH16>(qkv, base + (long)(t0 + tq) * Cs_in + c8i_phys(hh * HD + d));
    }
    __syncthreads();
    const int t = t0 + lane;
    if (t < T) {
      float q[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) q[d] = s_q[lane * 17 + d] * scale;
      float mx = -INFINITY;
      for (int u = 0; u < T; ++u) {
        float kr[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) *(float4*)(kr + 4 * j) = *(const float4*)(s_k + u * 16 + 4 * j);
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc = fmaf(q[d], kr[d], acc);
        s_e[u * 64 + lane] = acc;
        mx = fmaxf(mx, acc);
      }
      float sum = 0.f;
      for (int u = 0; u < T; ++u) {
        const float e = ocr_expf(s_e[u * 64 + lane] - mx);
        sum = sum + e;
        s_e[u * 64 + lane] = e;
      }
      float o[HD];
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = 0.f;
      for (int u = 0; u < T; ++u) {
        float vr[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) *(float4*)(vr + 4 * j) = *(const float4*)(s_v + u * 16 + 4 * j);
        const float pw = s_e[u * 64 + lane] / sum;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = fmaf(pw, vr[d], o[d]);
      }
      const long dst = (row0 + t) * Cs_out;
#pragma unroll
      for (int d = 0; d < HD; ++d) st1<kH16>(out, dst + c8i_phys(hh * HD + d), o[d]);
    }
  }
}

// The same chains once more, a workgroup per LINE: eight waves = the eight heads.  attn_lds_kernel's wave per (line, head)
// stages its head's K / V / Q values with 4-byte loads (36 per lane, two load -> barrier round trips) and keeps one line's
// scores in LDS; 16 384 such workgroups took 0.24 ms per launch however the score chains were scheduled.  Here the line's
// [T][Cs_in] block is read ONCE with 16-byte loads by all 512 threads and scattered into per-head K [T][16], V [T][16] and
// Q [16][T] arrays (Q transposed: a lane reads its own query, consecutive lanes consecutive words), a lane's T <= TMAX scores
// stay in registers, and every value, chain and order is attn_kernel's: score = fma chain over d ascending from 0, max,
// e = exp(score - max) summed in key order, p = e / sum, out = fma chain over keys ascending.
template <int HD, int HEADS, int TMAX>
__global__ void __launch_bounds__(HEADS * 64) attn_line_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N, int T,
                                                               int Cs_in, int Cs_out, float scale, const RagLevel rag, const int TP) {
  static_assert(HD <= 16, "head rows are padded to 16 floats");
  extern __shared__ float s_att[];
  const int tid = threadIdx.x, lane = tid & 63, hh = tid >> 6;
  const int n = blockIdx.x;
  long row0 = (long)n * T;                        // the line's first token
  if (rag.w) { T = rag.w[n]; row0 = rag.cw[n]; }  // ragged batch: the line's own length (TP = the longest line's)
  float* s_k = s_att;                             // [HEADS][TP][16]
  float* s_v = s_k + HEADS * TP * 16;             // [HEADS][TP][16]
  float* s_q = s_v + HEADS * TP * 16;             // [HEADS][16][TP]
  constexpr int D = HEADS * HD;
  const int q4n = Cs_in >> 2;
  for (int i = tid; i < T * q4n; i += HEADS * 64) {
    const int u = i / q4n, qd = i - u * q4n;
    const float4 v4 = ld4<kH16>(qkv, (row0 + u) * Cs_in + 4 * qd);
    const float vals[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lc = c8i_logical(4 * qd + j);  // logical channel: q | k | v, each D = heads x HD wide
      if (lc >= 3 * D) continue;               // (pad channels of the octet layout)
      const int which = lc / D, r = lc - which * D;
      const int head = r / HD, d = r - head * HD;
      if (which == 0) s_q[(head * 16 + d) * TP + u] = vals[j];
      else (which == 1 ? s_k : s_v)[(head * TP + u) * 16 + d] = vals[j];
    }
  }
  __syncthreads();
  const int t = lane;
  if (t >= T) return;
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] = s_q[(hh * 16 + d) * TP + t] * scale;
  const float* kh = s_k + hh * TP * 16;
  const float* vh = s_v + hh * TP * 16;
  float e[TMAX];
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < TMAX; ++u) {
    if (u < T) {  // (uniform)
      float kr[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) *(float4*)(kr + 4 * j) = *(const float4*)(kh + u * 16 + 4 * j);
      float acc = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc = fmaf(q[d], kr[d], acc);
      e[u] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < TMAX; ++u) {
    if (u < T) {
      e[u] = ocr_expf(e[u] - mx);
      sum = sum + e[u];
    }
  }
  float o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int u = 0; u < TMAX; ++u) {
    if (u < T) {
      float vr[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) *(float4*)(vr + 4 * j) = *(const float4*)(vh + u * 16 + 4 * j);
      const float pw = e[u] / sum;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = fmaf(pw, vr[d], o[d]);
    }
  }
  const long dst = (row0 + t) * Cs_out;
#pragma unroll
  for (int d = 0; d < HD; ++d) st1<kH16>(out, dst + c8i_phys(hh * HD + d), o[d]);
}

#ifndef OCR_TU_H16
bool attn_ragged_fits(int T) { return ((size_t)T * 96 + 64 * 17) * sizeof(float) <= 150 * 1024; }
#endif
void OCR_L(launch_attn)(const float* qkv, float* out, int N, int T, int heads, int hd, int Cs_in, int Cs_out, float scale,
                        hipStream_t s, RagLevel rag, bool h16) {
  OCR_H16_TWIN(h16, launch_attn_h16(qkv, out, N, T, heads, hd, Cs_in, Cs_out, scale, s, rag, h16))
  // lines of at most 64 tokens (the recognizer's 320-pixel lines have 40): a workgroup per line, a wave per head
  // (OCR_ATTN_LINE=0: the wave-per-(line, head) kernel below, A/B; results are identical)
  if (rt_options().attn_line && heads == 8 && hd == 15 && T <= 64 && T > 0) {
    const size_t ldsl = (size_t)3 * 8 * T * 16 * sizeof(float);
    static LdsAttrMemo attr_line;
    if (ldsl <= 64 * 1024 || raise_dynamic_lds((const void*)attn_line_kernel<15, 8, 64>, 100 * 1024, attr_line)) {
      hipLaunchKernelGGL((attn_line_kernel<15, 8, 64>), dim3((unsigned)N), dim3(512), ldsl, s, qkv, out, N, T, Cs_in, Cs_out, scale, rag, T);
      return;
    }
  }
  const size_t lds = ((size_t)T * 96 + 64 * 17) * sizeof(float);
  if (lds <= 150 * 1024) {
    static LdsAttrMemo attr_state;  // per device: the pool drives several from one process
    if (lds <= 64 * 1024 || raise_dynamic_lds((const void*)attn_lds_kernel<15>, 150 * 1024, attr_state)) {
      hipLaunchKernelGGL(attn_lds_kernel<15>, dim3((unsigned)(N * heads)), dim3(64), lds, s, qkv, out, N, T, heads, Cs_in,
                         Cs_out, scale, rag);
      return;
    }
  }
  if (rag.w) { fprintf(stderr, "launch_attn: a ragged batch needs the LDS form (attn_ragged_fits)\n"); return; }
  const long total = (long)N * heads * T;
  // hd is 15 for the only attention block on the path (rec plan); the runtime checks it at load.
  hipLaunchKernelGGL(attn_kernel<15>, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, qkv, out, N, T, heads, Cs_in,
                     Cs_out, scale);
}

#ifndef OCR_TU_H16  // (plain f32 logits and partials in both precisions)
// =====================================================================================
// Row softmax over plain [rows][C] + greedy-CTC inputs (arg max / max prob per row).
// Canonical order (the same in conv_finish<OUT_HEAD> + head_combine_kernel and in
// the oracle): columns in groups of 128; per group its max m_g and s_g = chain0 + chain1, chain h = the
// columns with ((c >> 2) & 1) == h summed in ascending order of exp(x_c - m_g); per row M = max_g m_g,
// S = sum over ascending g of s_g * exp(m_g - M); p_c = exp(x_c - M) / S; arg max = first maximum of the
// LOGITS, its probability exp(0) / S.
// This kernel is the unfused form (probabilities requested: parity taps, the 2-class cls head):
// one wave per row, lane l walks chains l and l + 64 of the 2 * ceil(C/128) chains.
// =====================================================================================
__global__ void __launch_bounds__(256) softmax_argmax_kernel(const float* __restrict__ logits, float* __restrict__ probs,
                                                             int* __restrict__ amax, float* __restrict__ pmax, long rows,
                                                             int C) {
  __shared__ float s_m[4][128], s_s[4][128];  // per wave: group max, chain sums (up to 64 groups = 8192 columns)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r = (long)blockIdx.x * 4 + wv;
  if (r >= rows) return;
  const float* src = logits + r * C;
  const int G = (C + 127) >> 7;
  for (int ch = lane; ch < 2 * G; ch += 64) {
    const int g = ch >> 1, h = ch & 1, c0 = g << 7;
    float mx = -INFINITY;
    for (int c = c0; c < min(c0 + 128, C); ++c) mx = fmaxf(mx, src[c]);
    float part = 0.f;
    for (int c = c0; c < min(c0 + 128, C); ++c)
      if (((c >> 2) & 1) == h) part = part + ocr_expf(src[c] - mx);
    s_m[wv][g] = mx;  // both chains of a group write the same value
    s_s[wv][ch] = part;
  }
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  float M = -INFINITY;
  for (int g = 0; g < G; ++g) M = fmaxf(M, s_m[wv][g]);
  float S = 0.f;
  for (int g = 0; g < G; ++g) {
    const float sg = s_s[wv][2 * g] + s_s[wv][2 * g + 1];
    const float t = sg * ocr_expf(s_m[wv][g] - M);
    S = S + t;
  }
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float x = src[c];
    if (probs) probs[r * C + c] = ocr_expf(x - M) / S;
    if (x > best) { best = x; bi = c; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) {
    if (amax) amax[r] = bi;
    if (pmax) pmax[r] = ocr_expf(best - M) / S;
  }
}
void launch_softmax_argmax(const float* logits, float* probs, int* amax, float* pmax, long rows, int C, hipStream_t s) {
  hipLaunchKernelGGL(softmax_argmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, probs, amax, pmax,
                     rows, C);
}

// Fused head, second half: one thread per row folds the per-group partials in ascending group order.
__global__ void __launch_bounds__(256) head_combine_kernel(const float* __restrict__ hmax, const float* __restrict__ hsum,
                                                           const int* __restrict__ hidx, long rows, int G,
                                                           int* __restrict__ amax, float* __restrict__ pmax) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* m = hmax + r * G;
  const float* sg = hsum + r * G;
  const int* ix = hidx + r * G;
  float M = -INFINITY;
  int bi = 0x7fffffff;
  for (int g = 0; g < G; ++g)
    if (m[g] > M) { M = m[g]; bi = ix[g]; }  // ascending groups: the first maximum wins
  float S = 0.f;
  for (int g = 0; g < G; ++g) {
    const float t = sg[g] * ocr_expf(m[g] - M);
    S = S + t;
  }
  if (amax) amax[r] = bi;
  if (pmax) pmax[r] = ocr_expf(M - M) / S;
}
void launch_head_combine(const float* hmax, const float* hsum, const int* hidx, long rows, int groups, int* amax, float* pmax,
                         hipStream_t s) {
  hipLaunchKernelGGL(head_combine_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, hmax, hsum, hidx, rows, groups,
                     amax, pmax);
}
#endif  // OCR_TU_H16

// =====================================================================================
// DB head tail: deconv 2x2 s2 (Cin -> 1) + bias + sigmoid, fused with the reference's
// `cbuf[i] = (uchar)(p*255)` and 8-bit threshold (/root/reference/src/ocr_det.cpp:143-154):
// writes the f32 probability map and the {0,1} bitmap in one pass.
// One thread = one input pixel -> 4 output pixels.
// =====================================================================================
__global__ void __launch_bounds__(256) det_tail_kernel(const DetTailArgs a) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  int n, y, x;
  decompose(m, a.H * a.W, a.W, n, y, x);
  const long src = m * a.Cs;  // element index of the pixel's first channel
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // The pixel's channels arrive as two 16-byte loads per octet (C8I: evens, then odds) and are consumed in
  // ascending logical order c = 8o, 8o+1, ... - the same fma chain as a channel-by-channel walk.
  auto octet = [&](int o8, const float4& ev, const float4& od) {
    const float v[8] = {ev.x, od.x, ev.y, od.y, ev.z, od.z, ev.w, od.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = o8 + j;
      if (c < a.C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = fmaf(v[j], a.w[c * 4 + q], acc[q]);
      }
    }
  };
  if (a.Cs == 24) {  // the DB head: the pixel's six 16-byte loads in flight before the first FMA (same chain order)
    float4 r[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) r[k] = ld4<kH16>(a.in, src + 4 * k);
#pragma unroll
    for (int k = 0; k < 3; ++k) octet(8 * k, r[2 * k], r[2 * k + 1]);
  } else {
    for (int o8 = 0; o8 < a.Cs; o8 += 8) octet(o8, ld4<kH16>(a.in, src + o8), ld4<kH16>(a.in, src + o8 + 4));
  }
  float pr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float v = acc[q] + a.bias;
    const float e = ocr_expf(-v);
    const float d = 1.0f + e;
    pr[q] = 1.0f / d;
  }
  // rows 2y and 2y+1, columns 2x and 2x+1: one 8-byte store per row (2-byte for the bitmap)
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const long o = ((long)n * 2 * a.H + 2 * y + dy) * (2 * a.W) + 2 * x;
    *(float2*)(a.prob + o) = make_float2(pr[2 * dy], pr[2 * dy + 1]);
    if (a.bitmap) {
      // (unsigned char)(p*255): truncation; p in [0,1]
      const int b0 = (int)(pr[2 * dy] * 255.0f) > a.ithresh ? 1 : 0, b1 = (int)(pr[2 * dy + 1] * 255.0f) > a.ithresh ? 1 : 0;
      *(unsigned short*)(a.bitmap + o) = (unsigned short)(b0 | (b1 << 8));
    }
  }
}
void OCR_L(launch_det_tail)(const DetTailArgs& a, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_det_tail_h16(a, s))
  hipLaunchKernelGGL(det_tail_kernel, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, s, a);
}

// =====================================================================================
// DB head, both transposed convs in one kernel (DbHeadArgs, kernels_net.h).  VALU: 24 x 96 + 96 x 4 fmas per input
// pixel with wave-uniform weights (scalar loads), against 2 x 1.4 GB of traffic for the tensor it removes.
// =====================================================================================
template <int C>
__global__ void __launch_bounds__(256) db_head_kernel(const DbHeadArgs a) {
  static_assert(C % 8 == 0, "whole octets");
  // Weights are wave-uniform scalar operands (two adjacent output channels per v_pk_fma_f32).  Measured alternatives:
  // weights in LDS read as 16-byte broadcasts 4.3 ms (every read sunk next to its FMA: one LDS round trip per two
  // FMAs), scalar loads fenced per channel pair 0.9 ms (a wait in front of every pair); left to the scheduler the
  // scalar loads are hoisted and partly parked in VGPR lanes, and that is still the fastest form (0.67 ms).
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  int n, y, x;
  long OW = 4L * a.W, opix0;  // width / first pixel of the pixel's probability map
  if (a.rin.h) {  // ragged batch of images: the map is two levels finer than this tensor
    int w;
    rag_decompose(a.rin, a.N, a.H, m, (long)blockIdx.x * 256, n, y, x, w);
    OW = 4L * w;
    opix0 = (long)(a.rin.cw[n] >> (2 * (a.rin.shift - 2)));
  } else {
    decompose(m, a.H * a.W, a.W, n, y, x);
    opix0 = (long)n * 16 * a.H * a.W;
  }
  // the pixel's channels in LOGICAL order (C8I: an octet's evens, then its odds), all loads in flight at once
  float4 r[C / 4];
#pragma unroll
  for (int k = 0; k < C / 4; ++k) r[k] = ld4<kH16>(a.in, m * a.Cs + 4 * k);
  float xl[C];
#pragma unroll
  for (int o = 0; o < C / 8; ++o) {
    const float4 ev = r[2 * o], od = r[2 * o + 1];
    xl[8 * o] = ev.x; xl[8 * o + 1] = od.x; xl[8 * o + 2] = ev.y; xl[8 * o + 3] = od.y;
    xl[8 * o + 4] = ev.z; xl[8 * o + 5] = od.z; xl[8 * o + 6] = ev.w; xl[8 * o + 7] = od.w;
  }
  const long obase = opix0 + (long)(4 * y) * OW + 4 * x;
  float pr[4][4];  // [row of the 4x4 block][column]
  // (two adjacent output channels per instruction: each lane of a v_pk_fma_f32 is the same IEEE fma as the scalar one)
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // quadrant of the first deconv: rows 2*(q>>1).., columns 2*(q&1)..
    float hq[C];
#pragma unroll
    for (int c = 0; c < C; c += 2) {
      ocr_f2 acc = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const float* w = a.w1 + (k * 4 + q) * C + c;
        acc = __builtin_elementwise_fma(ocr_f2{xl[k], xl[k]}, ocr_f2{w[0], w[1]}, acc);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pc = c8i_phys(c + j);
        float v = (j ? acc.y : acc.x) + a.bias1[pc];
        const float u = v * a.bn_s[pc];
        v = u + a.bn_t[pc];
        hq[c + j] = fmaxf(v, 0.0f);
      }
    }
    ocr_f2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};  // the second deconv's four outputs of this quadrant
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float* w = a.w2 + c * 4;
      o01 = __builtin_elementwise_fma(ocr_f2{hq[c], hq[c]}, ocr_f2{w[0], w[1]}, o01);
      o23 = __builtin_elementwise_fma(ocr_f2{hq[c], hq[c]}, ocr_f2{w[2], w[3]}, o23);
    }
    const float o4[4] = {o01.x, o01.y, o23.x, o23.y};
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      const float v = o4[q2] + a.bias2;
      const float e = ocr_expf(-v);
      const float d = 1.0f + e;
      pr[2 * (q >> 1) + (q2 >> 1)][2 * (q & 1) + (q2 & 1)] = 1.0f / d;
    }
  }
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    const long o = obase + row * OW;
    *(float4*)(a.prob + o) = make_float4(pr[row][0], pr[row][1], pr[row][2], pr[row][3]);
    if (a.bitmap) {
      unsigned bits = 0;  // (unsigned char)(p*255): truncation; p in [0,1]
#pragma unroll
      for (int col = 0; col < 4; ++col) bits |= ((int)(pr[row][col] * 255.0f) > a.ithresh ? 1u : 0u) << (8 * col);
      *(unsigned*)(a.bitmap + o) = bits;
    }
  }
}
// The same head with the first transposed conv on the matrix cores (a.wfrag): a wave takes 32 input pixels, lanes (p, 0)
// and (p, 1) of pixel p share its 96 first-stage values.  The columns of the weight image are permuted (net.hip, "dbhf:")
// so that lane (p, h) ends up with exactly quadrants 2h and 2h+1 - rows 2h, 2h+1 of the pixel's 4x4 block of the map -
// in its own accumulators: value l = 24*(q & 1) + physical channel sits in acc[l / 16][l % 16], and the second stage
// needs no exchange between lanes.  Each value's chain is the matrix-core deconv's (k ascending in logical order, two per
// instruction), the stages after it are db_head_kernel's: same bits.
template <bool H16>
__global__ void __launch_bounds__(256) db_head_mfma_kernel(const DbHeadArgs a) {
  constexpr int C = 24;
  const int lane = threadIdx.x & 63, p = lane & 31, h = lane >> 5;
  const long m0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
  if (m0 >= a.M) return;  // the whole wave
  const bool live = m0 + p < a.M;
  const long m = live ? m0 + p : a.M - 1;
  float4 av[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) av[j] = ld4<H16>(a.in, m * C + 8 * j + 4 * h);
  const float4* __restrict__ wf = (const float4*)a.wfrag + lane;
  floatx16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float4 bv[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) bv[t] = wf[(j * 3 + t) * 64];
#define OCR_H_SWEEP(X)                                                                                                    \
  _Pragma("unroll") for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[t].X, av[j].X, acc[t], 0, 0, 0);
    OCR_H_SWEEP(x) OCR_H_SWEEP(y) OCR_H_SWEEP(z) OCR_H_SWEEP(w)
#undef OCR_H_SWEEP
  }
  int n, y, x;
  long OW = 4L * a.W, opix0;
  if (a.rin.h) {
    int w;
    rag_decompose(a.rin, a.N, a.H, m, m0, n, y, x, w);
    OW = 4L * w;
    opix0 = (long)(a.rin.cw[n] >> (2 * (a.rin.shift - 2)));
  } else {
    decompose(m, a.H * a.W, a.W, n, y, x);
    opix0 = (long)n * 16 * a.H * a.W;
  }
  // bias, BN, relu on the lane's 48 values, in place (the per-channel vectors are in physical order, like the accumulators)
#pragma unroll
  for (int l = 0; l < 2 * C; ++l) {
    const int pc = l % C;
    float v = acc[l / 16][l % 16] + a.bias1[pc];
    const float u = v * a.bn_s[pc];
    v = u + a.bn_t[pc];
    acc[l / 16][l % 16] = fmaxf(v, 0.0f);
  }
  float pr[2][4];  // [row 2h + r of the 4x4 block][column]
#pragma unroll
  for (int ql = 0; ql < 2; ++ql) {  // quadrant 2h + ql: rows 2h.., columns 2ql..
    ocr_f2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int l = ql * C + c8i_phys(c);
      const float hv = acc[l / 16][l % 16];
      const float* w = a.w2 + c * 4;
      o01 = __builtin_elementwise_fma(ocr_f2{hv, hv}, ocr_f2{w[0], w[1]}, o01);
      o23 = __builtin_elementwise_fma(ocr_f2{hv, hv}, ocr_f2{w[2], w[3]}, o23);
    }
    const float o4[4] = {o01.x, o01.y, o23.x, o23.y};
#pragma unroll
    for (int q2 = 0; q2 < 4; ++q2) {
      const float v = o4[q2] + a.bias2;
      const float e = ocr_expf(-v);
      const float d = 1.0f + e;
      pr[q2 >> 1][2 * ql + (q2 & 1)] = 1.0f / d;
    }
  }
  if (!live) return;
  const long obase = opix0 + (long)(4 * y + 2 * h) * OW + 4 * x;
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const long o = obase + row * OW;
    *(float4*)(a.prob + o) = make_float4(pr[row][0], pr[row][1], pr[row][2], pr[row][3]);
    if (a.bitmap) {
      unsigned bits = 0;
#pragma unroll
      for (int col = 0; col < 4; ++col) bits |= ((int)(pr[row][col] * 255.0f) > a.ithresh ? 1u : 0u) << (8 * col);
      *(unsigned*)(a.bitmap + o) = bits;
    }
  }
}
bool OCR_L(launch_db_head)(const DbHeadArgs& a, int C, hipStream_t s) {
  OCR_H16_TWIN(a.h16, launch_db_head_h16(a, C, s))
  if (C != 24 || a.Cs != 24) return false;
  if (a.wfrag) {
    hipLaunchKernelGGL(db_head_mfma_kernel<kH16>, dim3((unsigned)((a.M + 127) / 128)), dim3(256), 0, s, a);
    return true;
  }
  hipLaunchKernelGGL(db_head_kernel<24>, dim3((unsigned)((a.M + 255) / 256)), dim3(256), 0, s, a);
  return true;
}

// ---- layout conversion taps (debug / parity): C8I [M][Cs] -> logical [M][C] ----
__global__ void __launch_bounds__(256) c8i_to_plain_kernel(const float* __restrict__ in, float* __restrict__ out, long M,
                                                           int C, int Cs) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * C) return;
  const long m = t / C;
  const int c = (int)(t - m * C);
  out[t] = ld1<kH16>(in, m * Cs + c8i_phys(c));
}
void OCR_L(launch_c8i_to_plain)(const float* in, float* out, long M, int C, int Cs, hipStream_t s, bool h16) {
  OCR_H16_TWIN(h16, launch_c8i_to_plain_h16(in, out, M, C, Cs, s, h16))
  hipLaunchKernelGGL(c8i_to_plain_kernel, dim3((unsigned)((M * C + 255) / 256)), dim3(256), 0, s, in, out, M, C, Cs);
}

#ifndef OCR_TU_H16
// ---- numerics probe (tests): does the hardware match the arithmetic contract? ----
__global__ void probe_kernel(const float* a, const float* b, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = a[i] / b[i];
  out[n + i] = sqrtf(fabsf(a[i]));
  out[2 * n + i] = ocr_expf(a[i]);
  out[3 * n + i] = fmaf(a[i], b[i], a[i]);
  float t = a[i] * b[i];
  out[4 * n + i] = t + a[i];
  out[5 * n + i] = rintf(a[i] * 1.44269504088896341f);
  // hard-swish as the epilogues compute it: the quad form (range guard + division-free path) with this
  // element and three neighbours, and the scalar contract form
  float q0 = a[i], q1 = a[(i + 1) % n], q2 = b[i], q3 = a[(i + 7) % n];
  ocr_hswish4(q0, q1, q2, q3);
  out[6 * n + i] = q0;
  out[7 * n + i] = q2;
}
void launch_probe(const float* a, const float* b, float* out, int n, hipStream_t s) {
  hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
}
#endif  // OCR_TU_H16

#ifdef OCR_TU_H16
}  // namespace h16
#endif
}  // namespace ocr
