// Expand 1x1 conv -> depthwise 5x5 conv of ONE small map per workgroup (gfx950): the inverted-residual blocks of the
// classifier (cls ops 19-60: conv 16 -> 88 | dw 5x5 | SE | conv 88 -> 16, ..., conv 32 -> 200 | dw 5x5 | SE | conv 200 -> 32)
// on their 3 x 96 and 2 x 96 maps.
//
// Launched one op at a time the expanded tensor is written once (1x1 conv) and read once (depthwise conv) - 5.5x / 6.25x
// the block's input - by launches of 40-190 us that are all latency (launches below 250 us).  Here the
// expanded tensor exists only in LDS, 32 channels at a time:
//   E   matrix cores -> LDS   a chunk = one 32-column tile of the 1x1 conv over the WHOLE map (9 or 6 pixel tiles dealt to
//                             the 8 waves): pixels straight from global memory as the MFMA operand (C8I: a lane's 16 bytes
//                             are four consecutive k steps), weights from the conv's fragment image, the accumulator through
//                             the conv's BN + hard-swish, then into the region R[quad][row][2 + x] - zero columns on both
//                             sides and one zero row stand for the depthwise conv's padding
//   W   LDS -> VALU           wave q owns channel quad q of the chunk, a lane one output COLUMN: every input row it needs is
//                             read once per kx and feeds all output rows (a 5x5 window on a 3-row map: 15 reads for 45
//                             real taps); the taps of an output arrive in (ky, kx) ascending order from 0, the contract's
//                             chain.  Rows above the map come first in that order and leave the accumulator at +0: skipped;
//                             rows below it are issued as fma(0, w, acc) (they turn a -0 accumulator into +0 exactly as the
//                             padded chain does).  BN + hard-swish, 16-byte stores to the depthwise conv's output tensor
//   row sums                  (when the SE pool follows) the wave writes its outputs back into its own quad's part of R and
//                             twelve lanes add a (row, channel) each left to right: the pool's first pass in the contract's
//                             order - what dw_conv_kernel's ROWSUM form leaves in gap_part_ - without a second read of the
//                             tensor.  (Storing the chunk from there as whole lines - all threads, lane = pixel x quad, one
//                             more barrier per chunk - measured 5 % SLOWER than the lanes' own 16-byte stores: not kept.)
// Arithmetic per value is the two launches' (same chains, same epilogue sequences): bit-identical
// (tests/test_gpu_parity.py; OCR_XDW=0 in the A/B test).  f32 contract only.
#include <hip/hip_runtime.h>

#include "conv_device.h"
#include "kernels_net.h"
#include "lds_attr.h"

namespace ocr {

namespace {

constexpr int XDW_K = 5, XDW_PAD = 2, XDW_MAXW = 96, XDW_RW = XDW_MAXW + XDW_K - 1, XDW_THREADS = 512, XDW_WAVES = 8;

__host__ __device__ constexpr size_t xdw_lds_bytes(int hin, int nchunk) { return (size_t)8 * (hin + 1) * XDW_RW * 16 + (size_t)nchunk * 2 * 32 * 4; }

// a wave-uniform quad of a parameter image as ONE scalar load (constant address space: nothing in this kernel writes the images)
typedef float xdw_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 xdw_ld_s(const float* p) {
  const xdw_f4 v = *(const __attribute__((address_space(4))) xdw_f4*)p;
  return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void xdw_bn(float4& v, const float4& sc, const float4& sh) {
  float t;
  t = v.x * sc.x; v.x = t + sh.x;
  t = v.y * sc.y; v.y = t + sh.y;
  t = v.z * sc.z; v.z = t + sh.z;
  t = v.w * sc.w; v.w = t + sh.w;
}
// hard-swish of NQ quads: ONE range pass and one branch for all of them (ocr_common.h: the division-free form is the IEEE
// quotient while every |y| is inside [2^-119, 2^125); anything else takes the division)
template <int NQ>
__device__ __forceinline__ void xdw_hswish(float4 (&v)[NQ]) {
  float mn = INFINITY, mx = 0.0f;
#pragma unroll
  for (int i = 0; i < NQ; ++i) { ocr_absrange(mn, mx, v[i].x, v[i].y); ocr_absrange(mn, mx, v[i].z, v[i].w); }
  if (ocr_hsw_fast_ok(mn, mx)) {
#pragma unroll
    for (int i = 0; i < NQ; ++i) ocr_hswish4_fast(v[i].x, v[i].y, v[i].z, v[i].w);
  } else {
#pragma unroll
    for (int i = 0; i < NQ; ++i) { v[i].x = ocr_hswish_div(v[i].x); v[i].y = ocr_hswish_div(v[i].y); v[i].z = ocr_hswish_div(v[i].z); v[i].w = ocr_hswish_div(v[i].w); }
  }
}

}  // namespace

template <int CI8, int HIN, int SH>
__global__ void __launch_bounds__(XDW_THREADS, 4) xdw_kernel(const XdwArgs a) {
  constexpr int K = XDW_K, PAD = XDW_PAD, RW = XDW_RW;
  constexpr int HOUT = (HIN + 2 * PAD - K) / SH + 1;
  constexpr int ROWS = HIN + 1;                      // + the zero row
  constexpr int VROWS = (HOUT - 1) * SH + K - PAD;   // input rows 0 .. VROWS-1 are touched by some output row (>= HIN: padding)
  constexpr int MAXT = (HIN * XDW_MAXW / 32 + XDW_WAVES - 1) / XDW_WAVES;  // pixel tiles of a wave
  constexpr int NPASS = (XDW_MAXW + 63) / 64;        // output columns of a lane
  extern __shared__ float4 s_xdw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int n = blockIdx.x;
  const int W = a.W, Cs_e = a.Cs_e;
  const int npi = HIN * W, npo = HOUT * W;
  const int nchunk = (Cs_e + 31) >> 5;
  const int ntile = (npi + 31) >> 5;
  float4* const R = s_xdw;                              // [8 quads][ROWS][RW]
  float* const s_e = (float*)(R + 8 * ROWS * RW);       // [chunk][BN scale | BN shift][32]: the 1x1 conv's
  const float* const xin = a.x + (long)n * npi * (CI8 * 8);
  float* const outn = a.out + (long)n * npo * Cs_e;
  // ---- once per workgroup: the zero frame of R, every chunk's parameters, the wave's pixel operands (the same for every
  // chunk), the first chunk's fragments - the chunk loop below waits for nothing that comes from global memory except
  // the fragments it asked for a chunk earlier
  for (int i = tid; i < 8 * ROWS * RW; i += XDW_THREADS) R[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = tid; i < nchunk * 64; i += XDW_THREADS) {
    const int c = (i >> 6) * 32 + (i & 31);
    s_e[i] = c < Cs_e ? ((i & 32) ? a.e_sh[c] : a.e_sc[c]) : 0.f;
  }
  float4 xv[MAXT][CI8];
#pragma unroll
  for (int u = 0; u < MAXT; ++u) {
    const int px = (wave + XDW_WAVES * u) * 32 + p;
#pragma unroll
    for (int j = 0; j < CI8; ++j)
      xv[u][j] = px < npi ? *(const float4*)(xin + (long)px * (CI8 * 8) + 8 * j + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 wf[CI8];
#pragma unroll
  for (int j = 0; j < CI8; ++j) wf[j] = ((const float4*)a.wfrag)[((long)j * a.NTtot) * 64 + lane];
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();  // the previous chunk's taps and row sums are done with R (first chunk: frame and parameters are in place)
    // ---- E: the chunk's 32 columns of the 1x1 conv over the whole map
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int tile = wave + XDW_WAVES * u;
      if (tile >= ntile) break;
      const int px = tile * 32 + p;
      floatx16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
      for (int j = 0; j < CI8; ++j) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].x, xv[u][j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].y, xv[u][j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].z, xv[u][j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j].w, xv[u][j].w, acc, 0, 0, 0);
      }
      const int y = px / W, x = px - y * W;
      const float* const se = s_e + ch * 64 + 4 * h;
      float4 ev[4];
      bool keep[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        keep[g] = px < npi && ch * 32 + 8 * g + 4 * h < Cs_e;  // (channels past the tensor: their waves skip the taps below)
        ev[g] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        xdw_bn(ev[g], *(const float4*)(se + 8 * g), *(const float4*)(se + 32 + 8 * g));
        if (!keep[g]) ev[g] = make_float4(1.f, 1.f, 1.f, 1.f);  // (keeps the quads that count on the division-free path)
      }
      xdw_hswish<4>(ev);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (keep[g]) R[((2 * g + h) * ROWS + y) * RW + x + PAD] = ev[g];
    }
    if (ch + 1 < nchunk) {  // the next chunk's fragments travel during the taps
#pragma unroll
      for (int j = 0; j < CI8; ++j) wf[j] = ((const float4*)a.wfrag)[((long)j * a.NTtot + ch + 1) * 64 + lane];
    }
    __syncthreads();
    // ---- W: depthwise taps, wave = channel quad, lane = output column
    if (ch * 32 + 4 * wave < Cs_e) {
      // the quad's taps and BN vectors are wave-uniform: scalar loads straight from the layer's images (a per-lane LDS read
      // of a uniform address costs the LDS a full 64-lane pass: 60 of them per pass made this phase LDS-bound)
      const int c0 = __builtin_amdgcn_readfirstlane(ch * 32 + 4 * wave);
      const float* __restrict__ const wq = a.dw_w + c0;
      float4* const Rq = R + wave * ROWS * RW;
      const float4 dsc = xdw_ld_s(a.d_sc + c0), dsh = xdw_ld_s(a.d_sh + c0);
      float4 res[NPASS][HOUT];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        if (ps * 64 >= W) break;
        const int ox = ps * 64 + lane;
        const float4* const col = Rq + (ox < W ? ox : W - 1);  // region column of tap kx = 0 (idle lanes read a valid one)
        ocr_f2 lo[HOUT], hi[HOUT];
#pragma unroll
        for (int o = 0; o < HOUT; ++o) { lo[o] = ocr_f2{0.f, 0.f}; hi[o] = ocr_f2{0.f, 0.f}; }
        // (the row loop stays a loop: unrolled, the compiler asks for every tap's scalar quad up front - 100 SGPRs - and spills)
#pragma unroll 1
        for (int iy = 0; iy < HIN; ++iy) {
          float4 v[K];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) v[kx] = col[iy * RW + kx];
#pragma unroll
          for (int o = 0; o < HOUT; ++o) {
            const int ky = iy - o * SH + PAD;  // uniform
            if (ky < 0 || ky >= K) continue;
            const float* const wrow = wq + (long)ky * K * Cs_e;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const float4 w = xdw_ld_s(wrow + (long)kx * Cs_e);
              lo[o] = __builtin_elementwise_fma(ocr_f2{v[kx].x, v[kx].y}, ocr_f2{w.x, w.y}, lo[o]);
              hi[o] = __builtin_elementwise_fma(ocr_f2{v[kx].z, v[kx].w}, ocr_f2{w.z, w.w}, hi[o]);
            }
          }
        }
        // the padding rows below the map: fma(0, w, acc) in the chain's order (compile-time rows: few)
#pragma unroll
        for (int iy = HIN; iy < VROWS; ++iy) {
#pragma unroll
          for (int o = 0; o < HOUT; ++o) {
            const int ky = iy - o * SH + PAD;
            if (ky < 0 || ky >= K) continue;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const float4 w = xdw_ld_s(wq + (long)(ky * K + kx) * Cs_e);
              lo[o] = __builtin_elementwise_fma(ocr_f2{0.f, 0.f}, ocr_f2{w.x, w.y}, lo[o]);
              hi[o] = __builtin_elementwise_fma(ocr_f2{0.f, 0.f}, ocr_f2{w.z, w.w}, hi[o]);
            }
          }
        }
#pragma unroll
        for (int o = 0; o < HOUT; ++o) {
          res[ps][o] = make_float4(lo[o].x, lo[o].y, hi[o].x, hi[o].y);
          xdw_bn(res[ps][o], dsc, dsh);
        }
        xdw_hswish<HOUT>(res[ps]);
        if (ox < W) {
#pragma unroll
          for (int o = 0; o < HOUT; ++o) *(float4*)(outn + ((long)o * W + ox) * Cs_e + c0) = res[ps][o];
        }
      }
      if (a.part) {
        // the pool's row sums of this quad: the wave's outputs go into ITS region (no other wave reads it; its own taps are
        // done), then one lane per (row, channel) adds its row left to right
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int ox = ps * 64 + lane;
          if (ox >= W) continue;
#pragma unroll
          for (int o = 0; o < HOUT; ++o) Rq[o * RW + PAD + ox] = res[ps][o];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 4 * HOUT) {
          const int r = lane >> 2, comp = lane & 3;
          const float* src = (const float*)(Rq + r * RW + PAD) + comp;
          float s = 0.f;
          for (int x = 0; x < W; ++x) s = s + src[4 * x];
          a.part[((long)n * HOUT + r) * Cs_e + ch * 32 + 4 * wave + comp] = s;
        }
      }
    }
  }
}

namespace {
template <int CI8, int HIN, int SH>
bool launch_xdw_one(const XdwArgs& a, hipStream_t s, bool query) {
  constexpr int HOUT = (HIN + 2 * XDW_PAD - XDW_K) / SH + 1;
  if (a.Hout != HOUT) return false;
  const size_t lds = xdw_lds_bytes(HIN, (a.Cs_e + 31) / 32);
  static LdsAttrMemo attr_state;
  if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)xdw_kernel<CI8, HIN, SH>, (int)lds, attr_state)) return false;
  if (lds > 80 * 1024) return false;  // (two workgroups per CU)
  if (query) return true;
  hipLaunchKernelGGL((xdw_kernel<CI8, HIN, SH>), dim3((unsigned)a.N), dim3(XDW_THREADS), lds, s, a);
  return true;
}
}  // namespace

// The shapes on this path: 1x1 conv from 16 or 32 stored channels into a multiple of 8, depthwise 5x5 pad 2 stride (1 | 2, 1)
// on uniform batches of 3- or 2-row maps at most 96 pixels wide.  query = true only asks.
bool launch_xdw(const XdwArgs& a, hipStream_t s, bool query) {
  if (a.K != XDW_K || a.PH != XDW_PAD || a.PW != XDW_PAD || a.SW != 1 || a.W < 1 || a.W > XDW_MAXW || a.N < 1) return false;
  if (a.Cs_e % 8 || a.Cs_e < 8 || a.NTtot * 32 < a.Cs_e) return false;
  if (a.Cs_in == 16) {
    if (a.Hin == 3 && a.SH == 1) return launch_xdw_one<2, 3, 1>(a, s, query);
    if (a.Hin == 3 && a.SH == 2) return launch_xdw_one<2, 3, 2>(a, s, query);
    if (a.Hin == 2 && a.SH == 1) return launch_xdw_one<2, 2, 1>(a, s, query);
  } else if (a.Cs_in == 32) {
    if (a.Hin == 3 && a.SH == 1) return launch_xdw_one<4, 3, 1>(a, s, query);
    if (a.Hin == 3 && a.SH == 2) return launch_xdw_one<4, 3, 2>(a, s, query);
    if (a.Hin == 2 && a.SH == 1) return launch_xdw_one<4, 2, 1>(a, s, query);
  }
  return false;
}

}  // namespace ocr
