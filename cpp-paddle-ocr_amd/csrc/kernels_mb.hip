// One MobileNetV3 bottleneck with squeeze-excite as ONE kernel (round 4: the classifier's small-launch tail).
//
//   conv 1x1 Cin -> Cexp + BN + act  ->  depthwise KxK (stride (SH, 1)) + BN + act  ->  global average pool
//   -> SE (fc -> relu -> fc -> hard sigmoid)  ->  x * gate  ->  conv 1x1 Cexp -> Cout + BN [+ residual]
//
// is six launches (and four tensors of Cexp channels) in the plan; the classifier (ch_ppocr_mobile_v2.0 cls, SURVEY.md
// A.3) runs eight such blocks on maps of 2 - 6 rows x 96 columns per text line, every launch 40 - 190 us of streaming
// 0.2 - 0.6 GB through HBM.  Here a workgroup owns ONE text line: the block's input (<= 24 KB) sits in LDS, the expanded
// tensor never exists as a whole - it is produced a CHUNK of CC channels at a time (expand -> depthwise are per-channel
// independent), twice:
//   pass A  chunk by chunk: expand, depthwise, the pool's row sums (x ascending from 0, the contract's order) -> SE gate
//   pass B  chunk by chunk again: expand, depthwise, x * gate (one rounding), and the project conv's accumulators advance
//           over the chunk's channels - chunk order is channel order, so every output's chain is the plan's ascending-k chain
// STATUS (round 4): correct - every materialised tensor and the network output bit-identical to the oracle - and NOT the
// default (OCR_FUSE_MB=1 turns it on).  First form measured on configs[1] (2048 lines): the eight blocks 11.6 ms against
// 2.8 ms for the 48 launches they replace.  A line's chain of (chunk x {stage weights, expand, depthwise, row sums}) phases
// is barrier-separated and latency-bound at one or two resident workgroups per CU (the 200-channel blocks get 8-channel
// chunks under a 78 KB working set: 50 chunk computations per line), the SE fully-connected layers read their weights from
// global memory inside dependent loops, and a 1024-thread form with 40-channel chunks spills (784 B of scratch per thread,
// 5.6 ms for one 200-channel block).  What it needs to pay: chunk parameters and SE weights prefetched a chunk ahead,
// packed FMAs, the division-free hard-swish, no dynamically indexed local arrays.  Its ceiling is modest: the two passes cost
// 5.6 M FMAs per line for the 200-channel blocks - ~65 k packed wave instructions, 30 us per line on a whole CU at full VALU
// rate, 0.24 ms per launch against 0.52 ms for the six launches now - and a per-line chain of barrier-separated phases will
// not run at full VALU rate.
// Same arithmetic per value as the six kernels it replaces: f32, one fma chain per contraction from 0
// in ascending k, mul-then-add BN, IEEE division in hard-swish, the pool's two sequential passes - results are bit-identical
// (tests/test_gpu_parity.py: every materialised tensor of the production launch list against the oracle).  Only weights
// travel from L2; activations in: the block's input once, out: its output once.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels_net.h"

namespace ocr {

namespace {

__device__ __forceinline__ float4 fma4(float x, const float4& w, const float4& a) {
  return make_float4(fmaf(x, w.x, a.x), fmaf(x, w.y, a.y), fmaf(x, w.z, a.z), fmaf(x, w.w, a.w));
}
__device__ __forceinline__ float4 fma4v(const float4& x, const float4& w, const float4& a) {
  return make_float4(fmaf(x.x, w.x, a.x), fmaf(x.y, w.y, a.y), fmaf(x.z, w.z, a.z), fmaf(x.w, w.w, a.w));
}
__device__ __forceinline__ float4 bn_act(const float4& v, const float4& s, const float4& t, int act) {
  float4 r;
  float u;
  u = v.x * s.x; r.x = u + t.x;
  u = v.y * s.y; r.y = u + t.y;
  u = v.z * s.z; r.z = u + t.z;
  u = v.w * s.w; r.w = u + t.w;
  if (act >= 0) { r.x = ocr_act(act, 0.f, 0.f, r.x); r.y = ocr_act(act, 0.f, 0.f, r.y); r.z = ocr_act(act, 0.f, 0.f, r.z); r.w = ocr_act(act, 0.f, 0.f, r.w); }
  return r;
}

constexpr int MB_ITEMS = 3;  // project-conv items (pixel pair x 4 output channels) a thread may own

}  // namespace

template <int K>
__global__ void __launch_bounds__(256) mbconv_se_kernel(const MbArgs a) {
  extern __shared__ float4 s_mb4[];
  float* sm = (float*)s_mb4;
  const int tid = threadIdx.x, n = blockIdx.x, NTH = blockDim.x;
  const int W = a.W, Pi = a.Hi * W, Po = a.Ho * W, CC = a.CC, CCS = CC + 4, Q = CC >> 2, Cin = a.Cin, CinS = Cin + 4;
  const int CoutP = (a.Cout + 3) & ~3, OQ = CoutP >> 2, CexpP = (a.Cexp + 7) & ~7, P2 = K / 2;
  // ---- LDS carve-up (floats)
  float* sX = sm;                       // [Pi][CinS]   block input, logical channel order
  float* sE = sX + Pi * CinS;           // [Pi][CCS]    expanded chunk
  float* sD = sE + Pi * CCS;            // [Po][CCS]    depthwise chunk (then gated)
  float* sW1 = sD + Po * CCS;           // [Cin][CC]
  float* sWd = sW1 + Cin * CC;          // [K*K][CC]
  float* sW2 = sWd + K * K * CC;        // [CC][CoutP]
  float* sBn = sW2 + CC * CoutP;        // [4][CC]      s1 | t1 | s2 | t2 of the chunk
  float* sRow = sBn + 4 * CC;           // [Ho][CexpP]  the pool's row sums
  float* sGate = sRow + a.Ho * CexpP;   // [CexpP]      pooled, then gate
  float* sHid = sGate + CexpP;          // [R]
  // ---- block input: C8I global -> logical LDS
  {
    const float* gin = a.in + (long)n * Pi * a.Cs_in;
    const int q4 = a.Cs_in >> 2;
    for (int i = tid; i < Pi * q4; i += NTH) {
      const int p = i / q4, q = i - p * q4;
      const float4 v = *(const float4*)(gin + (long)p * a.Cs_in + 4 * q);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lc = c8i_logical(4 * q + j);
        if (lc < Cin) sX[p * CinS + lc] = e[j];
      }
    }
  }
  // expand + depthwise of the chunk of expanded channels c0 .. c0 + CC - 1 -> sD (channels past Cexp: zeros)
  auto compute_chunk = [&](int c0, bool with_w2) {
    for (int i = tid; i < Cin * CC; i += NTH) {
      const int k = i / CC, c = i - k * CC;
      sW1[i] = c0 + c < a.Cexp ? a.w1[(long)k * CexpP + c0 + c] : 0.f;
    }
    for (int i = tid; i < K * K * CC; i += NTH) {
      const int t = i / CC, c = i - t * CC;
      sWd[i] = c0 + c < a.Cexp ? a.wd[(long)t * a.Cs_exp + c8i_phys(c0 + c)] : 0.f;
    }
    for (int i = tid; i < CC; i += NTH) {
      const bool ok = c0 + i < a.Cexp;
      const int pc = c8i_phys(c0 + i);
      sBn[i] = ok ? a.s1[pc] : 0.f; sBn[CC + i] = ok ? a.t1[pc] : 0.f;
      sBn[2 * CC + i] = ok ? a.s2[pc] : 0.f; sBn[3 * CC + i] = ok ? a.t2[pc] : 0.f;
    }
    if (with_w2)
      for (int i = tid; i < CC * CoutP; i += NTH) {
        const int c = i / CoutP, o = i - c * CoutP;
        sW2[i] = (c0 + c < a.Cexp && o < a.Cout) ? a.w2[(long)(c0 + c) * ((a.Cout + 7) & ~7) + o] : 0.f;
      }
    __syncthreads();
    // ---- expand: item = 2 pixels x 4 channels
    for (int it = tid; it < (Pi >> 1) * Q; it += NTH) {
      const int q = it % Q, p0 = (it / Q) * 2;
      float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
      const float* x0 = sX + p0 * CinS;
      const float* x1 = x0 + CinS;
      for (int k = 0; k < Cin; k += 4) {
        const float4 xa = *(const float4*)(x0 + k), xb = *(const float4*)(x1 + k);
        const float4 w0 = *(const float4*)(sW1 + (k + 0) * CC + 4 * q), w1 = *(const float4*)(sW1 + (k + 1) * CC + 4 * q);
        const float4 w2 = *(const float4*)(sW1 + (k + 2) * CC + 4 * q), w3 = *(const float4*)(sW1 + (k + 3) * CC + 4 * q);
        acc0 = fma4(xa.x, w0, acc0); acc1 = fma4(xb.x, w0, acc1);
        acc0 = fma4(xa.y, w1, acc0); acc1 = fma4(xb.y, w1, acc1);
        acc0 = fma4(xa.z, w2, acc0); acc1 = fma4(xb.z, w2, acc1);
        acc0 = fma4(xa.w, w3, acc0); acc1 = fma4(xb.w, w3, acc1);
      }
      const float4 s = *(const float4*)(sBn + 4 * q), t = *(const float4*)(sBn + CC + 4 * q);
      *(float4*)(sE + p0 * CCS + 4 * q) = bn_act(acc0, s, t, a.act1);
      *(float4*)(sE + (p0 + 1) * CCS + 4 * q) = bn_act(acc1, s, t, a.act1);
    }
    __syncthreads();
    // ---- depthwise: item = 2 horizontally adjacent output pixels x 4 channels; taps in (ky, kx) ascending order,
    // zero padding contributes fma(0, w, acc) as in dw_conv_kernel
    const int W2 = W >> 1;
    for (int it = tid; it < a.Ho * W2 * Q; it += NTH) {
      const int q = it % Q, r = it / Q, ox0 = (r % W2) * 2, oy = r / W2;
      float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * a.SH - P2 + ky;
        const bool rv = iy >= 0 && iy < a.Hi;
        float4 in[K + 1];
#pragma unroll
        for (int j = 0; j < K + 1; ++j) {
          const int ix = ox0 - P2 + j;
          in[j] = (rv && ix >= 0 && ix < W) ? *(const float4*)(sE + (iy * W + ix) * CCS + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 w = *(const float4*)(sWd + (ky * K + kx) * CC + 4 * q);
          acc0 = fma4v(in[kx], w, acc0);
          acc1 = fma4v(in[kx + 1], w, acc1);
        }
      }
      const float4 s = *(const float4*)(sBn + 2 * CC + 4 * q), t = *(const float4*)(sBn + 3 * CC + 4 * q);
      *(float4*)(sD + (oy * W + ox0) * CCS + 4 * q) = bn_act(acc0, s, t, a.act2);
      *(float4*)(sD + (oy * W + ox0 + 1) * CCS + 4 * q) = bn_act(acc1, s, t, a.act2);
    }
    __syncthreads();
  };

  const int nchunks = (a.Cexp + CC - 1) / CC;
  // ================================================================ pass A: the pool
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * CC;
    compute_chunk(c0, false);
    // row sums: a lane per (output row, channel), x ascending from 0 (gap_rows_kernel's order)
    if (tid < a.Ho * CC) {
      const int oy = tid / CC, c = tid - oy * CC;
      const float* src = sD + (oy * W) * CCS + c;
      float s = 0.f;
      constexpr int U = 8;
      int x = 0;
      for (; x + U <= W; x += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[(x + u) * CCS];
#pragma unroll
        for (int u = 0; u < U; ++u) s = s + v[u];
      }
      for (; x < W; ++x) s = s + src[x * CCS];
      if (c0 + c < CexpP) sRow[oy * CexpP + c0 + c] = s;
    }
    __syncthreads();
  }
  // ---- pool's column pass, SE
  const float cnt = (float)(a.Ho * W);
  for (int c = tid; c < a.Cexp; c += NTH) {
    float s = 0.f;
    for (int y = 0; y < a.Ho; ++y) s = s + sRow[y * CexpP + c];
    sGate[c] = s / cnt;
  }
  __syncthreads();
  for (int j = tid; j < a.R; j += NTH) {
    const float* w = a.se_w1 + (long)j * a.Cexp;
    float acc = 0.f;
    for (int c = 0; c < a.Cexp; ++c) acc = fmaf(sGate[c], w[c], acc);
    acc = acc + a.se_b1[j];
    sHid[j] = fmaxf(acc, 0.0f);
  }
  __syncthreads();
  float gmine[2] = {0.f, 0.f};  // (Cexp <= 512)
  for (int c = tid, i = 0; c < a.Cexp; c += NTH, ++i) {
    const float* w = a.se_w2 + (long)c * a.R;
    float acc = 0.f;
    for (int j = 0; j < a.R; ++j) acc = fmaf(sHid[j], w[j], acc);
    acc = acc + a.se_b2[c];
    float t = acc * a.slope;
    t = t + a.offset;
    gmine[i] = fminf(fmaxf(t, 0.0f), 1.0f);
  }
  __syncthreads();  // (every pooled value has been read)
  for (int c = tid, i = 0; c < CexpP; c += NTH, ++i) sGate[c] = c < a.Cexp ? gmine[i] : 0.f;
  __syncthreads();
  // ================================================================ pass B: gate, project
  float4 yacc[MB_ITEMS][2];
#pragma unroll
  for (int i = 0; i < MB_ITEMS; ++i) yacc[i][0] = yacc[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int nitems = (Po >> 1) * OQ;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * CC;
    compute_chunk(c0, true);
    for (int it = tid; it < Po * Q; it += NTH) {  // x * gate: one rounding (the ew pass / the conv's folded gate)
      const int q = it % Q, p = it / Q;
      float4 v = *(const float4*)(sD + p * CCS + 4 * q);
      const float4 g = *(const float4*)(sGate + c0 + 4 * q);
      v.x = v.x * g.x; v.y = v.y * g.y; v.z = v.z * g.z; v.w = v.w * g.w;
      *(float4*)(sD + p * CCS + 4 * q) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MB_ITEMS; ++i) {
      const int it = tid + i * NTH;
      if (it < nitems) {
        const int oq = it % OQ, p0 = (it / OQ) * 2;
        const float* d0 = sD + p0 * CCS;
        const float* d1 = d0 + CCS;
        float4 acc0 = yacc[i][0], acc1 = yacc[i][1];
        for (int c = 0; c < CC; c += 4) {
          const float4 ga = *(const float4*)(d0 + c), gb = *(const float4*)(d1 + c);
          const float4 w0 = *(const float4*)(sW2 + (c + 0) * CoutP + 4 * oq), w1 = *(const float4*)(sW2 + (c + 1) * CoutP + 4 * oq);
          const float4 w2 = *(const float4*)(sW2 + (c + 2) * CoutP + 4 * oq), w3 = *(const float4*)(sW2 + (c + 3) * CoutP + 4 * oq);
          acc0 = fma4(ga.x, w0, acc0); acc1 = fma4(gb.x, w0, acc1);
          acc0 = fma4(ga.y, w1, acc0); acc1 = fma4(gb.y, w1, acc1);
          acc0 = fma4(ga.z, w2, acc0); acc1 = fma4(gb.z, w2, acc1);
          acc0 = fma4(ga.w, w3, acc0); acc1 = fma4(gb.w, w3, acc1);
        }
        yacc[i][0] = acc0; yacc[i][1] = acc1;
      }
    }
    __syncthreads();
  }
  // ---- BN (+ residual) and the C8I store through LDS (sE is free: [Po][CoutP + 4] logical)
  float* sY = sE;
  const int YS = CoutP + 4;
#pragma unroll
  for (int i = 0; i < MB_ITEMS; ++i) {
    const int it = tid + i * NTH;
    if (it < nitems) {
      const int oq = it % OQ, p0 = (it / OQ) * 2;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float acc[4] = {yacc[i][h].x, yacc[i][h].y, yacc[i][h].z, yacc[i][h].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int o = 4 * oq + j;
          float v = 0.f;
          if (o < a.Cout) {
            const int pc = c8i_phys(o);
            const float u = acc[j] * a.s3[pc];
            v = u + a.t3[pc];
            if (a.res) v = v + a.res[((long)n * Po + p0 + h) * a.Cs_out + pc];
          }
          sY[(p0 + h) * YS + o] = v;
        }
      }
    }
  }
  __syncthreads();
  {
    float* gout = a.out + (long)n * Po * a.Cs_out;
    const int q4 = a.Cs_out >> 2;
    for (int i = tid; i < Po * q4; i += NTH) {
      const int p = i / q4, q = i - p * q4;
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lc = c8i_logical(4 * q + j);
        e[j] = lc < a.Cout ? sY[p * YS + lc] : 0.f;  // the pad channels of the octet layout stay zero
      }
      *(float4*)(gout + (long)p * a.Cs_out + 4 * q) = make_float4(e[0], e[1], e[2], e[3]);
    }
  }
}

constexpr int kMbThreads = 256;
// LDS floats of a launch with chunk width cc
static size_t mb_lds_floats(const MbArgs& a, int cc) {
  const int Pi = a.Hi * a.W, Po = a.Ho * a.W, CoutP = (a.Cout + 3) & ~3, CexpP = (a.Cexp + 7) & ~7;
  const size_t e = (size_t)Pi * (cc + 4);
  const size_t y = (size_t)Po * (CoutP + 4);  // the output staging reuses the expanded chunk's region
  return (size_t)Pi * (a.Cin + 4) + std::max(e, y) + (size_t)Po * (cc + 4) + (size_t)a.Cin * cc + (size_t)a.K * a.K * cc + (size_t)cc * CoutP +
         4 * (size_t)cc + (size_t)a.Ho * CexpP + CexpP + a.R + 64;
}

// chunk width for this block: the widest of 40 / 32 / 24 / 16 / 8 whose working set leaves room for two workgroups per CU;
// 0: the block is not on this path
int mbconv_chunk(const MbArgs& a) {
  if ((a.K != 3 && a.K != 5) || a.W % 2 || a.Cin % 4 || a.Cin > 64 || a.Cexp > 512 || a.Cout > 64 || a.R > 128 || a.Ho < 1) return 0;
  const int Po = a.Ho * a.W, CoutP = (a.Cout + 3) & ~3;
  if ((Po >> 1) * (CoutP >> 2) > MB_ITEMS * kMbThreads || a.Ho * 8 > kMbThreads) return 0;
  for (int cc : {40, 32, 24, 16, 8})
    if (a.Ho * cc <= kMbThreads && mb_lds_floats(a, cc) * sizeof(float) <= 78 * 1024) return cc;
  return 0;
}

bool launch_mbconv(const MbArgs& a0, hipStream_t s) {
  MbArgs a = a0;
  if (a.CC <= 0) a.CC = mbconv_chunk(a);
  if (a.CC <= 0) return false;
  const size_t lds = mb_lds_floats(a, a.CC) * sizeof(float);
  if (a.K == 5) {
    static LdsAttrMemo memo;
    if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)mbconv_se_kernel<5>, (int)lds, memo)) return false;
    hipLaunchKernelGGL((mbconv_se_kernel<5>), dim3(a.N), dim3(kMbThreads), lds, s, a);
  } else {
    static LdsAttrMemo memo;
    if (lds > 64 * 1024 && !raise_dynamic_lds((const void*)mbconv_se_kernel<3>, (int)lds, memo)) return false;
    hipLaunchKernelGGL((mbconv_se_kernel<3>), dim3(a.N), dim3(kMbThreads), lds, s, a);
  }
  return true;
}

}  // namespace ocr
