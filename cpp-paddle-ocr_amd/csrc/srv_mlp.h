// srv_mlp_kernel: SVTR's MLP - y = x + fc2(gelu(fc1(x) + b1)) + b2 - as ONE kernel (included by srv_kernels.hip; f16 build only).
//
// Why.  Run as two launches of the GEMM family, fc1 writes its 4 C wide hidden tensor to HBM and fc2 reads it back: 1.5 GB each way
// per launch at SVTR-L's first stage, 3 of the 3.4 GB the pair moves - and both launches sit in the latency-bound regime of
// DESIGN.md section 10.  Here a workgroup owns 128 tokens and walks the hidden units in chunks of 128: the chunk's fc1 product
// (128 hidden x 128 tokens, K = C) goes bias -> GELU -> f16 into an LDS image laid out exactly like a pixel tile, which IS the B
// operand of the chunk's fc2 product (C outputs x 128 tokens, K = 128) accumulated in registers over all chunks.  The hidden
// tensor never exists in HBM; the stage stream (fc1 stages: a W1 tile + an X tile; fc2 stages: a W2 tile) runs through one
// LDS-DMA ring that never drains (60 ... 190 stages per workgroup), mostly out of L2 (every workgroup streams the same weights).
// Arithmetic = the two launches': the hidden values are rounded to f16 once, fc2 accumulates them in ascending k - the fused
// output equals the unfused f16 output bit for bit (tests/test_gpu_round6.py).
// Waves: 8, wave = (wn = wave >> 2, wm = wave & 3): tokens 32 wm .. + 31; fc1: hidden 64 wn .. + 63 of the chunk (two 32-row
// blocks); fc2: outputs (C / 2) wn .. (three / four / eight blocks).  C = 512: an fc2 K tile is 64 KB - two ring slots' worth - so
// it travels as two half-tiles of 256 output rows, each consumed by the four waves of one wn (one wave per SIMD stays busy).
#pragma once

struct MlpArgs {
  const void* x;        // [M][C] f16: input AND residual
  unsigned long long x_bytes;
  const void* w1;       // weight image of fc1: rows = 4 C hidden units, K = C
  unsigned long long w1_bytes;
  int w1_npad;
  const void* w2;       // weight image of fc2: rows = C outputs, K = 4 C
  unsigned long long w2_bytes;
  int w2_npad;
  const float* b1;      // [4 C]
  const float* b2;      // [C]
  void* y;              // [M][C] f16
  long M;
  // LayerNorm of the INPUT absorbed (SVTR's post-norm: x = LN(u) feeds this MLP and nothing else): x = the raw sum u, w1 = the image of
  // diag(gamma) W1, and with a token's mean m and rstd r (found here, from the X tiles of the first hidden chunk as they pass through LDS)
  //   LN(u) W1 + b1 = r (u W1') - r m s + c,   s = column sums of W1' (as rounded to f16), c = beta W1 + b1     (ln_s, ln_c: [4 C])
  // the residual is LN(u) = (u - m) r gamma + beta on the fly.  ln_g == nullptr: x is the MLP's input as it is, b1 its bias.
  const float* ln_g = nullptr;
  const float* ln_b = nullptr;
  const float* ln_s = nullptr;
  const float* ln_c = nullptr;
  float ln_eps = 0.f;
  unsigned long long* clocks = nullptr;  // development probe (-DSRV_MLP_CLOCKS): [blocks][8 waves][8] shader cycles per phase
};

template <int C>
struct MlpGeom {
  static constexpr int BM = 128, HC = 128, NW = 8, NT = 512, NS = 3;
  static constexpr int H = 4 * C, NCH = H / HC;        // hidden chunks
  static constexpr int NKT1 = C / 64;                  // fc1 K tiles per chunk
  static constexpr int SPLIT = C > 256 ? 2 : 1;        // fc2 K tiles travel as SPLIT row pieces
  static constexpr int ROWS2 = C / SPLIT;              // output rows per fc2 stage
  static constexpr int SPC = NKT1 + 2 * SPLIT;         // stages per chunk
  static constexpr int TN2 = C / 64;                   // 32-row output blocks per wave
  static constexpr unsigned SLOT = 32768u, HBUF = NS * SLOT, BIAS = HBUF + 32768u, LDS = BIAS + 2048u;  // BIAS: the chunk's 128 fc1 biases, two slots (+ 1 KB: the column sums s of the absorbed LayerNorm)
  static constexpr int W1I = HC / 8 / NW, XI = BM / 8 / NW, W2I = (ROWS2 / 8 + NW - 1) / NW;  // DMA instructions per wave per stage
  static_assert(C % 64 == 0 && ROWS2 * 128 <= (int)SLOT && (ROWS2 / 8) % NW == 0, "stage fits a ring slot, every wave issues equally");
};

template <int C, bool LN>  // LN: the LayerNorm of the input absorbed (MlpArgs::ln_*) - its own instantiation: the plain form's loops carry no test of it
__global__ void __launch_bounds__(512) srv_mlp_kernel(const MlpArgs a) {
  using G = MlpGeom<C>;
  constexpr int BM = G::BM, NW = G::NW, NKT1 = G::NKT1, SPLIT = G::SPLIT, ROWS2 = G::ROWS2, SPC = G::SPC, TN2 = G::TN2, NCH = G::NCH;
  constexpr int W1I = G::W1I, XI = G::XI, W2I = G::W2I;
  constexpr unsigned SLOT = G::SLOT, HBUF = G::HBUF, BIAS = G::BIAS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave >> 2, wm = wave & 3;
  const long m0 = (long)blockIdx.x * BM;

  v4u rs_x, rs_w1, rs_w2, rs_b1, rs_s;
  {
    const unsigned long long sb = (unsigned long long)a.ln_s;
    rs_s.x = __builtin_amdgcn_readfirstlane((unsigned)sb);
    rs_s.y = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
    rs_s.z = (unsigned)(4 * C * 4);
    rs_s.w = 0x00020000u;
    const unsigned long long b1b = (unsigned long long)(a.ln_g ? a.ln_c : a.b1);  // (absorbed LayerNorm: c carries b1)
    rs_b1.x = __builtin_amdgcn_readfirstlane((unsigned)b1b);
    rs_b1.y = __builtin_amdgcn_readfirstlane((unsigned)(b1b >> 32));
    rs_b1.z = (unsigned)(4 * C * 4);
    rs_b1.w = 0x00020000u;
    const unsigned long long xb = (unsigned long long)a.x, w1b = (unsigned long long)a.w1, w2b = (unsigned long long)a.w2;
    rs_x.x = __builtin_amdgcn_readfirstlane((unsigned)xb);
    rs_x.y = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    rs_x.z = __builtin_amdgcn_readfirstlane((unsigned)a.x_bytes);
    rs_x.w = 0x00020000u;
    rs_w1.x = __builtin_amdgcn_readfirstlane((unsigned)w1b);
    rs_w1.y = __builtin_amdgcn_readfirstlane((unsigned)(w1b >> 32));
    rs_w1.z = __builtin_amdgcn_readfirstlane((unsigned)a.w1_bytes);
    rs_w1.w = 0x00020000u;
    rs_w2.x = __builtin_amdgcn_readfirstlane((unsigned)w2b);
    rs_w2.y = __builtin_amdgcn_readfirstlane((unsigned)(w2b >> 32));
    rs_w2.z = __builtin_amdgcn_readfirstlane((unsigned)a.w2_bytes);
    rs_w2.w = 0x00020000u;
  }
  // ---- DMA plans (srv_gemm_kernel's: instruction i of a wave = rows 8 (wave + NW i) + (lane >> 3), slot lane & 7)
  const int gq = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  unsigned xvo[XI];
#pragma unroll
  for (int j = 0; j < XI; ++j) {
    const long m = m0 + 8 * (wave + NW * j) + (lane >> 3);
    xvo[j] = m < a.M ? (unsigned)((unsigned long long)m * (unsigned)C * 2ull) + (unsigned)(gq * 16) : SRV_OOB;
  }
  const unsigned lvo = (unsigned)(lane * 16);  // weight tiles: the image is stored swizzled, a straight copy
  int issued = 0;
  int i_c = 0, i_s = 0;  // chunk and stage-in-chunk of the next stage to issue
  auto issue_stage = [&](unsigned slot_base) __attribute__((always_inline)) {
    if (i_s == 0 && wave == 0) {  // the chunk's 128 fc1 biases (512 bytes) travel with its first stage: by scalar loads in the
      // epilogue they cost two exposed L2 round trips per chunk (stamps: 5200 of a chunk's 9900 cycles in the GELU epilogue)
      srv_dma16(lds0 + BIAS + (unsigned)(i_c & 1) * 512u, lane < 32 ? (unsigned)(i_c * 512 + lane * 16) : SRV_OOB, rs_b1, 0u);
      issued += 1;
      if (LN) {
        srv_dma16(lds0 + BIAS + 1024u + (unsigned)(i_c & 1) * 512u, lane < 32 ? (unsigned)(i_c * 512 + lane * 16) : SRV_OOB, rs_s, 0u);
        issued += 1;
      }
    }
    if (i_s < NKT1) {  // fc1: W1 tile (rows = the chunk's hidden units, K tile i_s) | X tile (K tile i_s)
      const unsigned wso = (unsigned)(((unsigned long long)i_s * (unsigned)a.w1_npad + (unsigned)(i_c * G::HC)) * 128ull);
#pragma unroll
      for (int i = 0; i < W1I; ++i) srv_dma16(slot_base + (unsigned)(wave + NW * i) * 1024u, lvo + (unsigned)(wave + NW * i) * 1024u, rs_w1, wso);
      const unsigned xso = (unsigned)i_s * 128u;
#pragma unroll
      for (int j = 0; j < XI; ++j) srv_dma16(slot_base + 16384u + (unsigned)(wave + NW * j) * 1024u, xvo[j], rs_x, xso);
      issued += W1I + XI;
    } else {  // fc2: W2 piece (K tile 2 i_c + j of the hidden axis, output rows ROWS2 * piece ..)
      const int q = i_s - NKT1, j = q / SPLIT, piece = q - j * SPLIT;
      const unsigned wso = (unsigned)(((unsigned long long)(2 * i_c + j) * (unsigned)a.w2_npad + (unsigned)(piece * ROWS2)) * 128ull);
#pragma unroll
      for (int i = 0; i < W2I; ++i) srv_dma16(slot_base + (unsigned)(wave + NW * i) * 1024u, lvo + (unsigned)(wave + NW * i) * 1024u, rs_w2, wso);
      issued += W2I;
    }
    if (++i_s == SPC) { i_s = 0; ++i_c; }
  };

  const int swz = (r >> 1) & 7;
  float ln_s1 = 0.f, ln_s2 = 0.f, ln_mean = 0.f, ln_rstd = 1.f;
  f16x acc1[2], acc2[TN2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[i][q] = 0.f;
#pragma unroll
  for (int i = 0; i < TN2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc2[i][q] = 0.f;

#ifdef SRV_MLP_CLOCKS
  unsigned long long clk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long kstart = __builtin_readcyclecounter();
#endif
  constexpr int S = NCH * SPC;
  int mark[3] = {0, 0, 0};
  int is_s = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s) { issue_stage(lds0 + (unsigned)s * SLOT); mark[s] = issued; ++is_s; }
  int c_c = 0, c_s = 0;
  for (int base = 0; base < S; base += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (base + u >= S) break;
#ifdef SRV_MLP_CLOCKS
      const unsigned long long k0 = __builtin_readcyclecounter();
#endif
      srv_wait_vm_le(__builtin_amdgcn_readfirstlane(issued - mark[u]));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this wave's writes of the hidden image are in LDS before the others read it)
#ifdef SRV_MLP_CLOCKS
      const unsigned long long k1 = __builtin_readcyclecounter();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef SRV_MLP_CLOCKS
      const unsigned long long k2 = __builtin_readcyclecounter();
#endif
#ifdef SRV_MLP_CLOCKS
      const unsigned long long k3 = k2;
      clk[0] += k1 - k0; clk[1] += k2 - k1;
#endif
      const unsigned char* slot = smem + (unsigned)u * SLOT;
      if (c_s < NKT1) {
        // ---- fc1 stage: acc1[i] += W1 rows (64 wn + 32 i + r) x X rows (32 wm + r)
        const unsigned char* sw = slot + (unsigned)(wn * 64 + r) * 128u;
        const unsigned char* sx = slot + 16384u + (unsigned)(wm * 32 + r) * 128u;
        // (fragments of k step s + 1 are requested before the matrix instructions of step s: with the reads inside the step a wave
        // paid an LDS round trip per step - stamps: 2300 cycles per stage for 8 matrix instructions)
        h8v fb[2], fa[2][2];
        auto ld1 = [&](int s, int b) __attribute__((always_inline)) {
          const unsigned ko = (unsigned)(((2 * s + h) ^ swz) * 16);
          fb[b] = *(const h8v*)(sx + ko);
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[b][i] = *(const h8v*)(sw + i * 4096 + ko);
        };
        ld1(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (s < 3) ld1(s + 1, (s + 1) & 1);
#pragma unroll
          for (int i = 0; i < 2; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][i], fb[s & 1], acc1[i], 0, 0, 0);
        }
        if constexpr (LN) {
          if (c_c == 0) {  // the token's row sums, from the X tile of each of the first chunk's K tiles (this half-wave's channels) - read
                           // again behind the matrix instructions: four more LDS reads per stage of the FIRST chunk, nothing in the loop above
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const h8v xf = *(const h8v*)(sx + (unsigned)(((2 * s + h) ^ swz) * 16));
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float f = (float)xf[e];
                ln_s1 += f;
                ln_s2 = fmaf(f, f, ln_s2);
              }
            }
          }
        }
#ifdef SRV_MLP_CLOCKS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long ka = __builtin_readcyclecounter();
        clk[5] += ka - k3;
#endif
        if (c_s == NKT1 - 1) {
          // ---- the chunk's hidden values: bias, GELU, one rounding to f16, into the hidden image [k tile wn][token][granule]
          // (register 4 q + e of lane (r, h), block i = hidden 64 wn + 32 i + 8 q + 4 h + e of the chunk, token 32 wm + r)
          const int t = wm * 32 + r;
          unsigned char* hrow = smem + HBUF + (unsigned)wn * 16384u + (unsigned)t * 128u;
          const int tsw = (t >> 1) & 7;
          const float* bl = (const float*)(smem + BIAS + (unsigned)(c_c & 1) * 512u) + wn * 64 + 4 * h;
          if (LN && c_c == 0) {  // (LN: compile time) every K tile of the token's row has passed: mean and rstd (the other half-wave holds the other channels)
            float a1 = ln_s1, b1_ = ln_s1, a2 = ln_s2, b2_ = ln_s2;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1" : "+v"(a1), "+v"(b1_), "+v"(a2), "+v"(b2_));
            ln_mean = (a1 + b1_) * (1.0f / (float)C);
            ln_rstd = 1.0f / sqrtf(fmaxf((a2 + b2_) * (1.0f / (float)C) - ln_mean * ln_mean, 0.f) + a.ln_eps);
          }
          const float ln_nmr = -ln_rstd * ln_mean;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
              const f4v b4 = *(const f4v*)(bl + i * 32 + 8 * q), b5 = *(const f4v*)(bl + i * 32 + 8 * q + 8);
              float v[8];
              if (LN) {  // r (u W1') - r m s + c
                const f4v s4 = *(const f4v*)(bl + 256 + i * 32 + 8 * q), s5 = *(const f4v*)(bl + 256 + i * 32 + 8 * q + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v[e] = fmaf(acc1[i][4 * q + e], ln_rstd, fmaf(ln_nmr, s4[e], b4[e]));
                  v[4 + e] = fmaf(acc1[i][4 * q + 4 + e], ln_rstd, fmaf(ln_nmr, s5[e], b5[e]));
                }
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  v[e] = acc1[i][4 * q + e] + b4[e];
                  v[4 + e] = acc1[i][4 * q + 4 + e] + b5[e];
                }
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc1[i][4 * q + e] = 0.f;
                acc1[i][4 * q + 4 + e] = 0.f;
              }
              srv_gelu8(v);
              h4v o4, o5;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                o4[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
                o5[e] = (_Float16)__builtin_amdgcn_fmed3f(v[4 + e], -65504.0f, 65504.0f);
              }
              *(h4v*)(hrow + (((4 * i + q) ^ tsw) << 4) + 8 * h) = o4;
              *(h4v*)(hrow + (((4 * i + q + 1) ^ tsw) << 4) + 8 * h) = o5;
            }
          }
#ifdef SRV_MLP_CLOCKS
          clk[6] += __builtin_readcyclecounter() - ka;
#endif
        }
      } else {
        // ---- fc2 stage: acc2[i] += W2 rows x hidden image rows (32 wm + r), K tile j of the chunk
        const int q = c_s - NKT1, j = q / SPLIT, piece = q - j * SPLIT;
        if (SPLIT == 1 || piece == wn) {
          const unsigned char* sw = slot + (unsigned)((SPLIT == 1 ? wn * (C / 2) : 0) + r) * 128u;
          const unsigned char* sx = smem + HBUF + (unsigned)j * 16384u + (unsigned)(wm * 32 + r) * 128u;
          h8v fb[2], fa[2][TN2];
          auto ld2 = [&](int s, int b) __attribute__((always_inline)) {
            const unsigned ko = (unsigned)(((2 * s + h) ^ swz) * 16);
            fb[b] = *(const h8v*)(sx + ko);
#pragma unroll
            for (int i = 0; i < TN2; ++i) fa[b][i] = *(const h8v*)(sw + i * 4096 + ko);
          };
          ld2(0, 0);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s < 3) ld2(s + 1, (s + 1) & 1);
#pragma unroll
            for (int i = 0; i < TN2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][i], fb[s & 1], acc2[i], 0, 0, 0);
          }
        }
      }
#ifdef SRV_MLP_CLOCKS
      const unsigned long long k4 = __builtin_readcyclecounter();
      clk[3] += k4 - k3;
#endif
      // the stage after next, into the slot the previous stage left (every wave is past this stage's barrier): issued BEHIND the
      // stage's matrix instructions - a DMA piece costs its wave ~150 issue cycles, here they pass while the matrix pipe works
      if (is_s < S) {
        const int nu = (u + 2) % 3;
        issue_stage(lds0 + (unsigned)nu * SLOT);
        mark[nu] = issued;
        ++is_s;
      }
#ifdef SRV_MLP_CLOCKS
      clk[2] += __builtin_readcyclecounter() - k4;
#endif
      if (++c_s == SPC) { c_s = 0; ++c_c; }
    }
  }
#ifdef SRV_MLP_CLOCKS
  if (a.clocks && lane == 0) {
    clk[4] = __builtin_readcyclecounter() - kstart;
    for (int i = 0; i < 8; ++i) a.clocks[((size_t)blockIdx.x * 8 + wave) * 8 + i] = clk[i];
  }
#endif
  // ---- y = acc2 + b2 + x.  srv_gemm_kernel's epilogue: eight swaps per 32 x 32 block hand a lane 16-byte chunks of its token's row;
  // memory sees WHOLE LINES - the residual x comes in and y goes out as lane -> (row, chunk) with a row's chunks on neighbouring
  // lanes, through 8 KB of LDS per wave (the ring's memory, one barrier).  In 32-byte pieces straight from the accumulator layout
  // (the first form) the PMC passes counted 1.55 GB fetched and 0.51 GB written per launch at C = 256 against 0.50 GB of tensors.
  constexpr int CPR = 4 * TN2;                       // 16-byte chunks of a token's half row (C / 2 channels)
  constexpr bool POW2 = (CPR & (CPR - 1)) == 0;      // (C = 192: 12 chunks per half row - lanes walk the block's 384 chunks linearly, cells unswizzled)
  constexpr bool LINES = (32 * CPR) % 64 == 0;
  if constexpr (LINES) {
    constexpr int NI = 32 * CPR / 64;                // instructions per 32-token block: lane l of instruction t = chunk 64 t + l of the block, row-major
    constexpr int RPB = CPR >= 16 ? 1 : 16 / CPR;    // rows per 256 bytes of LDS: the XOR term changes every RPB rows
    __syncthreads();                                 // (every wave is through with the ring and the hidden image)
    unsigned char* const scr = smem + (unsigned)wave * (unsigned)(32 * CPR * 16);
    auto cell_of = [&](int row, int k) {
      return scr + (unsigned)row * (unsigned)(CPR * 16) + (unsigned)((POW2 ? (k ^ ((row / RPB) & (CPR - 1))) : k) * 16);
    };
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int row = (64 * t + lane) / CPR, kk = (64 * t + lane) % CPR;
      const long ml = m0 + wm * 32 + row;
      *(h8v*)cell_of(row, kk) = ml < a.M ? *(const h8v*)((const _Float16*)a.x + ml * C + wn * (C / 2) + 8 * kk) : h8v{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < TN2; ++i) {
      {
        f16x& A = acc2[i];
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
        const int n = wn * (C / 2) + i * 32 + 8 * h + 16 * c;
        unsigned char* const cell = cell_of(r, 4 * i + 2 * c + h);
        float bb[8], v[8];
        ld8(a.b2 + n, bb);
        const h8v rv = *(const h8v*)cell;
        if (LN) {  // the residual is LN(u), normalised here
          float gg[8], be[8];
          ld8(a.ln_g + n, gg);
          ld8(a.ln_b + n, be);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = acc2[i][8 * c + e] + bb[e];
            v[e] = t + (((float)rv[e] - ln_mean) * ln_rstd * gg[e] + be[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float t = acc2[i][8 * c + e] + bb[e];
            v[e] = t + (float)rv[e];
          }
        }
        h8v hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
        *(h8v*)cell = hv;
      }
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int row = (64 * t + lane) / CPR, kk = (64 * t + lane) % CPR;
      const long ml = m0 + wm * 32 + row;
      const h8v hv = *(const h8v*)cell_of(row, kk);
      if (ml < a.M) *(h8v*)((_Float16*)a.y + ml * C + wn * (C / 2) + 8 * kk) = hv;
    }
    return;
  }
  // ---- the piecewise form (C = 192): 32-byte pieces straight from the registers
  const long m = m0 + wm * 32 + r;
#pragma unroll
  for (int i = 0; i < TN2; ++i) {
    {
      f16x& A = acc2[i];
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
    if (m >= a.M) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = wn * (C / 2) + i * 32 + 8 * h + 16 * c;
      float bb[8], rv[8], v[8];
      ld8(a.b2 + n, bb);
      ld8((const _Float16*)a.x + m * C + n, rv);
      if (LN) {
        float gg[8], be[8];
        ld8(a.ln_g + n, gg);
        ld8(a.ln_b + n, be);
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = (rv[e] - ln_mean) * ln_rstd * gg[e] + be[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = acc2[i][8 * c + e] + bb[e];
        v[e] = t + rv[e];
      }
      st8((_Float16*)a.y + m * C + n, v);
    }
  }
}
