// srv_pgemm_kernel: the PERSISTENT form of the server networks' GEMM (included by srv_kernels.hip) for 1x1 stride-1 convs, linears
// and 2x2 transposed convs (GemmArgs::x1) - the layers whose K is 64 ... 2048 and whose time is memory time.
//
// Why a second form.  The first form (srv_gemm_kernel) runs one output tile per workgroup: at K = 64 ... 256 that is one to four
// K tiles between a cold start (two DMA stages requested, nothing to do until they land) and an epilogue during which nothing is
// in flight; measured on the res2 / res3 bottlenecks and SVTR's stage-1 linears it moves 1.4 - 2.7 TB/s of algorithmic bytes.
// Here a workgroup per CU walks a contiguous range of tiles (column tiles of one pixel tile first: it re-reads its X panel from
// L2) and the LDS-DMA ring never drains: the stages of tile t + 1 (and its RESIDUAL tile, which arrives by DMA too, into one of
// two LDS buffers) are in flight while tile t computes and leaves.  The epilogue runs in the accumulator layout on that
// residual image and writes its results IN PLACE (same lane, same address: no barrier between reading the residual and
// writing the result), one barrier, then whole 16-byte pieces of pixel rows go out.  Parameters come by scalar loads, the
// steady state has no vector load the compiler knows of: its own s_waitcnt vmcnt never drains the ring.
// DMA accounting: every wave counts the DMA instructions it has issued; each ring slot / residual buffer remembers the count
// at its issue, and a wait is "at most (issued - mark) outstanding" (in-order retirement of the VM counter).
#pragma once

template <typename T, int BM, int BN, int WM, int WN, int NS, bool OF32>
struct PGeom {
  static constexpr int NW = WM * WN, NT = 64 * NW;
  static constexpr int KG = ET<T>::KG, BK = ET<T>::BK;
  static constexpr int TN = BN / WN / 32, TM = BM / WM / 32;
  static constexpr int WI = BN / 8 / NW, XI = BM / 8 / NW, LPS = WI + XI;
  static constexpr unsigned STG = (unsigned)(BN + BM) * 128u;
  static constexpr int OS = OF32 ? 4 : (int)sizeof(T);          // bytes per output element
  static constexpr int ROWB = BN * OS, GPR = ROWB / 16;         // a pixel row of the residual / result image, its granules
  static constexpr int SWM = (GPR < 16 ? GPR : 16) - 1;         // granule g of row p sits at slot g ^ (p & SWM)
  static constexpr unsigned RBUF = (unsigned)BM * BN * (unsigned)sizeof(T);  // one residual buffer; OF32 (f16 in, f32 out): both as one
  static constexpr int RI = (int)(RBUF / 1024u) / NW;           // residual DMA instructions per wave
  static constexpr unsigned R0 = NS * STG, LDS = NS * STG + 2 * RBUF;
  static_assert(!OF32 || sizeof(T) == 2, "f32 output of f16 inputs (the CTC logits)");
  static_assert((BN / 8) % NW == 0 && (BM / 8) % NW == 0 && NW % 2 == 0, "DMA plan");
  static_assert((RBUF / 1024u) % NW == 0, "residual DMA plan");
  static_assert(BM * BN * OS <= 2 * (int)RBUF, "result image");
  static_assert(NS == 3, "the stage loop is unrolled over three ring slots");
  static_assert((BM * BN * OS / 16) % NT == 0, "every wave issues the same number of stores");
};

__device__ __forceinline__ void srv_wait_vm_le(int n) {  // wave-uniform n; the counter holds at most 63
  switch (n) {
#define W1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define W8(b) W1(b) W1(b + 1) W1(b + 2) W1(b + 3) W1(b + 4) W1(b + 5) W1(b + 6) W1(b + 7)
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    W1(1) W1(2) W1(3) W1(4) W1(5) W1(6) W1(7)
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    W1(9) W1(10) W1(11) W1(12) W1(13) W1(14) W1(15) W1(16) W1(17) W1(18) W1(19) W1(20) W1(21) W1(22) W1(23) W1(24) W1(25) W1(26) W1(27) W1(28)
    W1(29) W1(30) W1(31) W1(32) W1(33) W1(34) W1(35) W1(36) W1(37) W1(38) W1(39) W1(40) W1(41) W1(42) W1(43) W1(44) W1(45) W1(46) W1(47) W1(48)
    W1(49) W1(50) W1(51) W1(52) W1(53) W1(54) W1(55) W1(56) W1(57) W1(58) W1(59) W1(60) W1(61) W1(62)
#undef W8
#undef W1
    default: break;
  }
}

typedef float f8s __attribute__((ext_vector_type(8)));
// 32 consecutive floats of a parameter vector at a wave-uniform index (a 32-column block of the tile), by scalar loads: four
// requests, ONE wait (lgkmcnt: independent of the DMA ring's vmcnt)
struct P32 { f8s q[4]; };
__device__ __forceinline__ P32 srv_sload32(const float* p, int idx) {
  P32 v;
  const unsigned o0 = (unsigned)idx * 4u, o1 = o0 + 32u, o2 = o0 + 64u, o3 = o0 + 96u;
  asm volatile(
      "s_load_dwordx8 %0, %4, %5\n\ts_load_dwordx8 %1, %4, %6\n\ts_load_dwordx8 %2, %4, %7\n\ts_load_dwordx8 %3, %4, %8\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(v.q[0]), "=&s"(v.q[1]), "=&s"(v.q[2]), "=&s"(v.q[3])
      : "s"(p), "s"(o0), "s"(o1), "s"(o2), "s"(o3)
      : "memory");
  return v;
}
// one 16-byte store through a buffer descriptor: a lane whose offset is SRV_OOB stores nothing.  Always ONE instruction per
// call, whatever the lanes do: it is counted with the DMAs (loads, stores and LDS-DMA retire in order on one counter)
__device__ __forceinline__ void srv_store16(f4v d, unsigned voff, v4u rsrc) {
  // (s_nop 1: a store of more than 8 bytes reads its data registers for two more cycles - the compiler pads this for its own
  // stores, nothing inside an asm statement; without it the next instruction's write of d.x reached HBM in a quarter of the lanes)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(d), "v"(voff), "s"(rsrc) : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int NS, bool OF32>
__global__ void __launch_bounds__(64 * WM * WN) srv_pgemm_kernel(const GemmArgs a, int tiles_per_block) {
  using G = PGeom<T, BM, BN, WM, WN, NS, OF32>;
  constexpr int NW = G::NW, NT = G::NT, KG = G::KG, BK = G::BK, TN = G::TN, TM = G::TM, WI = G::WI, XI = G::XI, LPS = G::LPS;
  constexpr int OS = G::OS, ROWB = G::ROWB, GPR = G::GPR, SWM = G::SWM, RI = G::RI;
  constexpr unsigned STG = G::STG, RBUF = G::RBUF, R0 = G::R0;
  constexpr bool HALF = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(size_t)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int nb_n = (a.Ncols + BN - 1) / BN;
  const int ntiles = (int)((a.M + BM - 1) / BM) * nb_n;
  const int t0 = (int)blockIdx.x * tiles_per_block;
  const int t1 = min(t0 + tiles_per_block, ntiles);
  if (t0 >= t1) return;
  const int nkt = a.nkt;
  const int S = (t1 - t0) * nkt;  // stages of this workgroup's stream

  v4u rs_w, rs_x, rs_r, rs_y;
  {
    const unsigned long long wb = (unsigned long long)a.w, xb = (unsigned long long)a.x, rb = (unsigned long long)a.res, yb = (unsigned long long)a.y;
    rs_y.x = __builtin_amdgcn_readfirstlane((unsigned)yb);
    rs_y.y = __builtin_amdgcn_readfirstlane((unsigned)(yb >> 32));
    rs_y.z = __builtin_amdgcn_readfirstlane((unsigned)a.y_bytes);
    rs_y.w = 0x00020000u;
    rs_w.x = __builtin_amdgcn_readfirstlane((unsigned)wb);
    rs_w.y = __builtin_amdgcn_readfirstlane((unsigned)(wb >> 32));
    rs_w.z = __builtin_amdgcn_readfirstlane((unsigned)a.w_bytes);
    rs_w.w = 0x00020000u;
    rs_x.x = __builtin_amdgcn_readfirstlane((unsigned)xb);
    rs_x.y = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
    rs_x.z = __builtin_amdgcn_readfirstlane((unsigned)a.x_bytes);
    rs_x.w = 0x00020000u;
    rs_r.x = __builtin_amdgcn_readfirstlane((unsigned)rb);
    rs_r.y = __builtin_amdgcn_readfirstlane((unsigned)(rb >> 32));
    rs_r.z = __builtin_amdgcn_readfirstlane((unsigned)a.res_bytes);
    rs_r.w = 0x00020000u;
  }
  // ---- DMA plans (srv_gemm_kernel's: rows 8 (wave + NW i) + (lane >> 3), slot lane & 7, source granule slot ^ ((row >> 1) & 7))
  const int gq = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
  unsigned wvo[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) wvo[i] = (unsigned)((wave + NW * i) * 1024 + lane * 16);
  unsigned xvo[XI];  // the lane's X rows of the tile being ISSUED (set when the issue side enters a pixel tile)
  int issued = 0;    // DMA instructions this wave has issued
  int i_t = t0, i_kt = 0, i_mt = -1;
  auto issue_stage = [&](unsigned slot_base) __attribute__((always_inline)) {
    const int mt = i_t / nb_n, nt = i_t - mt * nb_n;
    if (mt != i_mt) {
      i_mt = mt;
#pragma unroll
      for (int j = 0; j < XI; ++j) {
        const long m = (long)mt * BM + 8 * (wave + NW * j) + (lane >> 3);
        xvo[j] = m < a.M ? (unsigned)((unsigned long long)m * (unsigned)a.Cin * sizeof(T)) + (unsigned)(gq * 16) : SRV_OOB;
      }
    }
    const unsigned wso = (unsigned)(((unsigned long long)i_kt * (unsigned)a.Npad + (unsigned)(nt * BN)) * 128ull);
    const unsigned xso = (unsigned)i_kt * (unsigned)(BK * sizeof(T));
#pragma unroll
    for (int i = 0; i < WI; ++i) srv_dma16(slot_base + (unsigned)(wave + NW * i) * 1024u, wvo[i], rs_w, wso);
#pragma unroll
    for (int j = 0; j < XI; ++j) srv_dma16(slot_base + (unsigned)BN * 128u + (unsigned)(wave + NW * j) * 1024u, xvo[j], rs_x, xso);
    issued += LPS;
    if (++i_kt == nkt) { i_kt = 0; ++i_t; }
  };
  // residual image of tile t -> buffer t & 1: piece (row p, slot q) holds granule q ^ (p & SWM) of the row
  const bool has_res = a.res_up != 0;
  auto issue_res = [&](int t) __attribute__((always_inline)) {
    const int mt = t / nb_n, nt = t - mt * nb_n;
    const unsigned rb = lds0 + R0 + (unsigned)(t & 1) * RBUF;
#pragma unroll
    for (int i = 0; i < RI; ++i) {
      const int piece = (wave + NW * i) * 64 + lane;
      const int p = piece / GPR, q = piece - p * GPR;
      const int g = q ^ (p & SWM);
      const long m = (long)mt * BM + p;
      unsigned off = SRV_OOB;
      if (m < a.M) {
        long sp = m;
        if (a.res_up == 2) {
          const int ohw = a.OH * a.OW;
          const int ni = (int)(m / ohw);
          const int rem = (int)(m - (long)ni * ohw);
          const int oy = rem / a.OW, ox = rem - oy * a.OW;
          sp = ((long)ni * (a.OH >> 1) + (oy >> 1)) * (a.OW >> 1) + (ox >> 1);
        }
        const int n = nt * BN + g * KG;
        if (n < a.Cs_out) off = (unsigned)(((unsigned long long)sp * (unsigned)a.Cs_out + (unsigned)n) * sizeof(T));
      }
      srv_dma16(rb + (unsigned)(wave + NW * i) * 1024u, off, rs_r, 0u);
    }
    issued += RI;
  };

  const int swz = (r >> 1) & 7;
  f16x acc[TN][TM];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  };
  zero_acc();

  // ---- epilogue of tile t (all waves; the accumulators hold it)
  auto epilogue = [&](int t) __attribute__((always_inline)) {
    const int mt = t / nb_n, nt = t - mt * nb_n;
    const int n0 = nt * BN;
    const long m0 = (long)mt * BM;
    // the result image: buffer t & 1 - holding the tile's residual, read and overwritten in place (same lane, same address), or
    // nothing yet.  (f16 in, f32 out: both buffers as one image; consecutive tiles are a barrier apart)
    unsigned char* const img = smem + R0 + (OF32 ? 0u : (unsigned)(t & 1) * RBUF);
    const int ps = r & SWM;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      // parameters of the 32 columns this (wave, i) covers: one vector at a time (32 SGPRs), applied to the accumulators in place
      const int pidx = n0 + wn * TN * 32 + i * 32;
      if (a.bias) {
        const P32 P = srv_sload32(a.bias, pidx);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pv = h ? P.q[q][4 + e] : P.q[q][e];
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j][4 * q + e] = acc[i][j][4 * q + e] + pv;
          }
      }
      if (a.scale) {
        {
          const P32 P = srv_sload32(a.scale, pidx);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float pv = h ? P.q[q][4 + e] : P.q[q][e];
#pragma unroll
              for (int j = 0; j < TM; ++j) acc[i][j][4 * q + e] = acc[i][j][4 * q + e] * pv;
            }
        }
        {
          const P32 P = srv_sload32(a.shift, pidx);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float pv = h ? P.q[q][4 + e] : P.q[q][e];
#pragma unroll
              for (int j = 0; j < TM; ++j) acc[i][j][4 * q + e] = acc[i][j][4 * q + e] + pv;
            }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cb = wn * TN * 32 + i * 32 + 8 * q;  // wave-uniform first channel of the eight this (i, q) covers in the tile
        const int co = cb + 4 * h;
        const int bo = co * OS, gi = bo >> 4, wi = bo & 15;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int p = wm * TM * 32 + j * 32 + r;
          unsigned char* at = img + p * ROWB + ((gi ^ ps) << 4) + wi;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
          if (has_res) {
            if constexpr (HALF) {
              const h4v rv = *(const h4v*)at;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] + (float)rv[e];
            } else {
              const f4v rv = *(const f4v*)at;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] + rv[e];
            }
          }
          if (a.act != SACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = srv_act(a.act, v[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + co + e >= a.Ncols) v[e] = 0.f;
          if constexpr (OS == 2) {
            h4v o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (_Float16)__builtin_amdgcn_fmed3f(v[e], -65504.0f, 65504.0f);
            *(h4v*)at = o4;
          } else {
            *(f4v*)at = f4v{v[0], v[1], v[2], v[3]};
          }
        }
      }
    }
    zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a raw barrier: __syncthreads() would wait for vmcnt(0) and drain the DMA ring)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int ncols_store = a.deconv ? a.Ncols : a.Cs_out;
#pragma unroll
    for (int it0 = 0; it0 < BM * GPR; it0 += NT) {  // (BM * GPR is a multiple of the workgroup: every wave issues every store)
      const int it = it0 + tid;
      const int p = it / GPR, q = it - p * GPR;
      const int g = q ^ (p & SWM);
      const long m = m0 + p;
      const int n = n0 + g * (16 / OS);
      const f4v d = *(const f4v*)(img + p * ROWB + (q << 4));
      unsigned off = SRV_OOB;
      if (m < a.M && n < ncols_store) {
        long opix = m;
        int co = n;
        if (a.deconv) {
          const int dq = n / a.CoutD;
          co = n - dq * a.CoutD;
          const int ohw = a.OH * a.OW;
          const int ni = (int)(m / ohw);
          const int rem = (int)(m - (long)ni * ohw);
          const int oy = rem / a.OW, ox = rem - oy * a.OW;
          opix = ((long)ni * (2 * a.OH) + 2 * oy + (dq >> 1)) * (2 * a.OW) + 2 * ox + (dq & 1);
        }
        off = (unsigned)(((unsigned long long)opix * (unsigned)a.Cs_out + (unsigned)co) * OS);
      }
      srv_store16(d, off, rs_y);
    }
    issued += (BM * GPR) / NT;
  };

  // ---- the stream: stage s lives in ring slot s % 3; marks = `issued` right after the slot's / buffer's DMAs went out
  int mark[3] = {0, 0, 0}, rmark[2] = {0, 0};
  int c_t = t0, c_kt = 0;
  if (has_res) { issue_res(t0); rmark[t0 & 1] = issued; }
  int is_s = 0;  // stages issued
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (is_s < S) { issue_stage(lds0 + (unsigned)s * STG); mark[s] = issued; ++is_s; }
  for (int base = 0; base < S; base += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      if (base + u >= S) break;
      srv_wait_vm_le(__builtin_amdgcn_readfirstlane(issued - mark[u]));
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (is_s < S) {  // stage base + u + 2 into the slot stage base + u - 1 left
        const int nu = (u + 2) % 3;
        issue_stage(lds0 + (unsigned)nu * STG);
        mark[nu] = issued;
        ++is_s;
      }
      if (has_res && c_kt == 0 && c_t + 1 < t1) {  // the next tile's residual: its buffer was read out by the tile before this one
        issue_res(c_t + 1);
        rmark[(c_t + 1) & 1] = issued;
      }
      const unsigned char* sw = smem + (unsigned)u * STG + (unsigned)(wn * TN * 32 + r) * 128u;
      const unsigned char* sx = smem + (unsigned)u * STG + (unsigned)BN * 128u + (unsigned)(wm * TM * 32 + r) * 128u;
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
      if (++c_kt == nkt) {
        if (has_res) {  // the tile's residual image has landed (and is visible: every wave waits for its own pieces, then the barrier)
          srv_wait_vm_le(__builtin_amdgcn_readfirstlane(issued - rmark[c_t & 1]));
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        epilogue(c_t);
        c_kt = 0;
        ++c_t;
      }
    }
  }
}
