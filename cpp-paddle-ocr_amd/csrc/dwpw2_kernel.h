// Fused depthwise -> pointwise block, second form (round 5): the haloed input region and the depthwise weights travel
// global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... offen lds`), not through registers.  Included by kernels_dwpw.hip
// (f32 build only), inside namespace ocr, after that file's helpers (static_for, F4, pk_*, UnitPos, dw_hsw6).
//
// What changes against dwpw_kernel (same tiles, same waves, same arithmetic in the same order - bit-identical):
//   * no register sets for the items in flight (GD x (pieces + weights): 32 VGPRs in the 5x5 block), no `S` phase
//     (ds_write_b128 costs ~13 LDS cycles plus the VGPR -> LDS transfer, MI355X_MICROARCH.md LDS table) and no per-item address
//     arithmetic: a piece outside the image is a lane whose buffer offset lies beyond num_records (the hardware returns
//     zeros), the chunk offset is the instruction's scalar offset, so an item's loads are DPW instructions per wave with
//     operands that change once per pixel tile;
//   * THREE region buffers where two workgroups per CU still fit (two otherwise): the DMA of item k+2 is issued in iteration
//     k and waited for (a counted vmcnt) before the barrier of iteration k+1 - two iterations of latency budget with nothing
//     held in registers;
//   * the LDS image of a region is what a wave's DMA instruction writes: 64 consecutive 16-byte slots.  Rows are
//     [IWP pixels][Q quads], IWP * Q a multiple of 16 slots, and the quad of a pixel sits at slot (q ^ swz(column)): the 16
//     lanes of a ds_read_b128 lane group ({0-3,12-15,20-27}, ... - two tile rows, sixteen different columns) then hit
//     sixteen different 4-bank slots for every tap.  (The padded rows of the first form assumed contiguous 16-lane groups
//     and were 2-way conflicted on half their banks.)  The swizzle is applied on the SOURCE side: lane l of instruction j
//     loads the piece that belongs in slot 64 j + l.  For SW = 2 even and odd columns are stored in two halves of the row so
//     that a tap's sixteen pixels are neighbours again;
//   * the depthwise weights and the folded bias of a chunk come from one image per layer ([chunk][K*K taps | bias][CK],
//     net.hip "dwq16:" / "dwq32:") behind the region: NW more DMA instructions, dealt to the waves with the fewest region pieces;
//   * the item loop is NOT unrolled over the buffers (their offsets are scalars added to K + 2 per-lane addresses per
//     item): the steady state is one copy of the step sequence, half the code of the first form.
// Everything else - the step sequence (taps of item k inside the matrix instructions of item k-1, fragment refills after
// the last use), the operand layout, the 1x1 conv's epilogue - is dwpw_kernel's.
#pragma once

typedef unsigned ocr_v4u __attribute__((ext_vector_type(4)));

// one LDS-DMA instruction: lane l's 16 bytes at (rsrc base + soff + voff) -> LDS byte address lds + 16 l; a lane whose voff
// is not below the descriptor's num_records delivers zeros.  M0 is written in the same statement.  (Advisor, round 5: name M0 as a clobber.  hipcc refuses that
// - "inline asm clobber list contains reserved registers: m0 ... may lead to undefined behaviour": M0 is a RESERVED register
// to this compiler, it never keeps a value of its own live in it across statements and writes it itself immediately in front
// of each instruction of its own that reads it, so there is nothing for a clobber to protect.)  Invisible to the compiler's s_waitcnt bookkeeping: counted by hand (dwpw2_kernel).
__device__ __forceinline__ void ocr_dma16(unsigned lds, unsigned voff, ocr_v4u rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// -DOCR_DWPW2_VMCNT0 (build.py --variant vmcnt0 -> lib/libocr_hip_vmcnt0.so): every counted wait becomes a full drain.  If the
// hand-kept counts were ever too LARGE the two builds would differ in their bits; tests/test_gpu_parity.py runs both against the oracle.
template <int N>
__device__ __forceinline__ void ocr_wait_vm() {
#ifdef OCR_DWPW2_VMCNT0
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
#define OCR_DMA_OOB 0xfffffff0u  // beyond every num_records below

template <int K, int SH, int SW, int CK, bool WIDE, int NB_>
struct DwPw2Geom {
  static constexpr int WP = WIDE ? 2 : 4, WC = WIDE ? 2 : 1;
  static constexpr int TW = 16, TH = 2 * WP, P = TH * TW;
  static constexpr int PR = (WIDE && CK == 16) ? 1 : 2;
  static constexpr int IH = (TH - 1) * SH + K, IW = (TW - 1) * SW + K;
  static constexpr int IWP = SW == 2 ? ((IW + 7) & ~7) : ((IW + 3) & ~3);  // row pitch in pixels
  static constexpr int Q = CK / 4, C8S = CK / 8, S = CK + 4, OP_TILE = P * S;
  static constexpr int ROWB = IWP * Q * 16;                                // bytes per region row
  static constexpr int NREG = (IH * IWP * Q + 63) / 64;                    // DMA instructions of a region
  static constexpr int NWP = (K * K + 1) * Q, NW = (NWP + 63) / 64;        // pieces / instructions of the weights + bias
  static constexpr int DPW = (NREG + NW + 3) / 4, NI = 4 * DPW;            // per wave, per workgroup (the rest: idle slots)
  static constexpr int BUF = NI * 1024, W_OFF = NREG * 1024;               // bytes per buffer; the weights' offset in it
  static constexpr int NB = NB_;  // region buffers: 3 = the DMA runs two items ahead, 2 = one (where three do not leave room for two workgroups per CU)
  static constexpr int NIT = (TH / PR) * TW * Q;
  // xp: the waves' store tiles (conv_device.h, OCR_XP_FLOATS each) behind the parameters
  static constexpr size_t lds_bytes(int nttot, bool xp) { return (size_t)NB * BUF + 2 * (size_t)OP_TILE * 4 + (size_t)nttot * 32 * 4 + (xp ? (size_t)4 * OCR_XP_FLOATS * 4 : 0); }
  // stored column of region column c, and the quad swizzle of a stored column
  static __host__ __device__ constexpr int colp(int c) { return SW == 2 ? (c & 1) * (IWP / 2) + (c >> 1) : c; }
  static __host__ __device__ constexpr int swz(int cp) { return Q == 4 ? (cp >> 2) & 3 : (cp >> 1) & 7; }
};

#ifdef OCR_DWPW_CLKRATE
__device__ unsigned long long ocr_dwpw_clkrate[2];
#endif
template <int K, int SH, int SW, int CK, bool WIDE, int NT, int NBUF, int TD, int LB, bool RAG, bool XP>
__global__ void __launch_bounds__(256, LB) dwpw2_kernel(const DwPwArgs a) {
  using G_ = DwPw2Geom<K, SH, SW, CK, WIDE, NBUF>;
  constexpr int WC = G_::WC, TW = G_::TW, TH = G_::TH, PR = G_::PR, IWP = G_::IWP, IH = G_::IH, IW = G_::IW;
  constexpr int S = G_::S, Q = G_::Q, C8S = G_::C8S, OP_TILE = G_::OP_TILE, ROWB = G_::ROWB;
  constexpr int NREG = G_::NREG, NWP = G_::NWP, NW = G_::NW, DPW = G_::DPW, BUF = G_::BUF, W_OFF = G_::W_OFF, NB = G_::NB;
  static_assert(Q == 4 || Q == 8, "16- or 32-channel chunks");
  static_assert(C8S % 2 == 0, "the fragment stream is walked two octets at a time");
  static_assert(G_::NIT == 256, "one depthwise item per thread and chunk");
  static_assert((IWP * Q) % 16 == 0 && (SW == 1 || (IWP / 2) % 4 == 0), "a region row must be whole bank rows");
  static_assert(NB == 2 || NB == 3, "two or three region buffers");
  static_assert((IH - 1) * ROWB < 65536, "a region's row offsets are ds_read immediates");
  extern __shared__ float4 s_dwpw4[];
  char* const s_reg = (char*)s_dwpw4;                   // [NB][BUF]  region | weights + bias | idle slots
  float* const s_op = (float*)(s_reg + NB * BUF);       // [2][P][S]  depthwise result = MFMA pixel operand
  float* const s_par = s_op + 2 * OP_TILE;              // [NTtot*32] the 1x1 conv's bias, every column
  float* const s_xp = s_par + a.c.NTtot * 32;                // XP: [4 waves][OCR_XP_FLOATS] store tiles
  const unsigned lds0 = (unsigned)(size_t)s_reg;        // LDS byte address of the buffers

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
#ifdef OCR_DWPW_CLKRATE  // development probe: shader cycles and 100 MHz ticks over the workgroup's life, summed over workgroups
  const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
#endif

  // ---- per-thread plans (tile-independent)
  // depthwise item of this thread: output pixel (rows rp*PR .. +PR-1, column txx) x quad q
  const int txx = tid & (TW - 1), r_ = tid / TW;
  const int rp = r_ % (TH / PR), q = r_ / (TH / PR);
  unsigned d_off[K];  // byte offset in a region buffer of (row rp*PR*SH, column txx*SW + kx), this thread's quad
#pragma unroll
  for (int kx = 0; kx < K; ++kx) {
    const int cp = G_::colp(txx * SW + kx);
    d_off[kx] = (unsigned)(rp * PR * SH) * ROWB + (unsigned)((cp * Q + (q ^ G_::swz(cp))) * 16);
  }
  const int d_op = (rp * PR * TW + txx) * S + 4 * q;
  // DMA slots of this wave: instruction j = wave + 4 i writes slots 64 j .. 64 j + 63 of a buffer
  int g_pos[DPW];        // region piece: (row << 16 | column << 8 | first channel); -1: none (idle slot, or a weights instruction)
  unsigned voff[DPW];    // the lane's buffer offset (region pieces: set per unit)
  bool isw[DPW];         // wave-uniform: instruction i of this wave loads weights
#pragma unroll
  for (int i = 0; i < DPW; ++i) {
    const int j = wave + 4 * i, slot = j * 64 + lane;
    isw[i] = j >= NREG;
    const int px = slot / Q, qs = slot - px * Q;
    const int row = px / IWP, cp = px - row * IWP;
    const int col = SW == 2 ? (cp >= IWP / 2 ? 2 * (cp - IWP / 2) + 1 : 2 * cp) : cp;
    const int qq = qs ^ G_::swz(cp);
    g_pos[i] = (j < NREG && row < IH && col < IW) ? ((row << 16) | (col << 8) | (4 * qq)) : -1;
    const int wl = (j - NREG) * 64 + lane;  // weights: pieces in image order
    voff[i] = (j >= NREG && j < NREG + NW && wl < NWP) ? (unsigned)wl * 16u : OCR_DMA_OOB;
  }
  // buffer descriptors: the weights image of the layer (fixed), the input sample of the unit being loaded (per unit)
  ocr_v4u rs_w, rs_in;
  {
    const unsigned long long wb = (unsigned long long)(CK == 16 ? a.dw_wq16 : a.dw_wq32);
    rs_w.x = __builtin_amdgcn_readfirstlane((unsigned)wb);
    rs_w.y = __builtin_amdgcn_readfirstlane((unsigned)(wb >> 32));
    rs_w.z = 0x80000000u;  // num_records (bytes): valid offsets are far below, OCR_DMA_OOB above
    rs_w.w = 0x00020000u;  // raw buffer, 32-bit data format
    rs_in = rs_w;
  }

  // ---- DMA: global -> LDS, item by item (its own unit / chunk counters run two items ahead of the taps)
  UnitPos<RAG, TH> g_pos_u;
  g_pos_u.init(u0, cblocks, a);
  int g_units = nunits, g_ch = 0, g_buf = 0;
  auto g_setup = [&]() __attribute__((always_inline)) {  // descriptor base and the lanes' offsets for unit g_pos_u
    const unsigned long long ib = (unsigned long long)a.dw_in + (unsigned long long)g_pos_u.in_pix(a) * (unsigned long long)Cs * 4ull;
    rs_in.x = __builtin_amdgcn_readfirstlane((unsigned)ib);
    rs_in.y = __builtin_amdgcn_readfirstlane((unsigned)(ib >> 32));
    const int iy0 = g_pos_u.ty * TH * SH - a.PH, ix0 = g_pos_u.tx * TW * SW - a.PW;
    const int iwn = g_pos_u.in_w(a), ihn = g_pos_u.in_h(a);
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      if (isw[i]) continue;
      const int iy = iy0 + (g_pos[i] >> 16), ix = ix0 + ((g_pos[i] >> 8) & 0xff);
      const bool ok = g_pos[i] >= 0 && (unsigned)iy < (unsigned)ihn && (unsigned)ix < (unsigned)iwn;
      voff[i] = ok ? (unsigned)(((iy * iwn + ix) * Cs + (g_pos[i] & 0xff)) * 4) : OCR_DMA_OOB;
    }
  };
  // Every call issues the same DPW instructions (past the last item it loads the last one again, into a buffer nobody
  // reads): the waits below are exact in-order counts.
  auto DMA = [&]() __attribute__((always_inline)) {
    const unsigned so_in = __builtin_amdgcn_readfirstlane((unsigned)(g_ch * CK * 4));
    const unsigned so_w = __builtin_amdgcn_readfirstlane((unsigned)(g_ch * (K * K + 1) * CK * 4));
    const unsigned base = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)g_buf * BUF + (unsigned)wave * 1024u);
#ifdef OCR_DWPW_NO_G  // development probe: no input traffic (every lane beyond num_records: zeros arrive)
#pragma unroll
    for (int i = 0; i < DPW; ++i) ocr_dma16(base + i * 4096u, OCR_DMA_OOB, rs_w, so_w);
#else
#pragma unroll
    for (int i = 0; i < DPW; ++i) {
      if (isw[i]) ocr_dma16(base + i * 4096u, voff[i], rs_w, so_w);
      else ocr_dma16(base + i * 4096u, voff[i], rs_in, so_in);
    }
#endif
    g_buf = g_buf + 1 == NB ? 0 : g_buf + 1;
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

  // ---- depthwise conv + its epilogue of one chunk, not interleaved (the first item): region rb -> s_op[ob]
  auto DW = [&](int rb, int ob) __attribute__((always_inline)) {
    const char* reg = s_reg + rb * BUF;
    const float* sw = (const float*)(reg + W_OFF) + 4 * q;
    float* so = s_op + ob * OP_TILE;
    F4 acc[PR];
#pragma unroll
    for (int o = 0; o < PR; ++o) { acc[o].lo = ocr_f2{0.f, 0.f}; acc[o].hi = ocr_f2{0.f, 0.f}; }
#pragma unroll
    for (int r = 0; r < (PR - 1) * SH + K; ++r) {
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 v = *(const float4*)(reg + d_off[kx] + r * ROWB);
#pragma unroll
        for (int o = 0; o < PR; ++o) {
          const int ky = r - o * SH;  // compile-time after unrolling
          if (ky < 0 || ky >= K) continue;
          const float4 w = *(const float4*)(sw + (ky * K + kx) * CK);
          acc[o].lo = __builtin_elementwise_fma(ocr_f2{v.x, v.y}, ocr_f2{w.x, w.y}, acc[o].lo);
          acc[o].hi = __builtin_elementwise_fma(ocr_f2{v.z, v.w}, ocr_f2{w.z, w.w}, acc[o].hi);
        }
      }
    }
    const float4 b = *(const float4*)(sw + K * K * CK);
    const ocr_f2 blo = {b.x, b.y}, bhi = {b.z, b.w};
#pragma unroll
    for (int o = 0; o < PR; ++o) {
      dw_hsw6(acc[o], blo, bhi);
      *(float4*)(so + d_op + o * TW * S) = make_float4(acc[o].lo.x, acc[o].lo.y, acc[o].hi.x, acc[o].hi.y);
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
  UnitPos<RAG, TH> m_pos;
  m_pos.init(u0, cblocks, a);
  int b_cb = m_pos.cb;  // column block of the unit whose fragments are being fetched (a unit's column block is its index mod cblocks)
  int b_units = nunits, b_step = 0;
  const float4* const w_lane = (const float4*)c.wfrag + (long)wc * NT * 64 + lane;
  const float4* p_w = w_lane + (long)b_cb * WC * NT * 64;
  float4 bq[C8S][NT];
  auto advanceB = [&]() __attribute__((always_inline)) {  // p_w -> the fragments of the item after the one just fetched
    b_step += C8S;
    if (b_step == KK) {  // next unit: back to the first step of ITS column block (past the end: the last one again)
      b_step = 0;
      if (b_units > 1) { --b_units; b_cb = b_cb + 1 == cblocks ? 0 : b_cb + 1; }
      p_w = w_lane + (long)b_cb * WC * NT * 64;
    } else {
      p_w += C8S * wstride;
    }
  };
  auto loadB = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < C8S; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) bq[j][t] = p_w[j * wstride + t * 64];
    advanceB();
  };
  const float ps6 = a.pw_ep.s6, pa1 = a.pw_ep.a1;
  const int pix = wp * 32 + p, pix_y = pix / TW, pix_x = pix & (TW - 1);
  const int op_off = pix * S + 4 * h;
  // the 1x1 conv's folded epilogue + 16-byte stores for this lane's pixel (as dwpw_kernel)
  auto finish = [&]() __attribute__((always_inline)) {
    const int nt0 = (m_pos.cb * WC + wc) * NT;
    const int oy = m_pos.ty * TH + pix_y, ox = m_pos.tx * TW + pix_x;
    const int own = m_pos.out_w(a);
    const bool inside = oy < m_pos.out_h(a) && ox < own;
    const int r0 = nt0 * 32 + 4 * h;
    const float* sp = s_par + r0;
    const long oidx = (m_pos.out_pix(a) + (long)oy * own + ox) * c.Cs_out + r0;  // element index of the lane's first column
    const ocr_f2 S6 = {ps6, ps6}, A1 = {pa1, pa1};
    // XP: the wave's 32 x 32 block of a column tile leaves through its LDS tile as whole 128-byte lines (conv_device.h,
    // conv_finish): store i of a lane is quad q = lane % 8 of pixel 8 i + lane / 8 of the wave's two tile rows.  Worth 4-8 %
    // on the layers with one column tile, nothing beyond (kernels_dwpw.hip, launch_two).
    float* const xt = s_xp + wave * OCR_XP_FLOATS;
    long xo[4];
    bool xin[4];
    const int xq = lane & 7;
    if constexpr (XP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int xpix = wp * 32 + 8 * i + (lane >> 3);
        const int yy = m_pos.ty * TH + xpix / TW, xx = m_pos.tx * TW + (xpix & (TW - 1));
        xin[i] = yy < m_pos.out_h(a) && xx < own;
        xo[i] = (m_pos.out_pix(a) + (long)yy * own + xx) * c.Cs_out + nt0 * 32 + 4 * xq;
      }
    }
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
        if constexpr (XP) *(float4*)(xt + p * OCR_XP_STRIDE + 8 * g + 4 * h) = make_float4(tl.x, tl.y, th.x, th.y);
        else if (inside && nt0 * 32 + 32 * t + 8 * g < c.ColsStore) st4<false>(c.out, oidx + 32 * t + 8 * g, make_float4(tl.x, tl.y, th.x, th.y));
        acc[t][4 * g] = 0.f; acc[t][4 * g + 1] = 0.f; acc[t][4 * g + 2] = 0.f; acc[t][4 * g + 3] = 0.f;
      }
      if constexpr (XP) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (nt0 * 32 + 32 * t + 4 * xq < c.ColsStore) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (xin[i]) st4<false>(c.out, xo[i] + 32 * t, *(const float4*)(xt + (8 * i + (lane >> 3)) * OCR_XP_STRIDE + 4 * xq));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };
  int m_ch = 0;
  auto mma_end = [&]() __attribute__((always_inline)) -> bool {
    if (++m_ch == nch) {  // the unit is complete
      finish();
      m_ch = 0;
      m_pos.next(cblocks, a);
      return true;
    }
    return false;
  };
  auto MMA = [&](int ob) __attribute__((always_inline)) {
    const float* so = s_op + ob * OP_TILE + op_off;
    float4 av[C8S];
#pragma unroll
    for (int j = 0; j < C8S; ++j) av[j] = *(const float4*)(so + 8 * j);
#pragma unroll
    for (int j = 0; j < C8S; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j][t].x, av[j].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j][t].y, av[j].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j][t].z, av[j].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[j][t].w, av[j].w, acc[t], 0, 0, 0);
      }
    (void)mma_end();
  };

  // ---- steady state: the taps of item k (region buffer rb) inside the matrix instructions of item k-1 (operand ob ^ 1),
  // the depthwise result of item k into operand ob, the fragments of item k into the registers item k-1 has used up
  constexpr int ROWS = (PR - 1) * SH + K, NS = ROWS * K, D = TD, RS = D + 1, NM = C8S * NT * 4, NST = NS + 1;
  auto FUSED = [&](int rb, int ob) __attribute__((always_inline)) {
    // (markers for build.py's ISA check: between them exactly C8S * NT vector loads - the fragment refills the counted waits assume)
    asm volatile("; OCR_DWPW2_FUSED_BEGIN %0" ::"n"(C8S * NT));
    const char* reg = s_reg + rb * BUF;
    const char* col[K];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) col[kx] = reg + d_off[kx];
    const float* wq = (const float*)(reg + W_OFF) + 4 * q;
    float* so = s_op + ob * OP_TILE;
    const float* som = s_op + (ob ^ 1) * OP_TILE + op_off;
    const float4* const pb = p_w;  // the next item's fragments
    float4 av[C8S];
#pragma unroll
    for (int j = 0; j < C8S; ++j) av[j] = *(const float4*)(som + 8 * j);
    F4 dacc[PR];
#pragma unroll
    for (int o = 0; o < PR; ++o) { dacc[o].lo = ocr_f2{0.f, 0.f}; dacc[o].hi = ocr_f2{0.f, 0.f}; }
    float4 tv[RS], tw[RS][PR];
    auto fetch = [&](auto st_, int slot) __attribute__((always_inline)) {
      constexpr int st = decltype(st_)::value;
      constexpr int r = st / K, kx = st - r * K;
      tv[slot] = *(const float4*)(col[kx] + r * ROWB);
#pragma unroll
      for (int o = 0; o < PR; ++o) {
        const int ky = r - o * SH;
        if (ky >= 0 && ky < K) tw[slot][o] = *(const float4*)(wq + (ky * K + kx) * CK);
      }
    };
    auto comp = [](const float4& v, int c4) __attribute__((always_inline)) { return c4 == 0 ? v.x : c4 == 1 ? v.y : c4 == 2 ? v.z : v.w; };
#ifndef OCR_DWPW_NO_TAPS  // development probe (tools/micro/dwpw_probe.hip): no tap reads, no tap FMAs in the steady state
    static_for<D>([&](auto st_) __attribute__((always_inline)) { fetch(st_, decltype(st_)::value % RS); });
#endif
    __builtin_amdgcn_sched_barrier(0);
    static_for<NST>([&](auto st_) __attribute__((always_inline)) {
      constexpr int st = decltype(st_)::value;
      if constexpr (st < NS) {
#ifndef OCR_DWPW_NO_TAPS
        if constexpr (st + D < NS) fetch(std::integral_constant<int, st + D>{}, (st + D) % RS);
        constexpr int r = st / K, slot = st % RS;
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
      } else {  // the depthwise epilogue and the operand write
        const float4 b = *(const float4*)(wq + K * K * CK);
        const ocr_f2 blo = {b.x, b.y}, bhi = {b.z, b.w};
#pragma unroll
        for (int o = 0; o < PR; ++o) {
          dw_hsw6(dacc[o], blo, bhi);
          *(float4*)(so + d_op + o * TW * S) = make_float4(dacc[o].lo.x, dacc[o].lo.y, dacc[o].hi.x, dacc[o].hi.y);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<(st + 1) * NM / NST - st * NM / NST>([&](auto m_) __attribute__((always_inline)) {  // k-ascending per accumulator: octet, column tile, component
        constexpr int m = st * NM / NST + decltype(m_)::value;
        constexpr int j = m / (4 * NT), t = (m / 4) % NT, c4 = m % 4;
#ifdef OCR_DWPW_NO_MMA  // development probe: operands fetched, matrix pipe idle
        if (m == 0) acc[t][0] += bq[j][t].x * av[j].x;
#else
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(bq[j][t], c4), comp(av[j], c4), acc[t], 0, 0, 0);
#endif
        if (c4 == 3) bq[j][t] = pb[j * wstride + t * 64];  // the last use of this fragment register: refill it
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    asm volatile("; OCR_DWPW2_FUSED_END");
    advanceB();
  };

  // ---- pipeline
  for (int i = tid; i < c.NTtot * 32; i += 256) s_par[i] = i < c.ColsStore ? a.pw_ep.bias[i] : 0.f;
  g_setup();
  // Iteration k: taps of item k (buffer k % NB) inside the matrix work of item k-1, the fragments of item k.  The DMA runs
  // NB - 1 items ahead into the buffer the oldest item has left (every wave is past the barrier that closed that item's
  // iteration).  Before the barrier a wave waits until ITS pieces of item k+1 have landed - a counted vmcnt: younger than
  // those are this iteration's fragment refills (and, with three buffers, the DMA just issued); stores of a unit finished an
  // iteration ago are older by now, and the count is the minimum, so a piece of item k+1 never passes as "still in
  // flight".  The 1x1 conv's epilogue and stores of a finished unit come after that wait; nothing else is drained.
  if constexpr (NB == 3) {
    DMA();  // item 0 -> buffer 0
    DMA();  // item 1 -> buffer 1
    ocr_wait_vm<DPW>();  // item 0 has landed (this wave's part)
    __syncthreads();
    DW(0, 0);
    loadB();
    DMA();  // item 2 -> buffer 2
    ocr_wait_vm<C8S * NT + DPW>();  // item 1 has landed (younger: the fragments just requested, item 2)
    __syncthreads();
    int rb = 1;
    for (int k = 1; k < total; ++k) {
      FUSED(rb, k & 1);
      rb = rb + 1 == NB ? 0 : rb + 1;
      DMA();  // item k+2  (round 6: issued in FRONT of item k's taps instead - its buffer is free since the last barrier - the step measured the same, 1271-1290 against 1283-1290 img/s)
      ocr_wait_vm<C8S * NT + DPW>();
      (void)mma_end();
      __syncthreads();
    }
  } else {
    DMA();  // item 0 -> buffer 0
    ocr_wait_vm<0>();
    __syncthreads();
    DMA();  // item 1 -> buffer 1
    DW(0, 0);
    loadB();
    ocr_wait_vm<C8S * NT>();  // item 1 has landed (younger: the fragments just requested)
    __syncthreads();
    for (int k = 1; k < total; ++k) {
      DMA();  // item k+1 -> the buffer item k-1 has left
      FUSED(k & 1, k & 1);
      ocr_wait_vm<C8S * NT>();
      (void)mma_end();
      __syncthreads();
    }
  }
  ocr_wait_vm<0>();  // nothing of this workgroup may still be travelling towards its LDS when it ends
  MMA((total - 1) & 1);
#ifdef OCR_DWPW_CLKRATE
  if (tid == 0) {
    atomicAdd(&ocr_dwpw_clkrate[0], __builtin_readcyclecounter() - clk_c0);
    atomicAdd(&ocr_dwpw_clkrate[1], wall_clock64() - clk_r0);
  }
#endif
}
