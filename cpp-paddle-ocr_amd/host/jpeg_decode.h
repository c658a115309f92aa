// JPEG decoder (sequential and progressive Huffman) for the IPC service (the reference decodes with cv::imread / cv::imdecode, i.e. libjpeg
// with its defaults - /root/reference/src/ocr_ipc_service.cpp:42,336).  The image ships libjpeg.so without
// headers, so this is a restatement of what libjpeg(-turbo) computes with default settings, stage by stage:
//   * Huffman-coded sequential (SOF0 / SOF1) and progressive (SOF2: spectral selection and successive
//     approximation, jdphuff.c) DCT, 8-bit, restart intervals, interleaved and per-component scans,
//     1 or 3 components, sampling 4:4:4, 4:2:2 (h2v1) and 4:2:0 (h2v2);
//   * dequantisation + the "islow" integer IDCT (jidctint.c: 13-bit constants, two passes, PASS1_BITS = 2);
//   * "fancy" (triangle) chroma upsampling, h2v1 and h2v2 (jdsample.c), image edges replicated;
//   * YCbCr -> RGB with the 16-bit fixed-point tables of jdcolor.c.
// tests/test_ipc_service.py checks the output bit for bit against PIL (libjpeg-turbo) on encoded test images.
// Arithmetic coding, lossless, 12-bit, CMYK and other samplings are refused (decode fails).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace PaddleOCR {
namespace jpeg {

struct Huff {
  // canonical code tables: for code length l (1..16): maxcode[l], valptr[l], mincode[l]
  int maxcode[18], valptr[17], mincode[17];
  uint8_t vals[256];
  bool set = false;
  void build(const uint8_t counts[16], const uint8_t* symbols) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += counts[l - 1];
      k += counts[l - 1];
      maxcode[l] = counts[l - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    memcpy(vals, symbols, (size_t)k);
    set = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int bw = 0, bh = 0;        // blocks per row / column (padded to whole MCUs)
  int dw = 0, dh = 0;        // downsampled size in samples (unpadded)
  int pred = 0;
  std::vector<int16_t> coef;   // bw*bh blocks x 64, natural order, not yet dequantised
  std::vector<uint8_t> plane;  // bw*8 x bh*8 samples
};

// Entropy-decoded image: quantised DCT coefficients per component, what the scans of the file say and nothing more.
// The pixel half of decoding (dequantisation, IDCT, upsampling, colour) can then run on the host (Decoder::pixels)
// or on the device (ocr_pipe_stage_jpeg / ocr_jpeg_decode of include/ocr_hip.h) with identical results.
struct Coefs {
  int rows = 0, cols = 0, ncomp = 0, hmax = 1, vmax = 1;
  struct Comp {
    std::vector<int16_t> coef;  // bw*bh blocks x 64, blocks row-major, natural order inside
    uint16_t quant[64] = {};    // natural order
    int bw = 0, bh = 0, dw = 0, dh = 0;
  } comp[3];
};

class Decoder {
 public:
  // decodes to packed BGR; false on anything unsupported or malformed
  bool decode(const uint8_t* data, size_t size, std::vector<uint8_t>& bgr, int& rows, int& cols) {
    if (!parse(data, size)) return false;
    reconstruct();
    return output(bgr, rows, cols);
  }
  // the bit-serial half only
  bool decode_coefficients(const uint8_t* data, size_t size, Coefs& out) {
    if (!parse(data, size)) return false;
    const uint8_t* zz = zigzag();
    out.rows = H_; out.cols = W_; out.ncomp = nc_; out.hmax = hmax_; out.vmax = vmax_;
    for (int i = 0; i < nc_; ++i) {
      Component& c = comp_[i];
      Coefs::Comp& o = out.comp[i];
      o.coef.swap(c.coef);
      for (int k = 0; k < 64; ++k) o.quant[zz[k]] = (uint16_t)qt_[c.tq][k];
      o.bw = c.bw; o.bh = c.bh; o.dw = c.dw; o.dh = c.dh;
    }
    return true;
  }
  // the pixel half on the host, from coefficients (what the device back-end computes)
  static bool pixels(const Coefs& in, std::vector<uint8_t>& bgr, int& rows, int& cols) {
    Decoder d;
    const uint8_t* zz = zigzag();
    d.H_ = in.rows; d.W_ = in.cols; d.nc_ = in.ncomp; d.hmax_ = in.hmax; d.vmax_ = in.vmax;
    for (int i = 0; i < in.ncomp; ++i) {
      Component& c = d.comp_[i];
      const Coefs::Comp& s = in.comp[i];
      c.h = i == 0 ? in.hmax : 1; c.v = i == 0 ? in.vmax : 1; c.tq = i;
      c.bw = s.bw; c.bh = s.bh; c.dw = s.dw; c.dh = s.dh;
      c.coef = s.coef;
      c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
      for (int k = 0; k < 64; ++k) d.qt_[i][k] = s.quant[zz[k]];
    }
    d.reconstruct();
    return d.output(bgr, rows, cols);
  }

 private:
  bool parse(const uint8_t* data, size_t size) {
    d_ = data; n_ = size; pos_ = 0;
    if (n_ < 4 || d_[0] != 0xFF || d_[1] != 0xD8) return false;
    pos_ = 2;
    bool have_frame = false, have_scan = false;
    for (;;) {
      int m = next_marker();
      if (m < 0) { if (have_scan) break; return false; }  // truncated after some data: libjpeg fills in the rest
      if (m == 0xD9) { if (have_scan) break; return false; }
      if (m == 0xC0 || m == 0xC1 || m == 0xC2) { if (have_frame || !read_sof(m == 0xC2)) return false; have_frame = true; continue; }
      if ((m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) return false;  // lossless, differential, arithmetic
      if (m == 0xC4) { if (!read_dht()) return false; continue; }
      if (m == 0xDB) { if (!read_dqt()) return false; continue; }
      if (m == 0xDD) { if (!read_dri()) return false; continue; }
      if (m == 0xDA) {
        if (!have_frame || !read_sos() || !decode_scan()) return false;
        have_scan = true;
        continue;
      }
      if (m >= 0xD0 && m <= 0xD7) continue;  // stray RSTn
      if (!skip_segment()) return false;
    }
    return true;
  }

  // ---------------------------------------------------------------- markers / headers
  int next_marker() {
    while (pos_ + 1 < n_) {
      if (d_[pos_] != 0xFF) { ++pos_; continue; }
      const int m = d_[pos_ + 1];
      if (m == 0x00 || m == 0xFF) { ++pos_; continue; }
      pos_ += 2;
      return m;
    }
    return -1;
  }
  bool seg(size_t& len) {
    if (pos_ + 2 > n_) return false;
    len = ((size_t)d_[pos_] << 8) | d_[pos_ + 1];
    if (len < 2 || pos_ + len > n_) return false;
    return true;
  }
  bool skip_segment() { size_t len; if (!seg(len)) return false; pos_ += len; return true; }
  bool read_dqt() {
    size_t len;
    if (!seg(len)) return false;
    size_t p = pos_ + 2, end = pos_ + len;
    while (p < end) {
      const int pq = d_[p] >> 4, tq = d_[p] & 15;
      ++p;
      if (tq > 3 || pq > 1 || p + (pq ? 128 : 64) > end) return false;
      for (int i = 0; i < 64; ++i) { qt_[tq][i] = pq ? ((d_[p] << 8) | d_[p + 1]) : d_[p]; p += pq ? 2 : 1; }  // zigzag order
    }
    pos_ = end;
    return true;
  }
  bool read_dht() {
    size_t len;
    if (!seg(len)) return false;
    size_t p = pos_ + 2, end = pos_ + len;
    while (p < end) {
      const int tc = d_[p] >> 4, th = d_[p] & 15;
      ++p;
      if (tc > 1 || th > 3 || p + 16 > end) return false;
      int total = 0;
      for (int i = 0; i < 16; ++i) total += d_[p + i];
      if (total > 256 || p + 16 + total > end) return false;
      (tc ? ac_[th] : dc_[th]).build(d_ + p, d_ + p + 16);
      p += 16 + total;
    }
    pos_ = end;
    return true;
  }
  bool read_dri() {
    size_t len;
    if (!seg(len) || len != 4) return false;
    restart_ = (d_[pos_ + 2] << 8) | d_[pos_ + 3];
    pos_ += len;
    return true;
  }
  bool read_sof(bool progressive) {
    progressive_ = progressive;
    size_t len;
    if (!seg(len) || len < 8) return false;
    const uint8_t* q = d_ + pos_ + 2;
    if (q[0] != 8) return false;  // 8-bit precision only
    H_ = (q[1] << 8) | q[2];
    W_ = (q[3] << 8) | q[4];
    nc_ = q[5];
    if (H_ <= 0 || W_ <= 0 || (nc_ != 1 && nc_ != 3) || len < (size_t)(8 + 3 * nc_)) return false;
    if ((long)H_ * W_ > (64L << 20)) return false;  // 64 Mpixel cap: a service must not be made to allocate gigabytes by a header
    hmax_ = vmax_ = 1;
    for (int i = 0; i < nc_; ++i) {
      comp_[i].id = q[6 + 3 * i];
      comp_[i].h = q[7 + 3 * i] >> 4;
      comp_[i].v = q[7 + 3 * i] & 15;
      comp_[i].tq = q[8 + 3 * i];
      if (comp_[i].h < 1 || comp_[i].h > 2 || comp_[i].v < 1 || comp_[i].v > 2 || comp_[i].tq > 3) return false;
      hmax_ = comp_[i].h > hmax_ ? comp_[i].h : hmax_;
      vmax_ = comp_[i].v > vmax_ ? comp_[i].v : vmax_;
    }
    if (nc_ == 1) { comp_[0].h = comp_[0].v = 1; hmax_ = vmax_ = 1; }  // a single component is never subsampled
    if (nc_ == 3) {
      // accepted: luma at the maximum factors, both chroma 1x1 -> 4:4:4, 4:2:2 (2x1), 4:2:0 (2x2)
      if (comp_[0].h != hmax_ || comp_[0].v != vmax_ || comp_[1].h != 1 || comp_[1].v != 1 || comp_[2].h != 1 || comp_[2].v != 1) return false;
      if (hmax_ == 1 && vmax_ == 2) return false;  // h1v2 is not handled
    }
    mcux_ = (W_ + 8 * hmax_ - 1) / (8 * hmax_);
    mcuy_ = (H_ + 8 * vmax_ - 1) / (8 * vmax_);
    for (int i = 0; i < nc_; ++i) {
      Component& c = comp_[i];
      c.bw = mcux_ * c.h;
      c.bh = mcuy_ * c.v;
      c.dw = (W_ * c.h + hmax_ - 1) / hmax_;
      c.dh = (H_ * c.v + vmax_ - 1) / vmax_;
      c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 0);
      c.coef.assign((size_t)c.bw * c.bh * 64, 0);
    }
    pos_ += len;
    return true;
  }
  bool read_sos() {
    size_t len;
    if (!seg(len) || len < 3) return false;  // q[0] below is byte 3 of the segment: a truncated SOS must not be read past
    const uint8_t* q = d_ + pos_ + 2;
    ns_ = q[0];
    if (ns_ < 1 || ns_ > nc_ || len != (size_t)(6 + 2 * ns_)) return false;
    for (int i = 0; i < ns_; ++i) {
      int ci = -1;
      for (int j = 0; j < nc_; ++j) if (comp_[j].id == q[1 + 2 * i]) ci = j;
      if (ci < 0) return false;
      scan_[i] = ci;
      comp_[ci].td = q[2 + 2 * i] >> 4;
      comp_[ci].ta = q[2 + 2 * i] & 15;
      if (comp_[ci].td > 3 || comp_[ci].ta > 3) return false;
    }
    ss_ = q[1 + 2 * ns_]; se_ = q[2 + 2 * ns_]; ah_ = q[3 + 2 * ns_] >> 4; al_ = q[3 + 2 * ns_] & 15;
    if (!progressive_) {
      if (ss_ != 0 || se_ != 63 || ah_ != 0 || al_ != 0) return false;
    } else {
      if (ss_ > se_ || se_ > 63 || (ss_ == 0 && se_ != 0) || (ss_ > 0 && ns_ != 1) || al_ > 13) return false;
    }
    for (int i = 0; i < ns_; ++i) {
      const Component& c = comp_[scan_[i]];
      if ((ss_ == 0 && ah_ == 0 && !dc_[c.td].set) || (se_ > 0 && !ac_[c.ta].set)) return false;
    }
    pos_ += len;
    return true;
  }

  // ---------------------------------------------------------------- entropy decoding
  bool fill(int need) {
    while (bits_ < need) {
      int b = 0;
      if (pos_ < n_ && !hit_marker_) {
        b = d_[pos_];
        if (b == 0xFF) {
          if (pos_ + 1 < n_ && d_[pos_ + 1] == 0x00) pos_ += 2;
          else { hit_marker_ = true; b = 0; }  // a marker: feed zeros, like libjpeg's "insufficient data" path
        } else {
          ++pos_;
        }
      }
      acc_ = (acc_ << 8) | (uint32_t)b;
      bits_ += 8;
    }
    return true;
  }
  int getbits(int n) {
    if (n == 0) return 0;
    fill(n);
    bits_ -= n;
    return (int)((acc_ >> bits_) & ((1u << n) - 1));
  }
  int decode_symbol(const Huff& h) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
      code = (code << 1) | getbits(1);
      if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
    }
    return -1;
  }
  static int extend(int v, int t) { return v < (1 << (t - 1)) ? v - (1 << t) + 1 : v; }
  static const uint8_t* zigzag() {
    static const uint8_t zz[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    return zz;
  }
  // sequential block: DC difference + run/size AC codes
  bool block_sequential(Component& c, int16_t* b) {
    const uint8_t* zz = zigzag();
    const int t = decode_symbol(dc_[c.td]);
    if (t < 0 || t > 15) return false;
    c.pred += t ? extend(getbits(t), t) : 0;
    b[0] = (int16_t)c.pred;
    for (int k = 1; k < 64;) {
      const int rs = decode_symbol(ac_[c.ta]);
      if (rs < 0) return false;
      const int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r != 15) break;  // EOB
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return false;
      b[zz[k]] = (int16_t)extend(getbits(sz), sz);
      ++k;
    }
    return true;
  }
  // progressive, jdphuff.c: DC first / refine, AC first / refine
  bool block_dc_first(Component& c, int16_t* b) {
    const int t = decode_symbol(dc_[c.td]);
    if (t < 0 || t > 15) return false;
    c.pred += t ? extend(getbits(t), t) : 0;
    b[0] = (int16_t)(c.pred * (1 << al_));
    return true;
  }
  bool block_dc_refine(int16_t* b) {
    if (getbits(1)) b[0] = (int16_t)(b[0] | (1 << al_));
    return true;
  }
  bool block_ac_first(Component& c, int16_t* b) {
    const uint8_t* zz = zigzag();
    if (eobrun_ > 0) { --eobrun_; return true; }
    for (int k = ss_; k <= se_; ++k) {
      const int rs = decode_symbol(ac_[c.ta]);
      if (rs < 0) return false;
      const int r = rs >> 4, sz = rs & 15;
      if (sz) {
        k += r;
        if (k > 63) return false;
        b[zz[k]] = (int16_t)(extend(getbits(sz), sz) * (1 << al_));
      } else if (r == 15) {
        k += 15;
      } else {
        eobrun_ = 1 << r;
        if (r) eobrun_ += getbits(r);
        --eobrun_;
        break;
      }
    }
    return true;
  }
  bool block_ac_refine(Component& c, int16_t* b) {
    const uint8_t* zz = zigzag();
    const int p1 = 1 << al_, m1 = -(1 << al_);
    int k = ss_;
    if (eobrun_ == 0) {
      for (; k <= se_; ++k) {
        const int rs = decode_symbol(ac_[c.ta]);
        if (rs < 0) return false;
        int r = rs >> 4, sz = rs & 15, val = 0;
        if (sz) {
          if (sz != 1) return false;
          val = getbits(1) ? p1 : m1;
        } else if (r != 15) {
          eobrun_ = 1 << r;
          if (r) eobrun_ += getbits(r);
          break;  // the rest of this block is handled by the end-of-band logic below
        }
        // skip over already-nonzero coefficients (applying correction bits) and r zero ones
        do {
          int16_t& cf = b[zz[k]];
          if (cf != 0) {
            if (getbits(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
          } else if (--r < 0) {
            break;
          }
          ++k;
        } while (k <= se_);
        if (val && k <= 63) b[zz[k]] = (int16_t)val;
      }
    }
    if (eobrun_ > 0) {
      for (; k <= se_; ++k) {
        int16_t& cf = b[zz[k]];
        if (cf != 0 && getbits(1) && (cf & p1) == 0) cf = (int16_t)(cf + (cf >= 0 ? p1 : m1));
      }
      --eobrun_;
    }
    return true;
  }
  bool decode_one_block(Component& c, int16_t* b) {
    if (!progressive_) return block_sequential(c, b);
    if (ss_ == 0) return ah_ == 0 ? block_dc_first(c, b) : block_dc_refine(b);
    return ah_ == 0 ? block_ac_first(c, b) : block_ac_refine(c, b);
  }

  // ---------------------------------------------------------------- jidctint.c, jpeg_idct_islow
  static int descale(long x, int n) { return (int)((x + (1L << (n - 1))) >> n); }
  static uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
  static void idct(const int* in, uint8_t* out, int stride) {
    const int CB = 13, P1 = 2;
    const long F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137,
               F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    long ws[64];
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = 0; i < 8; ++i) {
        long s[8];
        for (int k = 0; k < 8; ++k) s[k] = pass == 0 ? in[8 * k + i] : ws[8 * i + k];
        long z2 = s[2], z3 = s[6];
        long z1 = (z2 + z3) * F0541;
        long tmp2 = z1 + z3 * (-F1847);
        long tmp3 = z1 + z2 * F0765;
        z2 = s[0]; z3 = s[4];
        long tmp0 = (z2 + z3) << CB, tmp1 = (z2 - z3) << CB;
        const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = s[7]; tmp1 = s[5]; tmp2 = s[3]; tmp3 = s[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        long z4 = tmp1 + tmp3;
        const long z5 = (z3 + z4) * F1175;
        tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const long o[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
        if (pass == 0) {
          for (int k = 0; k < 8; ++k) ws[8 * k + i] = descale(o[k], CB - P1);
        } else {
          for (int k = 0; k < 8; ++k) out[i * stride + k] = clamp8(descale(o[k], CB + P1 + 3) + 128);
        }
      }
    }
  }

  bool restart_if_due(int& until_restart, int& next_rst) {
    if (!restart_ || until_restart != 0) return true;
    bits_ = 0; acc_ = 0; hit_marker_ = false;  // byte-align, expect RSTn
    while (pos_ + 1 < n_ && !(d_[pos_] == 0xFF && d_[pos_ + 1] >= 0xD0 && d_[pos_ + 1] <= 0xD7)) ++pos_;
    if (pos_ + 1 >= n_ || d_[pos_ + 1] != 0xD0 + next_rst) return false;
    pos_ += 2;
    next_rst = (next_rst + 1) & 7;
    for (int i = 0; i < nc_; ++i) comp_[i].pred = 0;
    eobrun_ = 0;
    until_restart = restart_;
    return true;
  }
  bool decode_scan() {
    acc_ = 0; bits_ = 0; hit_marker_ = false; eobrun_ = 0;
    for (int i = 0; i < nc_; ++i) comp_[i].pred = 0;
    int until_restart = restart_, next_rst = 0;
    if (ns_ == 1) {
      // one component: its own block grid, not padded to whole MCUs
      Component& c = comp_[scan_[0]];
      const int nbx = (c.dw + 7) / 8, nby = (c.dh + 7) / 8;
      for (int by = 0; by < nby; ++by)
        for (int bx = 0; bx < nbx; ++bx) {
          if (!restart_if_due(until_restart, next_rst)) return false;
          if (!decode_one_block(c, &c.coef[((size_t)by * c.bw + bx) * 64])) return false;
          if (restart_) --until_restart;
        }
      return true;
    }
    for (int my = 0; my < mcuy_; ++my)
      for (int mx = 0; mx < mcux_; ++mx) {
        if (!restart_if_due(until_restart, next_rst)) return false;
        for (int i = 0; i < ns_; ++i) {
          Component& c = comp_[scan_[i]];
          for (int by = 0; by < c.v; ++by)
            for (int bx = 0; bx < c.h; ++bx)
              if (!decode_one_block(c, &c.coef[((size_t)(my * c.v + by) * c.bw + mx * c.h + bx) * 64])) return false;
        }
        if (restart_) --until_restart;
      }
    return true;
  }
  // dequantise + IDCT every block
  void reconstruct() {
    const uint8_t* zz = zigzag();
    int q[64], in[64];
    for (int i = 0; i < nc_; ++i) {
      Component& c = comp_[i];
      for (int k = 0; k < 64; ++k) q[zz[k]] = qt_[c.tq][k];
      for (int by = 0; by < c.bh; ++by)
        for (int bx = 0; bx < c.bw; ++bx) {
          const int16_t* b = &c.coef[((size_t)by * c.bw + bx) * 64];
          for (int k = 0; k < 64; ++k) in[k] = b[k] * q[k];
          idct(in, c.plane.data() + (size_t)by * 8 * c.bw * 8 + bx * 8, c.bw * 8);
        }
    }
  }

  // ---------------------------------------------------------------- jdsample.c fancy upsampling
  // full-resolution chroma plane (W_ x H_) from a component plane
  void upsample(const Component& c, std::vector<uint8_t>& full) const {
    const int stride = c.bw * 8;
    full.assign((size_t)W_ * H_, 0);
    if (c.h == hmax_ && c.v == vmax_) {
      for (int y = 0; y < H_; ++y) memcpy(&full[(size_t)y * W_], &c.plane[(size_t)y * stride], (size_t)W_);
      return;
    }
    const int dw = c.dw, dh = c.dh;
    std::vector<int> colsum((size_t)dw);
    std::vector<uint8_t> row((size_t)2 * dw + 2);
    if (vmax_ == 1) {  // h2v1
      for (int y = 0; y < H_; ++y) {
        const uint8_t* in = &c.plane[(size_t)y * stride];
        if (dw == 1) { row[0] = row[1] = in[0]; }
        else {
          row[0] = in[0];
          row[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
          for (int x = 1; x < dw - 1; ++x) {
            const int v = in[x] * 3;
            row[2 * x] = (uint8_t)((v + in[x - 1] + 1) >> 2);
            row[2 * x + 1] = (uint8_t)((v + in[x + 1] + 2) >> 2);
          }
          row[2 * dw - 2] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
          row[2 * dw - 1] = in[dw - 1];
        }
        memcpy(&full[(size_t)y * W_], row.data(), (size_t)W_);
      }
      return;
    }
    // h2v2: output rows 2r and 2r+1 from input row r and its upper / lower neighbour (edges replicated)
    for (int y = 0; y < H_; ++y) {
      const int r = y >> 1;
      const int rn = (y & 1) ? (r + 1 < dh ? r + 1 : dh - 1) : (r > 0 ? r - 1 : 0);
      const uint8_t* in0 = &c.plane[(size_t)r * stride];
      const uint8_t* in1 = &c.plane[(size_t)rn * stride];
      for (int x = 0; x < dw; ++x) colsum[x] = in0[x] * 3 + in1[x];
      if (dw == 1) { row[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4); row[1] = (uint8_t)((colsum[0] * 4 + 7) >> 4); }
      else {
        row[0] = (uint8_t)((colsum[0] * 4 + 8) >> 4);
        row[1] = (uint8_t)((colsum[0] * 3 + colsum[1] + 7) >> 4);
        for (int x = 1; x < dw - 1; ++x) {
          row[2 * x] = (uint8_t)((colsum[x] * 3 + colsum[x - 1] + 8) >> 4);
          row[2 * x + 1] = (uint8_t)((colsum[x] * 3 + colsum[x + 1] + 7) >> 4);
        }
        row[2 * dw - 2] = (uint8_t)((colsum[dw - 1] * 3 + colsum[dw - 2] + 8) >> 4);
        row[2 * dw - 1] = (uint8_t)((colsum[dw - 1] * 4 + 7) >> 4);
      }
      memcpy(&full[(size_t)y * W_], row.data(), (size_t)W_);
    }
  }

  // ---------------------------------------------------------------- jdcolor.c ycc_rgb_convert
  bool output(std::vector<uint8_t>& bgr, int& rows, int& cols) const {
    rows = H_; cols = W_;
    bgr.resize((size_t)W_ * H_ * 3);
    const int ystride = comp_[0].bw * 8;
    if (nc_ == 1) {
      for (int y = 0; y < H_; ++y)
        for (int x = 0; x < W_; ++x) {
          const uint8_t v = comp_[0].plane[(size_t)y * ystride + x];
          uint8_t* o = &bgr[((size_t)y * W_ + x) * 3];
          o[0] = o[1] = o[2] = v;
        }
      return true;
    }
    std::vector<uint8_t> cb, cr;
    upsample(comp_[1], cb);
    upsample(comp_[2], cr);
    int crr[256], cbb[256];
    long crg[256], cbg[256];
    for (int i = 0; i < 256; ++i) {
      const long x = i - 128;
      crr[i] = (int)((91881L * x + 32768) >> 16);
      cbb[i] = (int)((116130L * x + 32768) >> 16);
      crg[i] = -46802L * x;
      cbg[i] = -22554L * x + 32768;
    }
    for (int y = 0; y < H_; ++y)
      for (int x = 0; x < W_; ++x) {
        const int Y = comp_[0].plane[(size_t)y * ystride + x];
        const int b = cb[(size_t)y * W_ + x], r = cr[(size_t)y * W_ + x];
        uint8_t* o = &bgr[((size_t)y * W_ + x) * 3];
        o[2] = clamp8(Y + crr[r]);
        o[1] = clamp8(Y + (int)((cbg[b] + crg[r]) >> 16));
        o[0] = clamp8(Y + cbb[b]);
      }
    return true;
  }

  const uint8_t* d_ = nullptr;
  size_t n_ = 0, pos_ = 0;
  int qt_[4][64] = {};
  Huff dc_[4], ac_[4];
  Component comp_[3];
  int W_ = 0, H_ = 0, nc_ = 0, hmax_ = 1, vmax_ = 1, mcux_ = 0, mcuy_ = 0, restart_ = 0;
  bool progressive_ = false;
  int ns_ = 0, scan_[3] = {0, 0, 0}, ss_ = 0, se_ = 63, ah_ = 0, al_ = 0, eobrun_ = 0;
  uint32_t acc_ = 0;
  int bits_ = 0;
  bool hit_marker_ = false;
};

}  // namespace jpeg
}  // namespace PaddleOCR
