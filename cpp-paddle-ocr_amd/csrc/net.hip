#include "net.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace ocr {

#define HIP_OK(expr)                                                                       \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess) {                                                                \
      err = std::string(#expr) + ": " + hipGetErrorString(_e);                             \
      return false;                                                                        \
    }                                                                                      \
  } while (0)

// ------------------------------------------------------------------ plan text -> Plan
static std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> r;
  std::stringstream ss(s);
  std::string t;
  while (std::getline(ss, t, sep)) r.push_back(t);
  return r;
}

bool parse_plan(const char* text, Plan& plan, std::string& err) {
  std::stringstream ss(text);
  std::string line;
  while (std::getline(ss, line)) {
    if (line.empty() || line[0] == '#') continue;
    auto toks = split(line, ' ');
    if (toks[0] == "plan") {
      plan.name = toks.size() > 1 ? toks[1] : "";
      for (auto& t : toks)
        if (t.rfind("ntensors=", 0) == 0) plan.ntensors = atoi(t.c_str() + 9);
      continue;
    }
    PlanOp op;
    const std::string& k = toks[0];
    if (k == "conv") op.kind = PlanOp::CONV;
    else if (k == "dw") op.kind = PlanOp::DW;
    else if (k == "deconv") op.kind = PlanOp::DECONV;
    else if (k == "linear") op.kind = PlanOp::LINEAR;
    else if (k == "sefc") op.kind = PlanOp::SEFC;
    else if (k == "gap") op.kind = PlanOp::GAP;
    else if (k == "pool") op.kind = PlanOp::POOL;
    else if (k == "ew") op.kind = PlanOp::EW;
    else if (k == "concat") op.kind = PlanOp::CONCAT;
    else if (k == "ln") op.kind = PlanOp::LN;
    else if (k == "attn") op.kind = PlanOp::ATTN;
    else if (k == "softmax") op.kind = PlanOp::SOFTMAX;
    else if (k == "output") op.kind = PlanOp::OUTPUT;
    else { err = "plan: unknown op " + k; return false; }
    for (size_t i = 1; i < toks.size(); ++i) {
      auto eq = toks[i].find('=');
      if (eq == std::string::npos) { err = "plan: bad token " + toks[i]; return false; }
      const std::string key = toks[i].substr(0, eq), v = toks[i].substr(eq + 1);
      auto I = [&]() { return atoi(v.c_str()); };
      auto F = [&]() { return strtof(v.c_str(), nullptr); };
      if (key == "i") {
        if (op.kind == PlanOp::CONCAT) for (auto& t : split(v, ',')) op.ins.push_back(atoi(t.c_str()));
        else op.in = I();
      } else if (key == "up") { for (auto& t : split(v, ',')) op.ups.push_back(atoi(t.c_str())); }
      else if (key == "o") op.out = I();
      else if (key == "cin") op.cin = I();
      else if (key == "cout") op.cout = I();
      else if (key == "c") op.c = I();
      else if (key == "cr") op.cr = I();
      else if (key == "kh") op.kh = I();
      else if (key == "kw") op.kw = I();
      else if (key == "sh") op.sh = I();
      else if (key == "sw") op.sw = I();
      else if (key == "ph") op.ph = I();
      else if (key == "pw") op.pw = I();
      else if (key == "type") op.pool_max = (v == "max");
      else if (key == "heads") op.heads = I();
      else if (key == "hd") op.hd = I();
      else if (key == "scale") op.scale = F();
      else if (key == "eps") op.eps = F();
      else if (key == "slope") op.slope = F();
      else if (key == "offset") op.offset = F();
      else if (key == "w") op.w = v;
      else if (key == "w1") op.w1 = v;
      else if (key == "b1") op.b1 = v;
      else if (key == "w2") op.w2 = v;
      else if (key == "b2") op.b2 = v;
      else if (key == "g") op.g = v;
      else if (key == "b") op.b = v;
      else if (key == "ep") {
        for (auto& st : split(v, '|')) {
          auto c = st.find(':');
          const std::string sk = st.substr(0, c);
          auto args = split(st.substr(c + 1), ',');
          PlanStage s;
          if (sk == "bias") { s.kind = EP_BIAS; s.n0 = args[0]; }
          else if (sk == "smul") { s.kind = EP_SMUL; s.n0 = args[0]; }
          else if (sk == "sadd") { s.kind = EP_SADD; s.n0 = args[0]; }
          else if (sk == "bn") { s.kind = EP_BN; s.n0 = args[0]; s.n1 = args[1]; s.n2 = args[2]; s.n3 = args[3]; s.p0 = strtof(args[4].c_str(), nullptr); }
          else if (sk == "act") {
            s.kind = EP_ACT;
            if (args[0] == "relu") s.act = ACT_RELU;
            else if (args[0] == "hswish") s.act = ACT_HSWISH;
            else if (args[0] == "hsig") { s.act = ACT_HSIG; s.p0 = strtof(args[1].c_str(), nullptr); s.p1 = strtof(args[2].c_str(), nullptr); }
            else if (args[0] == "swish") s.act = ACT_SWISH;
            else if (args[0] == "sigmoid") s.act = ACT_SIGMOID;
            else { err = "plan: unknown act " + args[0]; return false; }
          }
          else if (sk == "mulc") { s.kind = EP_MULC; s.tid = atoi(args[0].c_str()); }
          else if (sk == "addt") { s.kind = EP_ADDT; s.tid = atoi(args[0].c_str()); }
          else if (sk == "addup") { s.kind = EP_ADDUP; s.tid = atoi(args[0].c_str()); s.up = atoi(args[1].c_str()); }
          else { err = "plan: unknown stage " + sk; return false; }
          op.ep.push_back(s);
        }
      } else { err = "plan: unknown key " + key; return false; }
    }
    plan.ops.push_back(op);
  }
  if (plan.ntensors <= 0) { err = "plan: missing header"; return false; }
  return true;
}

// ------------------------------------------------------------------ weights
Net::~Net() {
  for (auto& kv : dev_) (void)g_free(kv.second);
  if (arena_) (void)g_free(arena_);
  if (gap_part_) (void)g_free(gap_part_);
  if (head_part_) (void)g_free(head_part_);
  cache_.clear();
  for (auto e : ev_pool_) (void)hipEventDestroy(e);
  for (auto& p : ev_pending_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
}

float* Net::upload(const std::string& key, const std::vector<float>& v) {
  auto it = dev_.find(key);
  if (it != dev_.end()) return it->second;
  float* d = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 4) * sizeof(float);
  if (g_malloc(&d, bytes) != hipSuccess) return nullptr;
  if (!v.empty() && g_memcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  dev_[key] = d;
  return d;
}

const float* Net::dev_vec(const std::string& key) const {
  auto it = dev_.find(key);
  return it == dev_.end() ? nullptr : it->second;
}

// per-channel vector -> physical order, zero padded
static std::vector<float> perm_vec(const std::vector<float>& v, int C, bool plain) {
  if (plain) {  // logical order, zero padded to whole 32-channel tiles (+ one float4) for vector loads
    std::vector<float> o(((size_t)C + 31) / 32 * 32 + 4, 0.f);
    for (int c = 0; c < C; ++c) o[c] = v[c];
    return o;
  }
  std::vector<float> o(c8i_stride(C), 0.f);
  for (int c = 0; c < C; ++c) o[c8i_phys(c)] = v[c];
  return o;
}

// B-fragment image for conv_mfma_kernel: [tap][C8][NTtot][lane][4];
// get(col, k, tap) returns the weight of GEMM column `col`, input channel k, tap.
template <class Get>
static std::vector<float> build_frag(int taps, int cin, int cols, Get get) {
  const int C8 = c8i_stride(cin) / 8, tiles = (cols + 31) / 32, g = conv_nt_for(tiles);
  const int NT = (tiles + g - 1) / g * g;  // whole groups of the launch's tiles-per-wave (zero tiles pad)
  std::vector<float> f((size_t)taps * C8 * NT * 64 * 4, 0.f);
  for (int tap = 0; tap < taps; ++tap)
    for (int c8 = 0; c8 < C8; ++c8)
      for (int nt = 0; nt < NT; ++nt)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int col = nt * 32 + (lane & 31);
            const int k = c8 * 8 + 2 * s + (lane >> 5);
            float v = 0.f;
            if (col < cols && k < cin) v = get(col, k, tap);
            f[((((size_t)tap * C8 + c8) * NT + nt) * 64 + lane) * 4 + s] = v;
          }
  return f;
}

// precision "fp16": the same image in f16 (round to nearest even), four halfs per lane packed into the float vector
// upload() takes - what the HALF instantiations of the matrix-core kernels read as uint2 per lane (conv_device.h)
// the same for v_mfma_f32_32x32x16_f16 (single-tap convs, an even number of octets): [octet pair s][column tile][64 lanes][8
// halfs] - lane (p, h) holds the eight physical channels of octet 2s + h = what lanes (p, 0) and (p, 1) of that octet hold in
// the image above
static std::vector<float> frag_to_half_x16(const std::vector<float>& f, int C8, int taps = 1) {
  const size_t nt = f.size() / ((size_t)taps * C8 * 256);  // column tiles
  std::vector<float> out(f.size() / 2, 0.f);
  uint16_t* hp = reinterpret_cast<uint16_t*>(out.data());
  for (int tap = 0; tap < taps; ++tap)
    for (int s = 0; s < C8 / 2; ++s)
      for (size_t t = 0; t < nt; ++t)
        for (int lane = 0; lane < 64; ++lane) {
          const int p = lane & 31, h = lane >> 5, j = 2 * s + h;
          for (int i = 0; i < 8; ++i) {
            const _Float16 v = (_Float16)f[((((size_t)tap * C8 + j) * nt + t) * 64 + (p + 32 * (i >> 2))) * 4 + (i & 3)];
            memcpy(&hp[((((size_t)tap * (C8 / 2) + s) * nt + t) * 64 + lane) * 8 + i], &v, sizeof(uint16_t));
          }
        }
  return out;
}
static std::vector<float> frag_to_half(const std::vector<float>& f) {
  std::vector<float> out((f.size() + 1) / 2, 0.f);
  uint16_t* h = reinterpret_cast<uint16_t*>(out.data());
  for (size_t i = 0; i < f.size(); ++i) {
    const _Float16 v = (_Float16)f[i];
    memcpy(&h[i], &v, sizeof(uint16_t));
  }
  return out;
}


// ------------------------------------------------------------------ LAB fold (round 5 contract)
// The exported PP-LCNetV3 graphs carry  bias b | * s0 | + a0 [| hswish | * s1 | + a1]  behind every backbone conv (scalars
// s, a).  Executed stage by stage that chain is ~17 VALU instructions per pair of values - and on this
// chip every epilogue instruction of a matrix-core kernel is matrix time.  Like every inference engine's
// scale / shift folding, the loader rewrites it ONCE:
//   w'   = (float)((double)w * s0 * s_in)                   every weight, one rounding
//   b'_o = (float)(s0 * (b_o + a_in * sum_k w_ok) + a0)     in double, k ascending, one rounding
//   stages: bias b' [| act hsw6: u = y * clamp(y + 3, 0, 6) [| sfma: fmaf(u, s6, a1), s6 = (float)(s1 / 6.0)]]
// (s_in, a_in) = (1, 0), or - ABSORPTION - the (s6, a1) of a depthwise conv with the full chain whose output's only reader
// is this op, a 1x1 stride-1 conv that carries the chain itself (a 1x1 conv has no padding, the affine map of its input is
// exact in value): that depthwise conv then stops after hsw6, its tensor holds u.  No division, no range sweep, 5-6 VALU
// instructions per pair.  The CPU checker's loader (under oracle/) performs the same operations in the same
// order, so parity stays bit for bit; against the unfolded graph in float64 the error is what it was (tests/).
struct LabChain { bool on = false, act = false; float s0 = 1, a0 = 0, s1 = 1, a1 = 0; };
static float lab_s6(float s1) { return (float)((double)s1 / 6.0); }

static bool fold_lab(Plan& plan, WeightMap& W, std::string& err) {
  const size_t nops = plan.ops.size();
  auto need = [&](const std::string& n) -> const HostTensor* {
    auto it = W.find(n);
    if (it == W.end()) { err = "weights: missing parameter '" + n + "' (graph/params mismatch)"; return nullptr; }
    return &it->second;
  };
  std::vector<LabChain> lab(nops);
  std::vector<int> uses(plan.ntensors, 0), absorbs(nops, -1);
  std::vector<char> handed(nops, 0);
  int out_tid = -1;
  for (size_t oi = 0; oi < nops; ++oi) {
    const PlanOp& op = plan.ops[oi];
    if (op.kind == PlanOp::OUTPUT) { out_tid = op.in; continue; }
    if (op.in >= 0) uses[op.in]++;
    for (int t : op.ins) uses[t]++;
    for (auto& st : op.ep) if (st.tid >= 0) uses[st.tid]++;
    if (op.kind != PlanOp::CONV && op.kind != PlanOp::DW) continue;
    const auto& e = op.ep;
    const bool p3 = e.size() >= 3 && e[0].kind == EP_BIAS && e[1].kind == EP_SMUL && e[2].kind == EP_SADD;
    const bool p6 = p3 && e.size() == 6 && e[3].kind == EP_ACT && e[3].act == ACT_HSWISH && e[4].kind == EP_SMUL && e[5].kind == EP_SADD;
    if (!(p6 || (p3 && e.size() == 3))) continue;
    const HostTensor *s0 = need(e[1].n0), *a0 = need(e[2].n0);
    if (!s0 || !a0) return false;
    LabChain& L = lab[oi];
    L.on = true; L.s0 = s0->data[0]; L.a0 = a0->data[0];
    if (p6) {
      const HostTensor *s1 = need(e[4].n0), *a1 = need(e[5].n0);
      if (!s1 || !a1) return false;
      L.act = true; L.s1 = s1->data[0]; L.a1 = a1->data[0];
    }
  }
  for (size_t oi = 0; oi < nops; ++oi) {
    const PlanOp& d = plan.ops[oi];
    if (d.kind != PlanOp::DW || !lab[oi].act || d.out == out_tid || uses[d.out] != 1) continue;
    for (size_t oj = oi + 1; oj < nops; ++oj) {
      const PlanOp& c = plan.ops[oj];
      if (c.kind == PlanOp::CONCAT || c.kind == PlanOp::OUTPUT || c.in != d.out) continue;
      if (c.kind == PlanOp::CONV && lab[oj].on && c.kh == 1 && c.kw == 1 && c.sh == 1 && c.sw == 1 && c.ph == 0 && c.pw == 0) { absorbs[oj] = (int)oi; handed[oi] = 1; }
      break;
    }
  }
  for (size_t oi = 0; oi < nops; ++oi) {
    if (!lab[oi].on) continue;
    PlanOp& op = plan.ops[oi];
    const LabChain& L = lab[oi];
    const HostTensor* b = need(op.ep[0].n0);
    const HostTensor* w = need(op.w);
    if (!b || !w) return false;
    const size_t C = b->data.size();
    if (!C || w->data.size() % C) { err = "LAB fold: filter / bias size mismatch " + op.w; return false; }
    const size_t per = w->data.size() / C;  // weights per output channel (conv: [co][ci][kh][kw]; depthwise: [c][1][kh][kw])
    double s_in = 1.0, a_in = 0.0;
    if (absorbs[oi] >= 0) { s_in = (double)lab_s6(lab[absorbs[oi]].s1); a_in = (double)lab[absorbs[oi]].a1; }
    const std::string tag = "#lab" + std::to_string(oi);
    HostTensor wf = *w, bf = *b;
    for (size_t i = 0; i < wf.data.size(); ++i) { const double t = (double)w->data[i] * (double)L.s0; wf.data[i] = (float)(t * s_in); }
    for (size_t o = 0; o < C; ++o) {
      double sum = 0.0;
      if (absorbs[oi] >= 0) for (size_t k = 0; k < per; ++k) sum = sum + (double)w->data[o * per + k];
      const double t = a_in * sum;
      const double u = (double)b->data[o] + t;
      const double v = (double)L.s0 * u;
      bf.data[o] = (float)(v + (double)L.a0);
    }
    std::vector<PlanStage> ne;
    PlanStage sb; sb.kind = EP_BIAS; sb.n0 = op.ep[0].n0 + tag;
    ne.push_back(sb);
    if (L.act) {
      PlanStage sa; sa.kind = EP_ACT; sa.act = ACT_HSW6;
      ne.push_back(sa);
      if (!handed[oi]) { PlanStage sf; sf.kind = EP_SFMA; sf.p0 = lab_s6(L.s1); sf.p1 = L.a1; ne.push_back(sf); }
    }
    W[sb.n0] = std::move(bf);
    W[op.w + tag] = std::move(wf);
    op.w += tag;
    op.ep = std::move(ne);
  }
  return true;
}

bool Net::load(const char* plan_text, const WeightMap& W_in, std::string& err, bool half) {
  half_ = half;
  if (!parse_plan(plan_text, plan_, err)) return false;
  WeightMap W = W_in;  // + the folded filters / biases, under their own names
  if (!fold_lab(plan_, W, err)) return false;
  { const char* e = getenv("OCR_GRAPH"); graphs_ = !(e && e[0] == '0'); }
  if (const char* e = getenv("OCR_NET_BINDINGS")) max_bindings_ = (size_t)std::min(4096L, std::max(2L, atol(e)));  // (tests: a small cap forces evictions)
  host_w_ = W;
  auto need = [&](const std::string& n) -> const HostTensor* {
    auto it = W.find(n);
    if (it == W.end()) { err = "weights: missing parameter '" + n + "' (graph/params mismatch)"; return nullptr; }
    return &it->second;
  };
  // which tensors are kept in plain (logical) channel order
  std::vector<char> plain(plan_.ntensors, 0);
  plain[0] = 1;
  for (auto& op : plan_.ops) {
    if (op.kind == PlanOp::SOFTMAX) { plain[op.in] = 1; plain[op.out] = 1; }
    if (op.kind == PlanOp::DECONV && op.cout == 1) plain[op.out] = 1;
    if (op.kind == PlanOp::OUTPUT) out_tid_ = op.in;
  }
  tensors_.assign(plan_.ntensors, TensorDesc());
  for (int i = 0; i < plan_.ntensors; ++i) tensors_[i].plain = plain[i];
  // precision "fp16": every C8I activation tensor is stored as f16, except the per-image vectors (pool results, SE gates and
  // what is computed from them): they are reduction results, a few KB, and feed every pixel of their image
  if (half_) {
    std::vector<char> vec(plan_.ntensors, 0);
    for (auto& op : plan_.ops) {
      if (op.out < 0) continue;
      if (op.kind == PlanOp::GAP || op.kind == PlanOp::SEFC) vec[op.out] = 1;
      else if (op.kind == PlanOp::EW && op.in >= 0 && vec[op.in]) vec[op.out] = 1;
    }
    for (int i = 0; i < plan_.ntensors; ++i) tensors_[i].f16 = !plain[i] && !vec[i];
  }

  for (auto& op : plan_.ops) {
    // epilogue parameter images
    const int oc = (op.kind == PlanOp::CONV || op.kind == PlanOp::DECONV || op.kind == PlanOp::LINEAR) ? op.cout : op.c;
    const bool oplain = op.out >= 0 && plain[op.out];
    for (auto& st : op.ep) {
      if (st.kind == EP_BIAS) {
        auto p = need(st.n0); if (!p) return false;
        if ((int)p->numel() != oc) { err = "bias size mismatch " + st.n0; return false; }
        if (!upload((oplain ? "vecp:" : "vec:") + st.n0, perm_vec(p->data, oc, oplain))) { err = "hipMalloc failed"; return false; }
      } else if (st.kind == EP_SMUL || st.kind == EP_SADD) {
        auto p = need(st.n0); if (!p) return false;
        scalars_[st.n0] = p->data[0];
      } else if (st.kind == EP_BN) {
        auto g = need(st.n0), b = need(st.n1), m = need(st.n2), v = need(st.n3);
        if (!g || !b || !m || !v) return false;
        for (const HostTensor* t : {g, b, m, v})
          if ((int)t->numel() != oc) { err = "batch-norm vector size mismatch near " + st.n0 + " (different architecture?)"; return false; }
        std::vector<float> sc(oc), sh(oc);
        for (int c = 0; c < oc; ++c) {
          const float inv = 1.0f / sqrtf(v->data[c] + st.p0);
          const float s = g->data[c] * inv;
          const float mi = m->data[c] * inv;
          const float ms = mi * g->data[c];
          sc[c] = s;
          sh[c] = b->data[c] - ms;
        }
        if (!upload("bns:" + st.n0, perm_vec(sc, oc, oplain)) || !upload("bnt:" + st.n0, perm_vec(sh, oc, oplain))) { err = "hipMalloc failed"; return false; }
      }
    }
    switch (op.kind) {
      case PlanOp::CONV: {
        auto p = need(op.w); if (!p) return false;
        const int co = op.cout, ci = op.cin, kh = op.kh, kw = op.kw;
        if (p->dims.size() != 4 || p->dims[0] != co || p->dims[1] != ci || p->dims[2] != kh || p->dims[3] != kw) { err = "conv filter shape mismatch " + op.w; return false; }
        const float* w = p->data.data();
        if (ci == 3) {
          const int cs = c8i_stride(co);
          std::vector<float> img((size_t)kh * kw * 3 * cs, 0.f);
          for (int o = 0; o < co; ++o)
            for (int c = 0; c < 3; ++c)
              for (int t = 0; t < kh * kw; ++t) img[((size_t)t * 3 + c) * cs + c8i_phys(o)] = w[((size_t)o * 3 + c) * kh * kw + t];
          if (!upload("stem:" + op.w, img)) { err = "hipMalloc failed"; return false; }
        } else {
          if (op.sh != 1 || op.sw != 1) { err = "strided dense conv with Cin != 3 is not on this path"; return false; }
          // GEMM row R of the fragment image is PHYSICAL output channel R (logical for plain outputs)
          const bool pl = plain[op.out];
          const int cols = pl ? co : c8i_stride(co);
          auto f = build_frag(kh * kw, ci, cols, [&](int col, int k, int tap) {
            const int ch = pl ? col : c8i_logical(col);
            return ch < co ? w[((size_t)ch * ci + k) * kh * kw + tap] : 0.f;
          });
          if (!upload("frag:" + op.w, f)) { err = "hipMalloc failed"; return false; }
          if (half_ && !upload("frag16:" + op.w, frag_to_half(f))) { err = "hipMalloc failed"; return false; }
          if (half_ && ((kh == 1 && kw == 1) || (kh == 3 && kw == 3 && ci == 96)) && !pl && (c8i_stride(ci) / 8) % 2 == 0 &&
              !upload("frag16x:" + op.w, frag_to_half_x16(f, c8i_stride(ci) / 8, kh * kw))) { err = "hipMalloc failed"; return false; }
          if (kh == 3 && kw == 3 && ci == 96 && co == 24 && !pl && !upload("c24:" + op.w, conv3x3_c24_image(w, co, ci))) { err = "hipMalloc failed"; return false; }
          if (kh == 1 && kw == 1 && ci <= 24 && !pl && op.ep.empty()) {  // conv_rowsum_kernel's image (RSE blocks): [k logical, padded][physical column]
            const int cs_in = c8i_stride(ci);
            std::vector<float> img((size_t)cs_in * cols, 0.f);
            for (int k = 0; k < ci; ++k)
              for (int col = 0; col < cols; ++col) {
                const int ch = c8i_logical(col);
                if (ch < co) img[(size_t)k * cols + col] = w[(size_t)ch * ci + k];
              }
            if (!upload("rsw:" + op.w, img)) { err = "hipMalloc failed"; return false; }
          }
        }
      } break;
      case PlanOp::LINEAR: {
        auto p = need(op.w); if (!p) return false;
        const int ci = op.cin, co = op.cout;
        if (p->dims.size() != 2 || p->dims[0] != ci || p->dims[1] != co) { err = "linear weight shape mismatch " + op.w; return false; }
        const float* w = p->data.data();
        const bool pl = plain[op.out];
        const int cols = pl ? co : c8i_stride(co);
        auto f = build_frag(1, ci, cols, [&](int col, int k, int) {
          const int ch = pl ? col : c8i_logical(col);
          return ch < co ? w[(size_t)k * co + ch] : 0.f;
        });
        if (!upload("frag:" + op.w, f)) { err = "hipMalloc failed"; return false; }
        if (half_ && !upload("frag16:" + op.w, frag_to_half(f))) { err = "hipMalloc failed"; return false; }
      } break;
      case PlanOp::DECONV: {
        auto p = need(op.w); if (!p) return false;
        const int ci = op.cin, co = op.cout;
        if (p->dims.size() != 4 || p->dims[0] != ci || p->dims[1] != co || p->dims[2] != 2 || p->dims[3] != 2) { err = "deconv filter shape mismatch " + op.w; return false; }
        const float* w = p->data.data();
        if (co == 1) {
          std::vector<float> img((size_t)ci * 4);
          for (int c = 0; c < ci; ++c)
            for (int q = 0; q < 4; ++q) img[(size_t)c * 4 + q] = w[(size_t)c * 4 + q];
          if (!upload("tail:" + op.w, img)) { err = "hipMalloc failed"; return false; }
        } else {
          const int cp = c8i_stride(co);
          auto f = build_frag(1, ci, 4 * cp, [&](int col, int k, int) {
            const int q = col / cp, ch = c8i_logical(col % cp);
            return ch < co ? w[((size_t)k * co + ch) * 4 + q] : 0.f;
          });
          if (!upload("frag:" + op.w, f)) { err = "hipMalloc failed"; return false; }
          if (half_ && !upload("frag16:" + op.w, frag_to_half(f))) { err = "hipMalloc failed"; return false; }
          // [k][q][c] logical channels for the fused DB head (db_head_kernel)
          std::vector<float> img((size_t)ci * 4 * co);
          for (int k = 0; k < ci; ++k)
            for (int q = 0; q < 4; ++q)
              for (int c = 0; c < co; ++c) img[((size_t)k * 4 + q) * co + c] = w[((size_t)k * co + c) * 4 + q];
          if (!upload("dbh1:" + op.w, img)) { err = "hipMalloc failed"; return false; }
          // the same stage as a fragment image whose columns are in the fused head's order (db_head_mfma_kernel): lane
          // (p, h) of column tile t owns value l = 16t + 4g + i of quadrants 2h, 2h+1 = column 32t + 8g + 4h + i
          if (ci == 24 && co == 24) {
            auto hf = build_frag(1, ci, 4 * cp, [&](int col, int k, int) {
              const int t = col / 32, pp = col % 32, l = t * 16 + (pp / 8) * 4 + pp % 4;
              const int q = 2 * ((pp % 8) / 4) + l / cp, ch = c8i_logical(l % cp);
              return ch < co ? w[((size_t)k * co + ch) * 4 + q] : 0.f;
            });
            if (hf.size() != (size_t)3 * 3 * 256) { err = "DB head fragment image: unexpected tiling"; return false; }
            if (!upload("dbhf:" + op.w, hf)) { err = "hipMalloc failed"; return false; }
          }
        }
      } break;
      case PlanOp::DW: {
        auto p = need(op.w); if (!p) return false;
        const int C = op.c, K = op.kh;
        if (op.kh != op.kw || (K != 3 && K != 5)) { err = "depthwise kernel size not on this path"; return false; }
        if (p->dims.size() != 4 || p->dims[0] != C || p->dims[2] != K) { err = "dw filter shape mismatch " + op.w; return false; }
        const int cs = c8i_stride(C);
        std::vector<float> img((size_t)K * K * cs, 0.f);
        for (int c = 0; c < C; ++c)
          for (int t = 0; t < K * K; ++t) img[(size_t)t * cs + c8i_phys(c)] = p->data[(size_t)c * K * K + t];
        if (!upload("dw:" + op.w, img)) { err = "hipMalloc failed"; return false; }
        // the LDS-DMA form of the fused depthwise blocks (dwpw2_kernel.h) fetches a chunk's taps and its folded bias in one
        // burst: [chunk][K*K taps | bias][CK], physical channel order, for the chunk widths its instances use.  Only where the
        // depthwise conv ends with the hard-swish product (an absorbed chain, fold_lab): those are the ones that can be fused
        if (op.ep.size() == 2 && op.ep[0].kind == EP_BIAS && op.ep[1].kind == EP_ACT && op.ep[1].act == ACT_HSW6) {
          auto b = need(op.ep[0].n0); if (!b) return false;
          const std::vector<float> bp = perm_vec(b->data, C, false);
          for (int ck : {16, 32}) {
            if (cs % ck) continue;
            std::vector<float> qimg((size_t)(cs / ck) * (K * K + 1) * ck, 0.f);
            for (int ch = 0; ch < cs / ck; ++ch)
              for (int t = 0; t <= K * K; ++t)
                for (int i = 0; i < ck; ++i)
                  qimg[((size_t)ch * (K * K + 1) + t) * ck + i] = t < K * K ? img[(size_t)t * cs + ch * ck + i] : bp[ch * ck + i];
            if (!upload((ck == 16 ? "dwq16:" : "dwq32:") + op.w, qimg)) { err = "hipMalloc failed"; return false; }
          }
        }
      } break;
      case PlanOp::SEFC: {
        {
          auto w1 = need(op.w1), b1 = need(op.b1), w2 = need(op.w2), b2 = need(op.b2);
          if (!w1 || !b1 || !w2 || !b2) return false;
          if ((long)w1->numel() != (long)op.cr * op.c || (int)b1->numel() != op.cr || (long)w2->numel() != (long)op.c * op.cr ||
              (int)b2->numel() != op.c) { err = "squeeze-excite weight size mismatch " + op.w1; return false; }
        }
        for (const std::string* n : {&op.w1, &op.b1, &op.w2, &op.b2}) {
          auto p = need(*n); if (!p) return false;
          if (!upload("raw:" + *n, p->data)) { err = "hipMalloc failed"; return false; }
        }
      } break;
      case PlanOp::LN: {
        for (const std::string* n : {&op.g, &op.b}) {
          auto p = need(*n); if (!p) return false;
          if ((int)p->numel() != op.c) { err = "layer-norm vector size mismatch " + *n; return false; }
          if (!upload("raw:" + *n, p->data)) { err = "hipMalloc failed"; return false; }
        }
      } break;
      case PlanOp::ATTN:
        if (op.hd != 15) { err = "attention head dim not on this path"; return false; }
        break;
      default: break;
    }
  }
  if (out_tid_ < 0) { err = "plan has no output"; return false; }
  if (!upload("zeros", std::vector<float>(4096, 0.f))) { err = "hipMalloc failed"; return false; }
  return true;
}

bool Net::build_epilogue(const PlanOp& op, Epilogue& ep, bool conv_path, std::string& err) {
  ep.n = 0;
  if ((int)op.ep.size() > OCR_MAX_EP) { err = "epilogue too long"; return false; }
  const bool oplain = tensors_[op.out].plain;
  for (auto& st : op.ep) {
    EpStage& e = ep.st[ep.n++];
    memset(&e, 0, sizeof(e));
    e.kind = st.kind;
    e.act = st.act;
    e.p0 = st.p0;
    e.p1 = st.p1;
    switch (st.kind) {
      case EP_BIAS: e.v0 = dev_vec((oplain ? "vecp:" : "vec:") + st.n0); break;
      case EP_SMUL: case EP_SADD: e.p0 = scalars_[st.n0]; break;
      case EP_SFMA: break;  // (p0, p1 are the folded constants)
      case EP_BN: e.v0 = dev_vec("bns:" + st.n0); e.v1 = dev_vec("bnt:" + st.n0); break;
      case EP_ACT: break;
      case EP_MULC: case EP_GATERES:
        if (tensors_[st.tid].f16) { err = "precision fp16: a gate operand must be a per-image f32 vector"; return false; }
        e.v0 = tensor_ptr(st.tid);
        break;
      case EP_ADDT:
        if (tensors_[st.tid].f16 != tensors_[op.out].f16) { err = "precision fp16: a residual operand of another storage type"; return false; }
        e.v0 = tensor_ptr(st.tid);
        break;
      case EP_ADDUP:
        if (tensors_[st.tid].f16 != tensors_[op.out].f16) { err = "precision fp16: an upsampled operand of another storage type"; return false; }
        e.v0 = tensor_ptr(st.tid);
        e.a0 = st.up;
        e.a1 = tensors_[st.tid].w;
        e.a2 = tensors_[st.tid].h;
        break;
    }
  }
  return true;
}

// ------------------------------------------------------------------ shape binding
std::vector<int> Net::shape_key(int N, int H, int W, const int* widths, const int* heights) {
  if (!widths) return {N, H, W};
  std::vector<int> k = {heights ? -2 : -1, H};  // ragged: run lengths of the samples' sizes, in sample order
  for (int i = 0; i < N;) {
    int j = i;
    while (j < N && widths[j] == widths[i] && (!heights || heights[j] == heights[i])) ++j;
    k.push_back(widths[i]);
    if (heights) k.push_back(heights[i]);
    k.push_back(j - i);
    i = j;
  }
  return k;
}

bool Net::ragged_ok(int H, const int* widths, int N, std::string& why) {
  long pix = 0;
  int wmax = 0;
  for (int i = 0; i < N; ++i) {
    if (widths[i] < 1) { why = "line width < 1"; return false; }
    pix += (long)H * widths[i];
    wmax = std::max(wmax, widths[i]);
  }
  if (N < 1 || pix >= (1L << 30)) { why = "too many pixels for one ragged launch"; return false; }
  if (!attn_ragged_fits(wmax / 8 + 2)) { why = "line too wide for the ragged attention kernel"; return false; }
  return true;
}

const std::vector<int>& Net::ragged_widths(int tid) const {
  static const std::vector<int> none;
  if (!cur_ || tid < 0 || tid >= (int)tensors_.size() || tensors_[tid].lvl < 0) return none;
  return cur_->level_w[tensors_[tid].lvl];
}

bool Net::bind(int N, int H, int W, std::string& err, const int* widths, const int* heights) {
  stats_.binds++;
  std::unique_ptr<Binding> B(new Binding());
  const bool rag = widths != nullptr;
  const bool img = rag && heights != nullptr;  // ragged batch of images (own height and width) instead of lines
  B->n = N; B->h = H; B->w = W;
  B->pool = pool_;
  std::vector<Launch>& launches_ = B->launches;
  bool moved = false;  // a shared device buffer was reallocated: the other bindings' launches point into the old one
  fused_head_rows_ = -1;
  // 1. shapes
  auto& T = tensors_;
  auto setdims = [&](int t, int n, int h, int w, int c) {
    T[t].n = n; T[t].h = h; T[t].w = w; T[t].c = c;
    T[t].cs = T[t].plain ? c : c8i_stride(c);
    T[t].lvl = -1; T[t].pix = 0;
  };
  // ragged batch: width levels.  A level = the lines' widths after some chain of (kernel, stride, pad) along x;
  // ops that keep the width stay on their input's level.
  std::vector<std::vector<int>>& LW = B->level_w;
  std::vector<long> ltot;
  std::vector<int> lmax, lmin;
  auto add_level = [&](std::vector<int> w) -> int {
    for (size_t l = 0; l < LW.size(); ++l) if (LW[l] == w) return (int)l;
    long tot = 0;
    int mx = 0, mn = 0x7fffffff;
    for (int v : w) { tot += v; mx = std::max(mx, v); mn = std::min(mn, v); }
    LW.push_back(std::move(w));
    ltot.push_back(tot); lmax.push_back(mx); lmin.push_back(mn);
    return (int)LW.size() - 1;
  };
  auto derive = [&](int lvl, int k, int s, int p) -> int {  // < 0: a line got too narrow
    if (k == 1 && s == 1 && p == 0) return lvl;
    std::vector<int> w(LW[lvl].size());
    for (size_t i = 0; i < w.size(); ++i) {
      w[i] = (LW[lvl][i] + 2 * p - k) / s + 1;  // C++ truncation, as the uniform shapes below
      if (LW[lvl][i] + 2 * p - k < 0 || w[i] <= 0) return -1;
    }
    return add_level(std::move(w));
  };
  auto setrag = [&](int t, int h, int lvl, int c) {
    setdims(t, N, h, lmax[lvl], c);
    T[t].lvl = lvl;
    T[t].pix = (long)h * ltot[lvl];
  };
  // images: ONE table set at input resolution; a tensor's level is its power-of-two shift against the input
  std::vector<int>& IH = B->heights;
  long ipix = 0, irows = 0;   // pixels / rows of the batch at input resolution
  int ihmax = 0, iwmax = 0, ihmin = 0x7fffffff, iwmin = 0x7fffffff;
  auto setimg = [&](int t, int shift, int c) {
    setdims(t, N, ihmax >> shift, iwmax >> shift, c);
    T[t].lvl = shift;
    T[t].pix = ipix >> (2 * shift);
  };
  if (img) {
    B->widths.assign(widths, widths + N);
    IH.assign(heights, heights + N);
    for (int i = 0; i < N; ++i) {
      if (widths[i] < 32 || heights[i] < 32 || widths[i] % 32 || heights[i] % 32 || widths[i] > 32767 || heights[i] > 32767) {
        err = "ragged batch of images: sizes must be multiples of 32 (ResizeImgType0 output)";
        return false;
      }
      ipix += (long)widths[i] * heights[i];
      irows += heights[i];
      ihmax = std::max(ihmax, heights[i]); iwmax = std::max(iwmax, widths[i]);
      ihmin = std::min(ihmin, heights[i]); iwmin = std::min(iwmin, widths[i]);
    }
    if (ipix >= (1L << 31)) { err = "too many pixels for one ragged launch"; return false; }
    setimg(0, 0, 3);
  } else if (rag) {
    B->widths.assign(widths, widths + N);
    setrag(0, H, add_level(B->widths), 3);
  } else setdims(0, N, H, W, 3);
  for (auto& op : plan_.ops) {
    if (img && op.kind != PlanOp::OUTPUT) {
      const TensorDesc& i = T[op.kind == PlanOp::CONCAT ? op.ins.back() : op.in];
      // every spatial op of the detector keeps the size or halves it exactly (inputs are multiples of 32)
      auto step = [&](int kh, int kw, int sh, int sw, int ph, int pw) -> int {
        if (kh != kw || sh != sw || ph != pw || kh != 2 * ph + 1 || (sh != 1 && sh != 2)) return -1;
        return sh == 2 ? 1 : 0;
      };
      switch (op.kind) {
        case PlanOp::CONV: case PlanOp::DW: {
          const int d = step(op.kh, op.kw, op.sh, op.sw, op.ph, op.pw);
          if (i.lvl < 0 || d < 0 || i.lvl + d > 5) { err = "ragged batch of images: conv geometry not on this path"; return false; }
          setimg(op.out, i.lvl + d, op.kind == PlanOp::CONV ? op.cout : op.c);
        } break;
        case PlanOp::DECONV:
          if (i.lvl < 1) { err = "ragged batch of images: transposed conv above the input resolution"; return false; }
          setimg(op.out, i.lvl - 1, op.cout);
          break;
        case PlanOp::LINEAR: setimg(op.out, i.lvl, op.cout); break;
        case PlanOp::EW: if (i.lvl < 0) setdims(op.out, i.n, i.h, i.w, i.c); else setimg(op.out, i.lvl, i.c); break;
        case PlanOp::SEFC: case PlanOp::GAP: setdims(op.out, N, 1, 1, op.c); break;
        case PlanOp::CONCAT:
          for (size_t j = 0; j < op.ins.size(); ++j) {
            int lu = 0;
            while ((1 << lu) < op.ups[j]) ++lu;
            if ((1 << lu) != op.ups[j] || T[op.ins[j]].lvl != i.lvl + lu) { err = "ragged batch of images: concat sources must be power-of-two coarser levels"; return false; }
          }
          setimg(op.out, i.lvl, op.c);
          break;
        default: err = "ragged batch of images: op kind not on this path"; return false;
      }
      continue;
    }
    if (rag && op.kind != PlanOp::OUTPUT) {
      const TensorDesc& i = T[op.kind == PlanOp::CONCAT ? op.ins.back() : op.in];
      int lvl = i.lvl, oh = i.h, oc = i.c;
      bool per_line = false;  // the op's output is one vector per line
      switch (op.kind) {
        case PlanOp::CONV: case PlanOp::DW:
          if (lvl < 0) { err = "ragged batch: spatial op on a per-line vector"; return false; }
          oh = (i.h + 2 * op.ph - op.kh) / op.sh + 1;
          lvl = derive(lvl, op.kw, op.sw, op.pw);
          oc = op.kind == PlanOp::CONV ? op.cout : op.c;
          break;
        case PlanOp::POOL:
          if (lvl < 0) { err = "ragged batch: spatial op on a per-line vector"; return false; }
          oh = (i.h - op.kh) / op.sh + 1;
          lvl = derive(lvl, op.kw, op.sw, 0);
          oc = op.c;
          break;
        case PlanOp::LINEAR: oc = op.cout; break;
        case PlanOp::SEFC: case PlanOp::GAP: per_line = true; oc = op.c; break;
        case PlanOp::EW: case PlanOp::LN: case PlanOp::SOFTMAX: break;
        case PlanOp::CONCAT:
          for (size_t j = 0; j < op.ins.size(); ++j)
            if (op.ups[j] != 1 || T[op.ins[j]].lvl != lvl || T[op.ins[j]].h != oh) { err = "ragged batch: concat of different resolutions is not on this path"; return false; }
          oc = op.c;
          break;
        case PlanOp::ATTN: oc = op.heads * op.hd; break;
        default: err = "ragged batch: op kind not on this path (transposed conv)"; return false;
      }
      if (per_line) { setdims(op.out, N, 1, 1, oc); continue; }
      if (lvl < 0 && T[op.in].lvl >= 0) { err = "input too small for the network"; return false; }
      if (lvl < 0) { setdims(op.out, i.n, i.h, i.w, oc); continue; }  // per-line vector -> per-line vector (sefc chain)
      if (oh <= 0) { err = "input too small for the network"; return false; }
      setrag(op.out, oh, lvl, oc);
      continue;
    }
    switch (op.kind) {
      case PlanOp::CONV: {
        auto& i = T[op.in];
        setdims(op.out, i.n, (i.h + 2 * op.ph - op.kh) / op.sh + 1, (i.w + 2 * op.pw - op.kw) / op.sw + 1, op.cout);
      } break;
      case PlanOp::DW: {
        auto& i = T[op.in];
        setdims(op.out, i.n, (i.h + 2 * op.ph - op.kh) / op.sh + 1, (i.w + 2 * op.pw - op.kw) / op.sw + 1, op.c);
      } break;
      case PlanOp::DECONV: { auto& i = T[op.in]; setdims(op.out, i.n, i.h * 2, i.w * 2, op.cout); } break;
      case PlanOp::LINEAR: { auto& i = T[op.in]; setdims(op.out, i.n, i.h, i.w, op.cout); } break;
      case PlanOp::SEFC: case PlanOp::GAP: { auto& i = T[op.in]; setdims(op.out, i.n, 1, 1, op.c); } break;
      case PlanOp::POOL: {
        auto& i = T[op.in];
        setdims(op.out, i.n, (i.h - op.kh) / op.sh + 1, (i.w - op.kw) / op.sw + 1, op.c);  // C++ truncation on purpose
      } break;
      case PlanOp::EW: case PlanOp::LN: case PlanOp::SOFTMAX: { auto& i = T[op.in]; setdims(op.out, i.n, i.h, i.w, i.c); } break;
      case PlanOp::CONCAT: {
        auto& l = T[op.ins.back()];
        setdims(op.out, l.n, l.h * op.ups.back(), l.w * op.ups.back(), op.c);
      } break;
      case PlanOp::ATTN: { auto& i = T[op.in]; setdims(op.out, i.n, i.h, i.w, op.heads * op.hd); } break;
      case PlanOp::OUTPUT: break;
    }
    if (op.out >= 0 && (T[op.out].h <= 0 || T[op.out].w <= 0)) { err = "input too small for the network"; return false; }
  }
  // ragged batch: the sample tables.  Lines: per width level  w[N+1] | cw[N+1].  Images: one set at input resolution
  // w[N+1] | h[N+1] | cw[N+1] | ch[N+1] (w / h carry one entry past N: the tile walks step onto "sample N" after their
  // last unit).  Behind them the launches' own work tables (prefix sums of tiles / patches / bands per sample, N+1
  // entries each), appended while the launch list is built: the device buffer is sized for all of them up front.
  const size_t lstride = img ? 4 * ((size_t)N + 1) : 2 * ((size_t)N + 1);
  const size_t tab_base = rag ? lstride * (img ? 1 : LW.size()) : 0;
  const size_t tab_cap = rag ? tab_base + 96 * ((size_t)N + 1) : 0;
  if (rag) {
    B->rag_host.assign(tab_base, 0);
    if (img) {
      int* w = B->rag_host.data();
      int *h = w + (N + 1), *cw = h + (N + 1), *ch = cw + (N + 1);
      for (int i = 0; i < N; ++i) {
        w[i] = widths[i]; h[i] = heights[i];
        cw[i + 1] = cw[i] + widths[i] * heights[i];
        ch[i + 1] = ch[i] + heights[i];
      }
      w[N] = w[N - 1]; h[N] = h[N - 1];
    } else
    for (size_t l = 0; l < LW.size(); ++l) {
      int* w = B->rag_host.data() + l * lstride;
      int* cw = w + (N + 1);
      for (int i = 0; i < N; ++i) { w[i] = LW[l][i]; cw[i + 1] = cw[i] + w[i]; }
      w[N] = w[N - 1];
    }
    B->rag_host.reserve(tab_cap);
    for (size_t i = 0; i < pool_->free_list.size(); ++i)
      if (pool_->free_list[i].second >= tab_cap) {
        B->rag_dev = pool_->free_list[i].first;
        B->rag_cap = pool_->free_list[i].second;
        pool_->free_list.erase(pool_->free_list.begin() + i);
        break;
      }
    if (!B->rag_dev) {
      const size_t cap = tab_cap + tab_cap / 2;
      HIP_OK(g_malloc(&B->rag_dev, cap * sizeof(int)));
      B->rag_cap = cap;
    }
  }
  const int* rag_dev = B->rag_dev;
  auto rlevel = [&](const TensorDesc& t) {
    RagLevel r;
    if (rag && t.lvl >= 0) {
      if (img) { r.w = rag_dev; r.h = r.w + (N + 1); r.cw = r.h + (N + 1); r.ch = r.cw + (N + 1); r.shift = t.lvl; }
      else { r.w = rag_dev + (size_t)t.lvl * lstride; r.cw = r.w + (N + 1); }
    }
    return r;
  };
  // size of sample i on a tensor's level
  auto sw_of = [&](const TensorDesc& t, int i) { return img ? widths[i] >> t.lvl : LW[t.lvl][i]; };
  auto sh_of = [&](const TensorDesc& t, int i) { return img ? heights[i] >> t.lvl : t.h; };
  auto min_w = [&](const TensorDesc& t) { return img ? iwmin >> t.lvl : lmin[t.lvl]; };
  auto min_h = [&](const TensorDesc& t) { return img ? ihmin >> t.lvl : t.h; };
  auto rows_of = [&](const TensorDesc& t) -> long { return img ? irows >> t.lvl : (long)N * t.h; };
  // a launch's work table: prefix sums of count(sample), memoised by key; returns its device address and total
  std::map<std::string, std::pair<size_t, int>> tabs;
  bool tab_overflow = false;
  auto work_table = [&](const std::string& key, const std::function<int(int)>& count, int& total) -> const int* {
    auto it = tabs.find(key);
    if (it == tabs.end()) {
      if (B->rag_host.size() + N + 1 > tab_cap) { tab_overflow = true; total = 0; return rag_dev; }
      const size_t off = B->rag_host.size();
      B->rag_host.resize(off + N + 1, 0);
      int acc = 0;
      for (int i = 0; i < N; ++i) { acc += count(i); B->rag_host[off + i + 1] = acc; }
      it = tabs.emplace(key, std::make_pair(off, acc)).first;
    }
    total = it->second.second;
    return rag_dev + it->second.first;
  };
  auto tiles_table = [&](const TensorDesc& t, int th, int& total) {  // th x 16 pixel tiles per sample
    return work_table("tiles:" + std::to_string(t.lvl) + ":" + std::to_string(t.h) + ":" + std::to_string(th),  // (lines: tensors of one width level differ in height)
                      [&](int i) { return ((sh_of(t, i) + th - 1) / th) * ((sw_of(t, i) + 15) / 16); }, total);
  };
  // 2. arena with liveness reuse
  const int nops = (int)plan_.ops.size();
  // SE gate folding: `ew x -> x * gate[n][c]` whose only reader is a 1x1 conv disappears - the conv reads x and
  // the gate and forms the same product (one rounding) on its way into the matrix pipe.  Saves a full read +
  // write pass per SE block (cls: 9, rec: 2, det backbone: 2).  Not in keep-all mode (every plan tensor must
  // exist for the parity taps) and not with OCR_FUSE=0 (A/B; results are identical).
  std::vector<int> gate_src(nops, -1), gate_tid(nops, -1);
  std::vector<char> folded(nops, 0);
  std::vector<int> uses(plan_.ntensors, 0);
  for (auto& op : plan_.ops) {
    if (op.in >= 0) uses[op.in]++;
    for (int t : op.ins) uses[t]++;
    for (auto& st : op.ep) if (st.tid >= 0) uses[st.tid]++;
  }
  const bool fuse = keep_all_ != 1 && rt_options().fuse;  // (OCR_FUSE=0: one launch per plan op, as the parity taps' keep_all = 1)
  if (fuse) {
    for (int oi = 0; oi < nops; ++oi) {
      auto& op = plan_.ops[oi];
      if (op.kind != PlanOp::EW || op.ep.size() != 1 || op.ep[0].kind != EP_MULC || op.out == out_tid_ || uses[op.out] != 1) continue;
      for (int oj = oi + 1; oj < nops; ++oj) {
        auto& c = plan_.ops[oj];
        if (c.in != op.out) continue;
        const bool one = c.kind == PlanOp::CONV && c.kh == 1 && c.kw == 1 && c.sh == 1 && c.sw == 1 && c.ph == 0 && c.pw == 0 && c.cin != 3;
        if (one && !T[c.out].plain && T[op.in].pixels() < 0x7fffffffL) {
          folded[oi] = 1;
          gate_src[oj] = op.in;
          gate_tid[oj] = op.ep[0].tid;
        }
        break;
      }
    }
  }
  // Expand 1x1 conv -> depthwise 5x5 on small maps (kernels_xdw.hip; the classifier's inverted-residual blocks): the conv
  // runs inside the depthwise op's launch, chunk by chunk into LDS; its output tensor never exists.  Uniform batches, f32.
  std::vector<int> xdw_of(nops, -1);  // depthwise op -> the 1x1 conv it absorbs
  if (fuse && !rag && !half_ && rt_options().xdw) {
    auto bn_hswish = [](const std::vector<PlanStage>& ep) {
      return ep.size() == 2 && ep[0].kind == EP_BN && ep[1].kind == EP_ACT && ep[1].act == ACT_HSWISH;
    };
    for (int oi = 0; oi + 1 < nops; ++oi) {
      const PlanOp& c = plan_.ops[oi];
      const bool one = c.kind == PlanOp::CONV && c.kh == 1 && c.kw == 1 && c.sh == 1 && c.sw == 1 && c.ph == 0 && c.pw == 0 && c.cin != 3;
      if (!one || gate_src[oi] >= 0 || c.out == out_tid_ || uses[c.out] != 1 || T[c.in].plain || T[c.out].plain || !bn_hswish(c.ep)) continue;
      for (int oj = oi + 1; oj < nops; ++oj) {
        const PlanOp& d = plan_.ops[oj];
        if (d.kind == PlanOp::CONCAT || d.in != c.out) continue;
        if (d.kind == PlanOp::DW && d.kh == d.kw && bn_hswish(d.ep) && !T[d.out].plain && dev_vec("frag:" + c.w)) {
          XdwArgs q{};
          q.N = T[c.in].n; q.Hin = T[c.in].h; q.Hout = T[d.out].h; q.W = T[c.in].w; q.Cs_in = T[c.in].cs; q.Cs_e = T[d.out].cs;
          const int tiles = (T[c.out].cs + 31) / 32, nt = conv_nt_for(tiles);
          q.NTtot = (tiles + nt - 1) / nt * nt;
          q.K = d.kh; q.SH = d.sh; q.SW = d.sw; q.PH = d.ph; q.PW = d.pw;
          if (T[d.out].w == T[c.in].w && T[c.out].cs == T[d.out].cs && launch_xdw(q, nullptr, true)) xdw_of[oj] = oi;
        }
        break;
      }
    }
  }
  // Depthwise -> pointwise fusion (kernels_dwpw.hip): a depthwise conv whose only reader is an ungated 1x1 conv runs
  // inside that conv's launch; its output tensor never exists.
  std::vector<int> dwpw_of(nops, -1);  // conv op -> the depthwise op it absorbs
  if (fuse) {
    for (int oi = 0; oi + 1 < nops; ++oi) {
      auto& d = plan_.ops[oi];
      if (d.kind != PlanOp::DW || d.out == out_tid_ || uses[d.out] != 1) continue;
      // both epilogues must be the folded LAB chains the kernel compiles in (fold_lab): bias | hsw6 in the depthwise half
      // (its scale / shift absorbed by the conv), bias | hsw6 | sfma behind the conv
      auto is_lab = [](const std::vector<PlanStage>& ep, bool sfma) {
        return ep.size() == (sfma ? 3u : 2u) && ep[0].kind == EP_BIAS && ep[1].kind == EP_ACT && ep[1].act == ACT_HSW6 &&
               (!sfma || ep[2].kind == EP_SFMA);
      };
      if (!is_lab(d.ep, false)) continue;
      for (int oj = oi + 1; oj < nops; ++oj) {
        auto& c = plan_.ops[oj];
        if (c.in != d.out) continue;
        const bool one = c.kind == PlanOp::CONV && c.kh == 1 && c.kw == 1 && c.sh == 1 && c.sw == 1 && c.ph == 0 && c.pw == 0 && c.cin != 3;
        if (one && gate_src[oj] < 0 && !T[c.out].plain && is_lab(c.ep, true) && T[d.in].cs % 8 == 0 && !T[d.in].plain && c.cout % 8 == 0) {
          // shape on the fused path?  (asks the launcher, which also raises the kernel's LDS limit on this device)
          DwPwArgs q{};
          q.K = d.kh; q.SH = d.sh; q.SW = d.sw;
          q.c.Cs_in = T[d.out].cs;
          const int tiles = (T[c.out].cs + 31) / 32, nt = conv_nt_for(tiles);
          q.c.NTtot = (tiles + nt - 1) / nt * nt;
          q.dw_ep.sfma = 0; q.pw_ep.sfma = 1;
          q.dw_wq16 = dev_vec("dwq16:" + d.w); q.dw_wq32 = dev_vec("dwq32:" + d.w);  // (the LDS-DMA form's instance, where there is one: its own LDS attribute)
          q.c.half = half_ ? 1 : 0;  // (the f16 build's instance: its own LDS attribute)
          if (rag) q.rtiles = rag_dev;  // (the ragged instantiation is its own kernel: own LDS attribute)
          if (d.kh == d.kw && launch_dwpw(q, nullptr, true)) dwpw_of[oj] = oi;
        }
        break;
      }
    }
  }
  std::vector<char> fused_dw(nops, 0);
  for (int oj = 0; oj < nops; ++oj) if (dwpw_of[oj] >= 0) fused_dw[dwpw_of[oj]] = 1;
  for (int oj = 0; oj < nops; ++oj) if (xdw_of[oj] >= 0) fused_dw[xdw_of[oj]] = 1;  // (the absorbed 1x1 conv: no launch, no tensor)
  // ---- depthwise conv -> global average pool (the SE blocks): the conv leaves the pool's row sums (its first pass, a
  // second full read of the tensor otherwise) while it writes the tensor; only the column pass stays a launch.  Needs
  // enough bands (a thread owns whole rows then) to fill the chip.
  std::vector<char> dw_rowsum(nops, 0);
  if (fuse) {
    for (int oi = 1; oi < nops; ++oi) {
      const PlanOp& g = plan_.ops[oi];
      const PlanOp& d = plan_.ops[oi - 1];
      if (g.kind != PlanOp::GAP || d.kind != PlanOp::DW || d.out != g.in || fused_dw[oi - 1] || d.out == out_tid_) continue;
      if (xdw_of[oi - 1] >= 0) { dw_rowsum[oi - 1] = 1; continue; }  // (that kernel leaves the row sums whatever the size and stride)
      const TensorDesc& o = T[d.out];
      const int rows_per_band = (rag ? min_h(o) : o.h) >= 2 ? 2 : 1;
      const long threads = (rag ? rows_of(o) / rows_per_band : (long)o.n * ((o.h + rows_per_band - 1) / rows_per_band)) * (o.cs >> 2);
      // (stride-2 rows measured slower this way: rec op 21 0.55 -> 0.72 ms for a 0.18 ms pool pass)
      const long min_threads = rt_options().fuse_gap_min;  // OCR_FUSE_GAP_MIN (tests: 1 sends small shapes down this path too)
      if (threads >= min_threads && d.sh == 1 && d.sw == 1) dw_rowsum[oi - 1] = 1;
    }
  }
  // ---- DB head: deconv (C -> C, bias + BN + relu) -> deconv (C -> 1, bias + sigmoid) as one kernel; the C-channel map
  // between them (1.4 GB at configs[1]) is never written.
  std::vector<int> dbhead_of(nops, -1);  // tail op -> the deconv it absorbs
  if (fuse) {
    for (int oi = 0; oi + 1 < nops; ++oi) {
      const PlanOp& d = plan_.ops[oi];
      const PlanOp& t = plan_.ops[oi + 1];
      if (d.kind != PlanOp::DECONV || d.cout == 1 || t.kind != PlanOp::DECONV || t.cout != 1 || t.in != d.out) continue;
      if (uses[d.out] != 1 || d.out == out_tid_ || d.cin != 24 || d.cout != 24 || T[d.in].cs != 24) continue;
      if (d.ep.size() != 3 || d.ep[0].kind != EP_BIAS || d.ep[1].kind != EP_BN || d.ep[2].kind != EP_ACT || d.ep[2].act != ACT_RELU) continue;
      dbhead_of[oi + 1] = oi;
      fused_dw[oi] = 1;  // (same bookkeeping as a fused depthwise conv: no launch, no tensor)
    }
  }
  // ---- the DB neck's concat (four 24-channel maps, upsampled x1 / x2 / x4 / x8) -> conv 3x3 96 -> 24: the conv's LDS-tile
  // fill reads the four sources itself (kernels_net.hip, ConvArgs::cat_*); the 96-channel tensor (1.4 GB at configs[1]) is never
  // written.  Only where the conv runs on a kernel that has the folded fill: the 4x4x1 form (f32) or the f16-staged form
  // (precision "fp16", uniform batches).
  std::vector<int> cat_of(nops, -1);  // conv op -> the concat it absorbs
  // The recognizer's neck has the same shape without upsampling: concat (two 480-channel sequences) -> conv 1x3 960 -> 60; the
  // LDS-staged conv (f32) reads chunk c of its input from source c * BK / 480 (conv_lds_kernel, load_a).
  if (fuse && !half_) {
    for (int oi = 0; oi + 1 < nops; ++oi) {
      const PlanOp& k = plan_.ops[oi];
      if (k.kind != PlanOp::CONCAT || k.out == out_tid_ || uses[k.out] != 1 || k.ins.size() < 2 || k.ins.size() > 4) continue;
      bool same = true;
      for (size_t j = 0; j < k.ins.size(); ++j)
        same = same && k.ups[j] == 1 && T[k.ins[j]].cs == T[k.ins[0]].cs && !T[k.ins[j]].plain && T[k.ins[j]].lvl == T[k.out].lvl;
      if (!same || T[k.ins[0]].cs % 32 || T[k.ins[0]].cs * (int)k.ins.size() != T[k.out].cs) continue;
      for (int oj = oi + 1; oj < nops; ++oj) {
        const PlanOp& c = plan_.ops[oj];
        if (c.in != k.out) continue;
        const bool lds_conv = c.kind == PlanOp::CONV && c.kh * c.kw > 1 && c.sh == 1 && c.sw == 1 && !(c.kh == 3 && c.kw == 3 && c.cin == 96);
        if (lds_conv && !T[c.out].plain && gate_src[oj] < 0 && dwpw_of[oj] < 0) {
          cat_of[oj] = oi;
          folded[oi] = 1;
        }
        break;
      }
    }
  }
  if (fuse && (!rag || img)) {
    for (int oi = 0; oi + 1 < nops; ++oi) {
      const PlanOp& k = plan_.ops[oi];
      if (k.kind != PlanOp::CONCAT || folded[oi] || k.out == out_tid_ || uses[k.out] != 1 || k.ins.size() != 4 || T[k.out].cs != 96) continue;
      if (T[k.out].numel() >= (1ul << 31)) continue;  // (the folded fill indexes its sources with 32 bits)
      bool same = true;
      for (int t : k.ins) same = same && T[t].cs * (int)k.ins.size() == 96 && T[t].f16 == T[k.out].f16 && !T[t].plain;
      if (!same) continue;
      for (int oj = oi + 1; oj < nops; ++oj) {
        const PlanOp& c = plan_.ops[oj];
        if (c.in != k.out) continue;
        const bool conv33 = c.kind == PlanOp::CONV && c.kh == 3 && c.kw == 3 && c.sh == 1 && c.sw == 1 && c.ph == 1 && c.pw == 1 && c.cin == 96 && c.cout == 24;
        bool ep_ok = true;
        for (auto& st : c.ep) ep_ok = ep_ok && st.kind != EP_ADDUP && !(img && st.kind == EP_MULC);
        bool kernel_ok = half_ ? (rt_options().mfma_x16 && dev_vec("frag16x:" + c.w))
                               : (rt_options().conv_c24 && dev_vec("c24:" + c.w) != nullptr);
        if (conv33 && ep_ok && kernel_ok) {
          // the folded fill exists in ONE kernel per precision: ask its launcher now, on the device that will run it (shape
          // checks and the dynamic-LDS attribute) - a refusal at launch time would leave the conv without an input tensor.
          // Refused here, the concat is materialised and the conv keeps its fallback chain (conv3x3_tile, conv_lds).
          ConvArgs q{};
          q.KH = q.KW = 3; q.PH = q.PW = 1; q.H = q.OH = T[c.out].h; q.W = q.OW = T[c.out].w; q.N = T[c.out].n;
          q.Cs_in = 96; q.Cs_out = T[c.out].cs; q.Cout = c.cout; q.out_mode = OUT_C8I; q.NTtot = 1;
          q.cat_n = 4; q.cat_cs = 24; q.half = half_ ? 1 : 0;
          if (rag) { q.rin.w = q.rout.w = rag_dev; q.rin.h = q.rout.h = rag_dev; q.rtiles = rag_dev; q.rtiles_total = 1; }
          Epilogue qe{};
          qe.n = (int)std::min<size_t>(c.ep.size(), OCR_MAX_EP);
          for (int k = 0; k < qe.n; ++k) qe.st[k].kind = c.ep[k].kind;
          if (half_) { q.wfrag_x16 = dev_vec("frag16x:" + c.w); kernel_ok = launch_conv3x3_tile(q, qe, 1, nullptr, true); }
          else kernel_ok = launch_conv3x3_c24(q, qe, dev_vec("c24:" + c.w), nullptr, true);
        }
        if (conv33 && ep_ok && kernel_ok && !T[c.out].plain && gate_src[oj] < 0 && dwpw_of[oj] < 0) {
          cat_of[oj] = oi;
          folded[oi] = 1;  // (same bookkeeping as a folded gate multiply: no launch, no tensor, its reads happen in the conv)
        }
        break;
      }
    }
  }
  // ---- RSE blocks of the detector's neck: conv 1x1 (Cin <= 24, no epilogue) -> gap -> sefc -> ew (x * g + x [+ upsampled]).
  // Materialised, the conv's 96-channel output is written once and read twice (pool, ew) before the result is written
  // again: three passes over 1.4 GB at 240 x 240 x 64 images for a K = 12 matrix product.  Instead the conv runs twice:
  // first as conv_rowsum_kernel, which leaves only the pool's row sums, then - with the gate known - as the ordinary
  // conv kernel with the ew's stages in its epilogue (EP_GATERES + the FPN's addup), writing the block's output.  Both
  // passes compute the same chain per output, so nothing changes bit for bit.  Uniform batches and ragged batches of images
  // (the conv epilogue's per-image stages decode (n, y, x) from either).
  std::vector<int> rse_conv(nops, -1);   // ew op -> the conv it absorbs
  std::vector<char> rse_first(nops, 0);  // that conv: its launch is the row-sum pass, its tensor never exists
  if (fuse && (!rag || img)) {
    for (int oi = 0; oi + 3 < nops; ++oi) {
      const PlanOp& c = plan_.ops[oi];
      const PlanOp& g = plan_.ops[oi + 1];
      const PlanOp& f = plan_.ops[oi + 2];
      const PlanOp& e = plan_.ops[oi + 3];
      const bool one = c.kind == PlanOp::CONV && c.kh == 1 && c.kw == 1 && c.sh == 1 && c.sw == 1 && c.ph == 0 && c.pw == 0 && c.cin != 3;
      if (!one || !c.ep.empty() || c.cin > 24 || T[c.out].plain || T[c.in].plain || gate_src[oi] >= 0 || dwpw_of[oi] >= 0) continue;
      if (g.kind != PlanOp::GAP || g.in != c.out || f.kind != PlanOp::SEFC || f.in != g.out || e.kind != PlanOp::EW || e.in != c.out) continue;
      if (e.ep.size() < 2 || e.ep[0].kind != EP_MULC || e.ep[0].tid != f.out || e.ep[1].kind != EP_ADDT || e.ep[1].tid != c.out) continue;
      bool rest_ok = true;
      for (size_t k = 2; k < e.ep.size(); ++k) rest_ok = rest_ok && (e.ep[k].kind == EP_ADDUP || e.ep[k].kind == EP_ADDT) && e.ep[k].tid != c.out;
      if (!rest_ok || uses[c.out] != 3 || c.out == out_tid_ || (T[c.in].cs != 16 && T[c.in].cs != 24)) continue;
      rse_conv[oi + 3] = oi;
      rse_first[oi] = 1;
    }
  }
  std::vector<int> last(plan_.ntensors, -1);
  for (int oi = 0; oi < nops; ++oi) {
    auto& op = plan_.ops[oi];
    if (folded[oi] || fused_dw[oi]) continue;  // its reads happen in the conv it was folded / fused into
    if (rse_conv[oi] >= 0) {  // the second conv pass: reads the conv's input again, the gate and the upsampled operand; never the conv's (dead) output
      const PlanOp& c = plan_.ops[rse_conv[oi]];
      last[c.in] = oi;
      for (auto& st : op.ep) if (st.tid >= 0 && st.tid != c.out) last[st.tid] = oi;
      continue;
    }
    if (oi > 0 && rse_first[oi - 1]) continue;  // the pool after a row-sum pass reads gap_part_, not the conv's tensor
    if (gate_src[oi] >= 0) { last[gate_src[oi]] = oi; last[gate_tid[oi]] = oi; }
    else if (dwpw_of[oi] >= 0) last[plan_.ops[dwpw_of[oi]].in] = oi;
    else if (xdw_of[oi] >= 0) last[plan_.ops[xdw_of[oi]].in] = oi;
    else if (dbhead_of[oi] >= 0) last[plan_.ops[dbhead_of[oi]].in] = oi;
    else if (cat_of[oi] >= 0) { for (int t : plan_.ops[cat_of[oi]].ins) last[t] = oi; }
    else if (op.in >= 0) last[op.in] = oi;
    for (int t : op.ins) last[t] = oi;
    for (auto& st : op.ep) if (st.tid >= 0) last[st.tid] = oi;
  }
  last[out_tid_] = nops + 1;
  struct Blk { size_t off, sz; };
  std::vector<Blk> freeb;
  size_t top = 0;
  size_t gap_need = 0;
  auto alloc = [&](size_t n) {
    n = (n + 63) & ~(size_t)63;
    int best = -1;
    for (int i = 0; i < (int)freeb.size(); ++i)
      if (freeb[i].sz >= n && (best < 0 || freeb[i].sz < freeb[best].sz)) best = i;
    if (best >= 0) {
      size_t off = freeb[best].off;
      if (freeb[best].sz == n) freeb.erase(freeb.begin() + best);
      else { freeb[best].off += n; freeb[best].sz -= n; }
      return off;
    }
    size_t off = top;
    top += n;
    return off;
  };
  auto release = [&](size_t off, size_t n) {
    n = (n + 63) & ~(size_t)63;
    freeb.push_back({off, n});
    std::sort(freeb.begin(), freeb.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
    for (size_t i = 0; i + 1 < freeb.size();) {
      if (freeb[i].off + freeb[i].sz == freeb[i + 1].off) { freeb[i].sz += freeb[i + 1].sz; freeb.erase(freeb.begin() + i + 1); }
      else ++i;
    }
    if (!freeb.empty() && freeb.back().off + freeb.back().sz == top) { top = freeb.back().off; freeb.pop_back(); }
  };
  for (int oi = 0; oi < nops; ++oi) {
    auto& op = plan_.ops[oi];
    if (folded[oi] || fused_dw[oi]) { T[op.out].offset = 0; continue; }  // never materialised
    if (rse_first[oi]) T[op.out].offset = 0;  // (the row-sum pass writes gap_part_ only)
    else if (op.out >= 0) T[op.out].offset = alloc(T[op.out].numel());
    if (op.kind == PlanOp::GAP) gap_need = std::max(gap_need, (size_t)(rag ? rows_of(T[op.in]) : (long)T[op.in].n * T[op.in].h) * T[op.in].cs);
    // free tensors whose last reader is this op (never the op's own output)
    for (int t = 1; t < plan_.ntensors; ++t)
      if (keep_all_ == 0 && last[t] == oi && t != op.out && T[t].numel()) release(T[t].offset, T[t].numel());
  }
  // `top` may have shrunk at the end; capacity must cover the high-water mark
  size_t high = 0;
  std::vector<char> dead(plan_.ntensors, 0);
  for (int oi = 0; oi < nops; ++oi) if (folded[oi] || fused_dw[oi] || rse_first[oi]) dead[plan_.ops[oi].out] = 1;
  B->exists.assign(plan_.ntensors, 0);
  for (int t = 1; t < plan_.ntensors; ++t) B->exists[t] = !dead[t];
  for (int t = 1; t < plan_.ntensors; ++t) if (!dead[t]) high = std::max(high, T[t].offset + ((T[t].numel() + 63) & ~(size_t)63));
  if (high > arena_cap_) {
    if (arena_) (void)g_free(arena_);
    moved = true;
    arena_ = nullptr;
    arena_cap_ = 0;
    HIP_OK(g_malloc(&arena_, (high + 64) * sizeof(float)));  // + 256 B: conv1x1 block loads may run past the last row
    arena_cap_ = high;
  }
  if (gap_need > gap_part_cap_) {
    if (gap_part_) (void)g_free(gap_part_);
    moved = true;
    gap_part_ = nullptr;
    gap_part_cap_ = 0;
    HIP_OK(g_malloc(&gap_part_, gap_need * sizeof(float)));
    gap_part_cap_ = gap_need;
  }
  // 3. launches
  auto EB = [](const TensorDesc& t) { return t.f16 ? 2.0 : 4.0; };  // bytes per stored element (the launches' algorithmic bytes)
  char nm[160];
  for (int oi = 0; oi < nops; ++oi) {
    if (plan_.ops[oi].kind == PlanOp::OUTPUT || folded[oi] || fused_dw[oi]) continue;
    PlanOp second_pass;  // RSE block: the ew's launch is its conv again, with the ew's stages behind the conv's own
    if (rse_conv[oi] >= 0) {
      const PlanOp& e = plan_.ops[oi];
      second_pass = plan_.ops[rse_conv[oi]];
      second_pass.out = e.out;
      PlanStage gr = e.ep[0];
      gr.kind = EP_GATERES;
      second_pass.ep.push_back(gr);
      for (size_t k = 2; k < e.ep.size(); ++k) second_pass.ep.push_back(e.ep[k]);
    }
    const PlanOp& op = rse_conv[oi] >= 0 ? second_pass : plan_.ops[oi];
    const TensorDesc& o = T[op.out];
    float* optr = arena_ + o.offset;
    Launch L;
    if (rse_first[oi]) {  // first pass: the pool's row sums, nothing else
      const TensorDesc& in = T[op.in];
      ConvRowsumArgs a{};
      a.in = arena_ + in.offset; a.w = dev_vec("rsw:" + op.w); a.part = gap_part_;
      a.rows = rag ? rows_of(in) : (long)in.n * in.h; a.W = in.w; a.Cin = op.cin; a.Cs_in = in.cs; a.Cs_out = o.cs;
      a.N = in.n; a.H = in.h; a.rag = rlevel(in);
      a.h16 = in.f16;
      if (!a.w) { err = "RSE block: no row-sum weight image for " + op.w; return false; }
      snprintf(nm, sizeof nm, "%s.%02d.conv1x1_%d_%d_rowsum", plan_.name.c_str(), oi, op.cin, op.cout);
      L.name = nm;
      L.flops = 2.0 * in.pixels() * op.cin * op.cout;
      L.bytes = EB(in) * in.pixels() * op.cin + 4.0 * a.rows * op.cout;
      L.fn = [this, a](hipStream_t s) {
        if (!launch_conv_rowsum(a, s)) this->launch_error_ = "launch_conv_rowsum: shape accepted at bind time was refused at launch";
      };
    }
    else switch (op.kind) {
      case PlanOp::CONV: case PlanOp::LINEAR: case PlanOp::DECONV: {
        const TensorDesc& in = T[gate_src[oi] >= 0 ? gate_src[oi] : op.in];
        Epilogue ep;
        if (op.kind == PlanOp::CONV && op.cin == 3) {
          if (!build_epilogue(op, ep, false, err)) return false;
          StemArgs a{};
          a.out = optr; a.w = dev_vec("stem:" + op.w);
          a.N = in.n; a.H = in.h; a.W = in.w; a.OH = o.h; a.OW = o.w; a.Cs_out = o.cs;
          a.KH = op.kh; a.KW = op.kw; a.SH = op.sh; a.SW = op.sw; a.PH = op.ph; a.PW = op.pw;
          a.M = o.pixels();
          a.rin = rlevel(in); a.rout = rlevel(o);
          a.h16 = o.f16;
          if (in.f16) { err = "stem input must be the plain f32 image"; return false; }
          if (o.cs != 8 && o.cs != 16) { err = "stem width not on this path"; return false; }
          snprintf(nm, sizeof nm, "%s.%02d.stem%dx%d_3_%d", plan_.name.c_str(), oi, op.kh, op.kw, op.cout);
          L.name = nm;
          L.flops = 2.0 * a.M * op.kh * op.kw * 3 * op.cout;
          L.bytes = 4.0 * in.numel() + EB(o) * a.M * op.cout;
          const bool ext_in = (op.in == 0);
          const float* in_ptr = ext_in ? nullptr : arena_ + in.offset;
          L.fn = [this, a, ep, ext_in, in_ptr](hipStream_t s) mutable {
            StemArgs b = a;
            b.in = ext_in ? this->bound_x_ : in_ptr;
            launch_stem(b, ep, s);
          };
        } else if (op.kind == PlanOp::DECONV && op.cout == 1) {
          // tail: deconv -> 1 channel, bias (scalar), sigmoid  (+ fused u8 threshold)
          float bias = 0.f;
          bool ok = op.ep.size() == 2 && (op.ep[0].kind == EP_SADD || op.ep[0].kind == EP_BIAS) && op.ep[1].kind == EP_ACT && op.ep[1].act == ACT_SIGMOID;
          if (!ok) { err = "unexpected DB head tail"; return false; }
          bias = host_w_[op.ep[0].n0].data[0];
          DetTailArgs a{};
          a.in = arena_ + in.offset; a.prob = optr; a.bitmap = det_bitmap_; a.w = dev_vec("tail:" + op.w);
          a.N = in.n; a.H = in.h; a.W = in.w; a.C = op.cin; a.Cs = in.cs; a.bias = bias; a.ithresh = det_ithresh_;
          a.M = in.pixels();
          a.h16 = in.f16;
          snprintf(nm, sizeof nm, "%s.%02d.det_tail", plan_.name.c_str(), oi);
          L.name = nm;
          L.flops = 2.0 * a.M * op.cin * 4;
          L.bytes = EB(in) * a.M * op.cin + 4.0 * a.M * 4 + (det_bitmap_ ? 1.0 * a.M * 4 : 0.0);
          L.fn = [a](hipStream_t s) { launch_det_tail(a, s); };
          if (img && dbhead_of[oi] < 0) { err = "ragged batch of images: the DB head needs its fused kernel (production launch list, OCR_FUSE on)"; return false; }
          if (dbhead_of[oi] >= 0) {
            const PlanOp& d = plan_.ops[dbhead_of[oi]];
            const TensorDesc& din = T[d.in];
            Epilogue epd;
            if (!build_epilogue(d, epd, true, err)) return false;
            DbHeadArgs h{};
            h.in = arena_ + din.offset; h.prob = optr; h.bitmap = det_bitmap_;
            h.w1 = dev_vec("dbh1:" + d.w); h.bias1 = epd.st[0].v0; h.bn_s = epd.st[1].v0; h.bn_t = epd.st[1].v1;
            if (d.cin == 24 && d.cout == 24) h.wfrag = dev_vec("dbhf:" + d.w);
            h.w2 = a.w; h.M = din.pixels(); h.N = din.n; h.H = din.h; h.W = din.w; h.Cs = din.cs;
            h.bias2 = a.bias; h.ithresh = a.ithresh;
            h.h16 = din.f16;
            if (img) h.rin = rlevel(din);
            snprintf(nm, sizeof nm, "%s.%02d.db_head_%d", plan_.name.c_str(), dbhead_of[oi], d.cin);
            L.name = nm;
            L.flops = 2.0 * h.M * (4.0 * d.cin * d.cout + 16.0 * d.cout);
            L.bytes = EB(din) * h.M * d.cin + 4.0 * h.M * 16 + (det_bitmap_ ? 1.0 * h.M * 16 : 0.0);
            const int C = d.cin;
            L.fn = [this, h, C](hipStream_t s) {
              if (!launch_db_head(h, C, s)) this->launch_error_ = "launch_db_head: shape accepted at bind time was refused at launch";
            };
          }
        } else {
          if (!build_epilogue(op, ep, true, err)) return false;
          ConvArgs a{};
          // precision "fp16": a conv on an f16 tensor runs in f16; one on a per-image f32 vector (the classifier's pool -> fc)
          // keeps the f32 kernel and fragments
          const bool hconv = half_ && in.f16;
          a.in = arena_ + in.offset; a.out = optr; a.wfrag = dev_vec((hconv ? "frag16:" : "frag:") + op.w);
          a.half = hconv ? 1 : 0;
          if (hconv && op.kind == PlanOp::CONV) a.wfrag_x16 = dev_vec("frag16x:" + op.w);  // (null: no such image - other than 1x1 and the 3x3 96-channel convs, or an odd number of octets)
          if (!o.plain && o.f16 != in.f16) { err = "precision fp16: a dense conv between tensors of different storage"; return false; }
          a.N = in.n; a.H = in.h; a.W = in.w; a.Cs_in = in.cs; a.C8 = in.cs / 8;
          a.KH = op.kh; a.KW = op.kw; a.PH = op.ph; a.PW = op.pw;
          a.need_nyx = 0;
          if (in.plain) { err = "conv input must be C8I"; return false; }
          if (rag) {
            // pointwise, or a stride-1 "same" conv (the recognizer's 1x3 neck convs, the detector's 3x3 neck / head convs):
            // rows in = rows out, one level
            if (in.lvl < 0 || o.lvl != in.lvl || o.h != in.h || op.kind == PlanOp::DECONV) { err = "ragged batch: dense conv must keep its input's shape"; return false; }
            for (auto& st : op.ep) if (st.kind == EP_ADDUP && !img) { err = "ragged batch of lines: an upsampled operand after a dense conv is not on this path"; return false; }
            a.rin = rlevel(in); a.rout = rlevel(o);
            if (img && op.kh == 3 && op.kw == 3) a.rtiles = tiles_table(o, 8, a.rtiles_total);  // (the 8x16 LDS-tile kernel)
          }
          if (op.kind == PlanOp::DECONV) {
            a.OH = in.h; a.OW = in.w; a.Cs_out = o.cs; a.Cout = op.cout; a.CoutPadded = o.cs;
            a.ColsStore = 4 * o.cs; a.out_mode = OUT_DECONV; a.KH = a.KW = 1; a.PH = a.PW = 0;
          } else {
            a.OH = o.h; a.OW = o.w; a.Cs_out = o.cs; a.Cout = op.cout; a.CoutPadded = o.cs;
            a.ColsStore = o.plain ? op.cout : o.cs; a.out_mode = o.plain ? OUT_PLAIN : OUT_C8I;
          }
          const int tiles = (a.ColsStore + 31) / 32;
          const int nt = conv_nt_for(tiles);
          a.NTtot = (tiles + nt - 1) / nt * nt;
          // linear -> softmax with only (arg max, max prob) wanted: the softmax is folded into the linear's
          // epilogue (no logits tensor) and a per-row combine; canonical groups are 128 columns wide
          const bool fuse_head = op.kind == PlanOp::LINEAR && o.plain && oi + 1 < nops && plan_.ops[oi + 1].kind == PlanOp::SOFTMAX &&
                                 plan_.ops[oi + 1].in == op.out && (head_amax_ || head_pmax_) && !head_probs_ && keep_all_ == 0 &&
                                 (tiles <= 4 || nt == 4);
          if (fuse_head) {
            a.out_mode = OUT_HEAD;
            const long hrows = in.pixels();
            const int groups = a.NTtot / nt;
            const size_t need = (size_t)hrows * groups * 3;
            if (need > head_part_cap_) {
              if (head_part_) (void)g_free(head_part_);
              moved = true;
              head_part_ = nullptr;
              head_part_cap_ = 0;
              HIP_OK(g_malloc(&head_part_, need * sizeof(float)));
              head_part_cap_ = need;
            }
            a.head_max = head_part_;
            a.head_sum = head_part_ + (size_t)hrows * groups;
            a.head_idx = (int*)(head_part_ + (size_t)hrows * groups * 2);
            fused_head_rows_ = hrows;
            fused_head_groups_ = groups;
          }
          a.zeros = dev_vec("zeros");
          double cat_bytes = 0;
          if (cat_of[oi] >= 0) {  // the folded concat: the tile fill reads its sources
            const PlanOp& k = plan_.ops[cat_of[oi]];
            a.in = nullptr;
            a.cat_n = (int)k.ins.size();
            a.cat_cs = T[k.ins[0]].cs;
            for (int j = 0; j < a.cat_n; ++j) {
              a.cat_src[j] = arena_ + T[k.ins[j]].offset;
              a.cat_up[j] = k.ups[j];
              if (k.ups[j] < 1 || (k.ups[j] & (k.ups[j] - 1))) { err = "folded concat: upsampling factor is not a power of two"; return false; }
              cat_bytes += EB(T[k.ins[j]]) * T[k.ins[j]].numel();
            }
          }
          if (gate_src[oi] >= 0) { a.gate = arena_ + T[gate_tid[oi]].offset; a.gate_hw = in.h * in.w; }
          a.M = (long)in.n * a.OH * a.OW;
          if (op.kind == PlanOp::LINEAR) a.M = (long)in.n * in.h * in.w;
          if (rag) a.M = in.pixels();
          const int taps = a.KH * a.KW;
          const char* kind = op.kind == PlanOp::DECONV ? "deconv" : (op.kind == PlanOp::LINEAR ? "linear" : "conv");
          snprintf(nm, sizeof nm, "%s.%02d.%s%dx%d_%d_%d", plan_.name.c_str(), oi, kind, a.KH, a.KW, op.cin, op.cout);
          L.name = nm;
          const double cols = op.kind == PlanOp::DECONV ? 4.0 * op.cout : op.cout;
          L.flops = 2.0 * a.M * taps * op.cin * cols;
          if (a.gate) L.name += "_gated";
          if (a.cat_n) L.name += "_cat" + std::to_string(a.cat_n);
          L.bytes = (a.cat_n ? cat_bytes : EB(in) * a.M * op.cin) + (a.out_mode == OUT_HEAD ? 12.0 * a.M * (a.NTtot / nt) : EB(o) * a.M * cols) +
                    (hconv ? 2.0 : 4.0) * taps * op.cin * cols;
          // measured (round 1): LDS staging wins for multi-tap convs (3x3 96->24: 58 vs 51 TFLOP/s), the direct kernel for 1x1
          // (480->480: 88 vs 71; thin K: 54 vs 39).  precision "fp16": the LDS-staged and the 4x4x1 kernels are f32 only - every
          // dense conv goes through the direct kernel
          const bool use_lds = !hconv && !a.gate && a.out_mode == OUT_C8I && taps > 1 && in.cs >= 64;
          if (dwpw_of[oi] >= 0) {
            const PlanOp& d = plan_.ops[dwpw_of[oi]];
            const TensorDesc& din = T[d.in];
            Epilogue epd;
            if (!build_epilogue(d, epd, false, err)) return false;
            DwPwArgs f{};
            f.c = a;
            f.c.in = nullptr;
            f.dw_in = arena_ + din.offset; f.dw_w = dev_vec("dw:" + d.w);
            f.dw_wq16 = dev_vec("dwq16:" + d.w); f.dw_wq32 = dev_vec("dwq32:" + d.w);
            if (!lab_from_epilogue(epd, f.dw_ep) || !lab_from_epilogue(ep, f.pw_ep)) { err = "dwpw: epilogue is not the LAB chain"; return false; }
            f.H = din.h; f.W = din.w; f.K = d.kh; f.SH = d.sh; f.SW = d.sw; f.PH = d.ph; f.PW = d.pw;
            if (rag) {
              f.rin = rlevel(din); f.rout = rlevel(o);
              f.c.rin = f.c.rout = RagLevel();
              const int th = dwpw_tile_rows(f);
              if (th <= 0) { err = "dwpw: no instance for this shape"; return false; }
              f.rtiles = tiles_table(o, th, f.rtiles_total);
            }
            snprintf(nm, sizeof nm, "%s.%02d.dwpw%dx%d_%d_%d_s%d%d", plan_.name.c_str(), dwpw_of[oi], d.kh, d.kw, op.cin, op.cout, d.sh, d.sw);
            L.name = nm;
            L.flops += 2.0 * a.M * d.kh * d.kw * d.c;
            if (din.f16 != (half_ != 0)) { err = "precision fp16: a fused depthwise block on an f32 tensor is not on this path"; return false; }
            L.bytes = EB(din) * din.pixels() * d.c + EB(o) * a.M * cols + (half_ ? 2.0 : 4.0) * op.cin * cols + 4.0 * d.kh * d.kw * d.c;
            L.fn = [this, f](hipStream_t s) {
              if (!launch_dwpw(f, s)) this->launch_error_ = "launch_dwpw: shape accepted at bind time was refused at launch";
            };
          } else if (use_lds) {
            const float* c24 = dev_vec("c24:" + op.w);
            L.fn = [this, a, ep, nt, c24](hipStream_t s) {
              if (launch_conv3x3_c24(a, ep, c24, s)) return;
              bool cat_same_res = a.cat_n > 0;  // (a same-resolution concat folds into the LDS-staged conv's chunk loads)
              for (int j = 0; j < a.cat_n; ++j) cat_same_res = cat_same_res && a.cat_up[j] == 1;
              if (a.cat_n && !cat_same_res) { this->launch_error_ = "folded concat: the 3x3 conv's 4x4x1 kernel refused the launch"; return; }
              if (a.cat_n || !launch_conv3x3_tile(a, ep, nt, s)) launch_conv_lds(a, ep, nt, s);
            };
          }
          else {
            // Small GEMMs (a request's few text lines, the small-width rec launches): fewer column tiles per wave, as long
            // as that keeps dividing the fragment image's tile count, until the launch has ~2 workgroups per CU - a wave's
            // K walk is a chain of NT * K / 2 dependent-in-order MFMAs and with one wave per SIMD its length IS the
            // kernel's time (rec op 30 on 32 lines: 720 -> 240 MFMAs per wave).  Results do not depend on NT.
            int ntl = nt;
            if (a.out_mode != OUT_HEAD && rt_options().conv_small_nt) {  // OCR_CONV_SMALL_NT=0: keep the table's NT (A/B)
              auto wgs = [&](int t) { return ((a.M + 127) / 128) * (long)(a.NTtot / t); };
              while (ntl > 1 && wgs(ntl) < 512) {
                int t = ntl - 1;
                while (t > 1 && a.NTtot % t) --t;
                ntl = t;
              }
            }
            // Big 1x1 convs: two pixel tiles per wave (kernels_net.hip, conv_mfma_mt_kernel) - a weight fragment feeds two
            // MFMAs, half the workgroups stage parameters; worth it from K = 192 on while the launch still has ~4
            // workgroups per CU (rec ops 25/30/32/34, det ops 30/38).  Results do not depend on the tiling.
            const bool mt2 = rt_options().conv_mt2 && (nt == 3 || nt == 4) && taps == 1 && a.out_mode == OUT_C8I &&
                             (rt_options().conv_mt2_force || (ntl == nt && in.cs >= 192 && ((a.M + 255) / 256) * (long)(a.NTtot / nt) >= 1024));
            if (mt2) ntl = nt;  // (the two-tile kernel is instantiated for the table's NT)
            const bool half_tile = hconv && taps == 9 && (!rag || img);  // precision "fp16": the LDS-resident 3x3 tile kernel has an f16 form (uniform batches, ragged batches of images)
            L.fn = [this, a, ep, ntl, nt, mt2, half_tile](hipStream_t s) {
              if (half_tile && launch_conv3x3_tile(a, ep, nt, s)) return;
              if (a.cat_n) { this->launch_error_ = "folded concat: the 3x3 conv's f16 tile kernel refused the launch"; return; }
              if (mt2 && launch_conv_mfma_mt2(a, ep, ntl, s)) return;
              if (!launch_conv_mfma(a, ep, ntl, s)) this->launch_error_ = "launch_conv_mfma: this conv shape / output mode is not instantiated";
            };
          }
        }
      } break;
      case PlanOp::DW: {
        if (xdw_of[oi] >= 0) {  // the expand 1x1 conv in front of it runs inside this launch (kernels_xdw.hip)
          const PlanOp& c = plan_.ops[xdw_of[oi]];
          const TensorDesc& xin = T[c.in];
          Epilogue ce, de;
          if (!build_epilogue(c, ce, true, err) || !build_epilogue(op, de, false, err)) return false;
          XdwArgs a{};
          a.x = arena_ + xin.offset; a.wfrag = dev_vec("frag:" + c.w);
          a.e_sc = ce.st[0].v0; a.e_sh = ce.st[0].v1;
          a.dw_w = dev_vec("dw:" + op.w); a.d_sc = de.st[0].v0; a.d_sh = de.st[0].v1;
          a.out = optr; a.part = dw_rowsum[oi] ? gap_part_ : nullptr;
          a.N = xin.n; a.Hin = xin.h; a.Hout = o.h; a.W = xin.w; a.Cs_in = xin.cs; a.Cs_e = o.cs;
          const int tiles = (T[c.out].cs + 31) / 32, nt = conv_nt_for(tiles);
          a.NTtot = (tiles + nt - 1) / nt * nt;
          a.K = op.kh; a.SH = op.sh; a.SW = op.sw; a.PH = op.ph; a.PW = op.pw;
          if (!a.wfrag || !a.e_sc || !a.e_sh || !a.dw_w || !a.d_sc || !a.d_sh) { err = "expand -> depthwise block: a parameter image is missing for " + c.w; return false; }
          snprintf(nm, sizeof nm, "%s.%02d.xdw%dx%d_%d_%d_s%d%d%s", plan_.name.c_str(), oi, op.kh, op.kw, c.cin, op.c, op.sh, op.sw, dw_rowsum[oi] ? "_rowsum" : "");
          L.name = nm;
          L.flops = 2.0 * xin.pixels() * c.cin * c.cout + 2.0 * o.pixels() * op.kh * op.kw * op.c;
          L.bytes = EB(xin) * xin.pixels() * c.cin + EB(o) * o.pixels() * op.c;
          L.fn = [this, a](hipStream_t s) {
            if (!launch_xdw(a, s)) this->launch_error_ = "launch_xdw: shape accepted at bind time was refused at launch";
          };
          break;
        }
        const TensorDesc& in = T[op.in];
        Epilogue ep;
        if (!build_epilogue(op, ep, false, err)) return false;
        for (auto& st : op.ep) if (st.kind == EP_ADDUP) { err = "addup after a depthwise conv is not on this path"; return false; }
        DwArgs a{};
        a.in = arena_ + in.offset; a.out = optr; a.w = dev_vec("dw:" + op.w);
        a.N = in.n; a.H = in.h; a.W = in.w; a.OH = o.h; a.OW = o.w; a.Cs = o.cs; a.K = op.kh;
        a.SH = op.sh; a.SW = op.sw; a.PH = op.ph; a.PW = op.pw; a.M = o.pixels();
        if (dw_rowsum[oi]) a.rowsum = gap_part_;
        a.h16 = o.f16;
        if (in.f16 != o.f16) { err = "precision fp16: depthwise conv between tensors of different storage"; return false; }
        if (rag) {
          a.rin = rlevel(in); a.rout = rlevel(o);
          a.OW = min_w(o);            // the launcher picks its patch from the narrowest / lowest sample
          if (img) a.OH = min_h(o);
          const int to = dw_patch_to(a.OW, a.SW, img ? a.OH : o.h, a.K), pr = dw_patch_r(img ? a.OH : o.h, a.K);
          const bool rs = dw_rowsum[oi] != 0;
          const TensorDesc ot = o;
          a.rwork = work_table("dw:" + std::to_string(o.lvl) + ":" + std::to_string(o.h) + ":" + std::to_string(to) + ":" + std::to_string(pr) + (rs ? ":rs" : ""),
                               [&, to, pr, rs](int i) {
                                 const int bands = (sh_of(ot, i) + pr - 1) / pr;
                                 return rs ? bands : bands * ((sw_of(ot, i) + to - 1) / to);
                               }, a.rwork_total);
        }
        snprintf(nm, sizeof nm, "%s.%02d.dw%dx%d_%d_s%d%d%s", plan_.name.c_str(), oi, op.kh, op.kw, op.c, op.sh, op.sw, dw_rowsum[oi] ? "_rowsum" : "");
        L.name = nm;
        L.flops = 2.0 * a.M * op.kh * op.kw * op.c;
        L.bytes = EB(in) * in.pixels() * op.c + EB(o) * a.M * op.c;
        // the low maps' 5x5 layers: region staged through LDS (kernels_dwlds.hip); asked now, on the device that will run it
        const bool lds_dw = rt_options().dw_lds && launch_dw_lds(a, ep, nullptr, true);
        L.fn = [this, a, ep, lds_dw](hipStream_t s) {
          if (lds_dw) {
            if (!launch_dw_lds(a, ep, s)) this->launch_error_ = "launch_dw_lds: shape accepted at bind time was refused at launch";
            return;
          }
          launch_dw(a, ep, s);
        };
      } break;
      case PlanOp::EW: {
        const TensorDesc& in = T[op.in];
        Epilogue ep;
        if (!build_epilogue(op, ep, false, err)) return false;
        const float* ip = arena_ + in.offset;
        const long M = o.pixels();
        const int H2 = o.h, W2 = o.w, Cs = o.cs;
        if (rag && !img) for (auto& st : op.ep) if (st.kind == EP_ADDUP) { err = "ragged batch: upsampled operand is not on this path"; return false; }
        const RagLevel rl = rlevel(o);
        const int nl = o.n;
        snprintf(nm, sizeof nm, "%s.%02d.ew_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        L.bytes = EB(o) * M * op.c * (2.0 + (double)op.ep.size() - 1.0);
        const bool h16 = o.f16;
        if (in.f16 != o.f16) { err = "precision fp16: elementwise op between tensors of different storage"; return false; }
        L.fn = [ip, optr, M, H2, W2, Cs, ep, nl, rl, h16](hipStream_t s) { launch_ew(ip, optr, M, H2, W2, Cs, ep, s, nl, rl, h16); };
      } break;
      case PlanOp::GAP: {
        const TensorDesc& in = T[op.in];
        const float* ip = arena_ + in.offset;
        float* part = gap_part_;
        const int n = in.n, h = in.h, w = in.w, cs = in.cs;
        const RagLevel rl = rlevel(in);
        const long grows = rag ? rows_of(in) : 0;
        snprintf(nm, sizeof nm, "%s.%02d.gap_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        L.bytes = EB(in) * in.pixels() * op.c;
        const bool h16 = in.f16;
        if (oi > 0 && (dw_rowsum[oi - 1] || rse_first[oi - 1])) {  // the row sums are already in `part` (written by the depthwise conv / the conv's row-sum pass before this op)
          L.bytes = 4.0 * (double)n * h * op.c;
          L.fn = [part, optr, n, h, w, cs, rl](hipStream_t s) { launch_gap_cols(part, optr, n, h, w, cs, s, rl); };
        } else
        L.fn = [ip, part, optr, n, h, w, cs, rl, grows, h16](hipStream_t s) { launch_gap(ip, part, optr, n, h, w, cs, s, rl, grows, h16); };
      } break;
      case PlanOp::SEFC: {
        const TensorDesc& in = T[op.in];
        SeArgs a{};
        a.in = arena_ + in.offset; a.out = optr;
        a.w1 = dev_vec("raw:" + op.w1); a.b1 = dev_vec("raw:" + op.b1);
        a.w2 = dev_vec("raw:" + op.w2); a.b2 = dev_vec("raw:" + op.b2);
        a.C = op.c; a.Cs = o.cs; a.R = op.cr; a.slope = op.slope; a.offset = op.offset;
        const int n = in.n;
        snprintf(nm, sizeof nm, "%s.%02d.sefc_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        L.flops = 4.0 * n * op.c * op.cr;
        L.fn = [a, n](hipStream_t s) { launch_sefc(a, n, s); };
      } break;
      case PlanOp::CONCAT: {
        ConcatArgs a{};
        a.out = optr; a.H = o.h; a.W = o.w; a.Cs = o.cs; a.M = o.pixels();
        if (img) { a.rout = rlevel(o); a.N = N; }
        else if (rag) { a.H = 1; a.W = (int)a.M; }  // lines: same-resolution sources (checked above), one row axis
        a.nsrc = (int)op.ins.size();
        if (a.nsrc > 4) { err = "concat arity not on this path"; return false; }
        int off = 0;
        for (int j = 0; j < a.nsrc; ++j) {
          const TensorDesc& sj = T[op.ins[j]];
          if (sj.c % 8) { err = "concat source channels must be a multiple of 8"; return false; }
          if (sj.f16 != o.f16) { err = "precision fp16: concat of tensors of different storage"; return false; }
          a.src[j] = arena_ + sj.offset; a.coff[j] = off; a.scs[j] = sj.cs; a.up[j] = op.ups[j];
          off += sj.cs;
        }
        snprintf(nm, sizeof nm, "%s.%02d.concat_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        a.h16 = o.f16;
        L.bytes = 2.0 * EB(o) * a.M * op.c;
        L.fn = [a](hipStream_t s) { launch_concat(a, s); };
      } break;
      case PlanOp::POOL: {
        const TensorDesc& in = T[op.in];
        PoolArgs a{};
        a.in = arena_ + in.offset; a.out = optr; a.N = in.n; a.H = in.h; a.W = in.w; a.OH = o.h; a.OW = o.w;
        a.Cs = o.cs; a.KH = op.kh; a.KW = op.kw; a.SH = op.sh; a.SW = op.sw; a.is_max = op.pool_max;
        a.M = o.pixels();
        a.rin = rlevel(in); a.rout = rlevel(o);
        a.h16 = o.f16;
        if (in.f16 != o.f16) { err = "precision fp16: pool between tensors of different storage"; return false; }
        snprintf(nm, sizeof nm, "%s.%02d.pool_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        L.bytes = EB(in) * in.pixels() * op.c + EB(o) * a.M * op.c;
        L.fn = [a](hipStream_t s) { launch_pool(a, s); };
      } break;
      case PlanOp::LN: {
        const TensorDesc& in = T[op.in];
        const float* ip = arena_ + in.offset;
        const long rows = in.pixels();
        const int C = op.c, Cs = in.cs;
        const float eps = op.eps;
        const float* g = dev_vec("raw:" + op.g);
        const float* b = dev_vec("raw:" + op.b);
        snprintf(nm, sizeof nm, "%s.%02d.ln_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        L.bytes = 2.0 * EB(in) * rows * C;
        const bool h16 = in.f16;
        if (in.f16 != o.f16) { err = "precision fp16: layer norm between tensors of different storage"; return false; }
        L.fn = [ip, optr, rows, C, Cs, eps, g, b, h16](hipStream_t s) { launch_ln(ip, optr, rows, C, Cs, eps, g, b, s, h16); };
      } break;
      case PlanOp::ATTN: {
        const TensorDesc& in = T[op.in];
        if (in.h != 1) { err = "attention expects a sequence (H == 1): rec input height must reduce to 1"; return false; }
        const float* ip = arena_ + in.offset;
        const int n = in.n, t = in.w, heads = op.heads, hd = op.hd, csi = in.cs, cso = o.cs;
        const float sc = op.scale;
        const RagLevel rl = rlevel(in);
        if (rag && !attn_ragged_fits(t)) { err = "ragged batch: line too wide for the attention kernel"; return false; }
        double tt = (double)n * t * t;  // sum over lines of T^2
        if (rag) { tt = 0; for (int v : LW[in.lvl]) tt += (double)v * v; }
        snprintf(nm, sizeof nm, "%s.%02d.attn_%dx%d", plan_.name.c_str(), oi, op.heads, op.hd);
        L.name = nm;
        L.flops = 3.0 * 2.0 * heads * tt * hd + 2.0 * heads * tt * hd;
        L.bytes = EB(in) * in.pixels() * (3.0 + 1.0) * heads * hd;
        const bool h16 = in.f16;
        if (in.f16 != o.f16) { err = "precision fp16: attention between tensors of different storage"; return false; }
        L.fn = [ip, optr, n, t, heads, hd, csi, cso, sc, rl, h16](hipStream_t s) { launch_attn(ip, optr, n, t, heads, hd, csi, cso, sc, s, rl, h16); };
      } break;
      case PlanOp::SOFTMAX: {
        const TensorDesc& in = T[op.in];
        const float* ip = arena_ + in.offset;
        const long rows = in.pixels();
        const int C = op.c;
        snprintf(nm, sizeof nm, "%s.%02d.softmax_%d", plan_.name.c_str(), oi, op.c);
        L.name = nm;
        L.bytes = 4.0 * rows * C * 3;
        // `optr` (the plan's softmax tensor) is only filled when no external sink is set or probs are requested
        if (fused_head_rows_ == rows && oi > 0 && plan_.ops[oi - 1].kind == PlanOp::LINEAR && plan_.ops[oi - 1].out == op.in) {
          // second half of the fused head (the linear before this op ran in OUT_HEAD mode)
          const float* hm = head_part_;
          const float* hs = head_part_ + (size_t)rows * fused_head_groups_;
          const int* hi = (const int*)(head_part_ + (size_t)rows * fused_head_groups_ * 2);
          const int G = fused_head_groups_;
          L.bytes = 12.0 * rows * G;
          L.fn = [this, hm, hs, hi, rows, G](hipStream_t s) { launch_head_combine(hm, hs, hi, rows, G, this->head_amax_, this->head_pmax_, s); };
        } else {
          L.fn = [this, ip, optr, rows, C](hipStream_t s) {
            float* probs = (this->head_amax_ || this->head_pmax_) ? this->head_probs_ : optr;
            launch_softmax_argmax(ip, probs, this->head_amax_, this->head_pmax_, rows, C, s);
          };
        }
      } break;
      default: break;
    }
    // instance tag: the same op at another bound shape is another roofline row
    if (img) snprintf(nm, sizeof nm, "@%dx~%ldx~%ld", N, irows / N, ipix / irows);  // ragged images: mean height, mean width
    else if (rag) snprintf(nm, sizeof nm, "@%dx%dx~%ld", N, H, ltot[0] / N);  // ragged: the lines' mean width
    else snprintf(nm, sizeof nm, "@%dx%dx%d", N, H, W);
    L.name += nm;
    launches_.push_back(std::move(L));
  }
  if (tab_overflow) { err = "ragged batch: work-table space exhausted"; return false; }
  if (moved) cache_.clear();
  while (cache_.size() >= max_bindings_) {  // least recently used out
    auto old = cache_.begin();
    for (auto it = cache_.begin(); it != cache_.end(); ++it)
      if (it->second->stamp < old->second->stamp) old = it;
    cache_.erase(old);
  }
  B->tensors = tensors_;
  cur_ = B.get();
  cache_[shape_key(N, H, W, widths, heights)] = std::move(B);
  return true;
}

bool Net::run(const float* x, int N, int H, int W, hipStream_t s, std::string& err) {
  if (!cur_ || !cur_->widths.empty() || cur_->n != N || cur_->h != H || cur_->w != W) {
    auto it = cache_.find(shape_key(N, H, W, nullptr, nullptr));
    if (it != cache_.end()) {
      cur_ = it->second.get();
      tensors_ = cur_->tensors;
    } else {
      cur_ = nullptr;
      if (!bind(N, H, W, err)) { invalidate(); return false; }
    }
  }
  return run_bound(x, s, err);
}

bool Net::bind_ragged(int H, const int* widths, int N, std::string& err) {
  if (N < 1 || !widths) { err = "ragged batch: no lines"; return false; }
  if (!ragged_ok(H, widths, N, err)) return false;
  if (cur_ && cur_->heights.empty() && cur_->h == H && (int)cur_->widths.size() == N && std::equal(widths, widths + N, cur_->widths.begin())) return true;
  auto it = cache_.find(shape_key(N, H, 0, widths, nullptr));
  // (the key is the run-length form: a cached binding with the same runs has the same widths line by line)
  if (it != cache_.end()) {
    cur_ = it->second.get();
    tensors_ = cur_->tensors;
    return true;
  }
  cur_ = nullptr;
  // ragged bindings carry per-line tables: keep fewer of them than of the uniform ones
  size_t nrag = 0;
  for (auto& kv : cache_) nrag += !kv.second->widths.empty();
  while (nrag >= kMaxRaggedBindings) {
    auto old = cache_.end();
    for (auto i2 = cache_.begin(); i2 != cache_.end(); ++i2)
      if (!i2->second->widths.empty() && (old == cache_.end() || i2->second->stamp < old->second->stamp)) old = i2;
    cache_.erase(old);
    --nrag;
  }
  if (!bind(N, H, 0, err, widths)) { invalidate(); return false; }
  return true;
}

bool Net::run_ragged(const float* x, int H, const int* widths, int N, hipStream_t s, std::string& err) {
  if (!bind_ragged(H, widths, N, err)) return false;
  return run_bound(x, s, err);
}

bool Net::run_ragged_images(const float* x, const int* heights, const int* widths, int N, hipStream_t s, std::string& err) {
  if (N < 1 || !widths || !heights) { err = "ragged batch: no images"; return false; }
  const bool same = cur_ && (int)cur_->widths.size() == N && (int)cur_->heights.size() == N &&
                    std::equal(widths, widths + N, cur_->widths.begin()) && std::equal(heights, heights + N, cur_->heights.begin());
  if (!same) {
    auto it = cache_.find(shape_key(N, 0, 0, widths, heights));
    if (it != cache_.end()) {
      cur_ = it->second.get();
      tensors_ = cur_->tensors;
    } else {
      cur_ = nullptr;
      size_t nrag = 0;
      for (auto& kv : cache_) nrag += !kv.second->widths.empty();
      while (nrag >= kMaxRaggedBindings) {
        auto old = cache_.end();
        for (auto i2 = cache_.begin(); i2 != cache_.end(); ++i2)
          if (!i2->second->widths.empty() && (old == cache_.end() || i2->second->stamp < old->second->stamp)) old = i2;
        cache_.erase(old);
        --nrag;
      }
      if (!bind(N, 0, 0, err, widths, heights)) { invalidate(); return false; }
    }
  }
  return run_bound(x, s, err);
}

bool Net::run_bound(const float* x, hipStream_t s, std::string& err) {
  Binding& B = *cur_;
  B.stamp = ++clock_;
  stats_.runs++;
  if (B.rag_dev && !B.rag_uploaded) {  // the line tables travel once, on the stream that runs the launches
    HIP_OK(hipMemcpyAsync(B.rag_dev, B.rag_host.data(), B.rag_host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    B.rag_uploaded = true;
  }
  bound_x_ = x;
  // event timing needs plain launches - but only where an event would be placed: with a name filter (bench.py times the
  // dominant kernel alone inside its timed region) every binding without a matching launch keeps replaying its graph
  bool timed_here = timing_;
  if (timing_ && !timing_filter_.empty()) {
    timed_here = false;
    for (const auto& L : B.launches)
      if (L.name.find(timing_filter_) != std::string::npos) { timed_here = true; break; }
  }
  const std::string refuse = rt_refuse_launch();  // fault injection (tests): "" unless ocr_selftest_refuse_launch set it
  launch_error_.clear();
  auto issue = [&](Launch& L) {
    if (!refuse.empty() && L.name.find(refuse) != std::string::npos) { launch_error_ = "launch refused (self-test): " + L.name; return; }
    L.fn(s);
  };
  const bool graphs = graphs_ && !timed_here && !keep_all_ && !B.graph_failed && !B.launches.empty() && refuse.empty();
  const void* head[3] = {head_probs_, head_amax_, head_pmax_};
  const bool repeat = B.last_x == x;  // the caller feeds this shape from one buffer: worth recording
  B.last_x = x;
  if (graphs && B.graph_exec && B.graph_x == x && B.graph_stream == s && !memcmp(B.graph_head, head, sizeof head)) {
    HIP_OK(hipGraphLaunch(B.graph_exec, s));
    HIP_OK(hipGetLastError());
    stats_.graph_replays++;
    return true;
  }
  if (graphs && repeat) {
    if (B.graph_exec) { (void)hipGraphExecDestroy(B.graph_exec); B.graph_exec = nullptr; }
    hipGraph_t g = nullptr;
    bool ok = false;
    std::unique_lock<std::shared_mutex> capture_lock(capture_mutex());  // (hip_guard.h: no allocation / synchronous copy of another thread meanwhile)
    if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      for (auto& L : B.launches) issue(L);
      ok = hipStreamEndCapture(s, &g) == hipSuccess && g && launch_error_.empty() &&
           hipGraphInstantiate(&B.graph_exec, g, nullptr, nullptr, 0) == hipSuccess;
      if (g) (void)hipGraphDestroy(g);
      if (!launch_error_.empty()) { B.graph_exec = nullptr; B.graph_failed = true; err = launch_error_; return false; }
    }
    capture_lock.unlock();
    if (ok) {
      B.graph_x = x;
      B.graph_stream = s;
      memcpy(B.graph_head, head, sizeof head);
      HIP_OK(hipGraphLaunch(B.graph_exec, s));
      HIP_OK(hipGetLastError());
      return true;
    }
    B.graph_exec = nullptr;
    B.graph_failed = true;    // never tried again for this binding
    (void)hipGetLastError();  // capture not available: plain launches
  }
  for (size_t i = 0; i < B.launches.size(); ++i) {
    Launch& L = B.launches[i];
    if (timing_ && (timing_filter_.empty() || L.name.find(timing_filter_) != std::string::npos)) {
      hipEvent_t a, b;
      if (ev_pool_.size() >= 2) { a = ev_pool_.back(); ev_pool_.pop_back(); b = ev_pool_.back(); ev_pool_.pop_back(); }
      else { HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b)); }
      HIP_OK(hipEventRecord(a, s));
      issue(L);
      HIP_OK(hipEventRecord(b, s));
      ev_pending_.push_back({a, b, L.name, L.flops, L.bytes});
    } else {
      issue(L);
    }
    if (!launch_error_.empty()) { err = launch_error_; return false; }
  }
  HIP_OK(hipGetLastError());
  return true;
}

void Net::collect_timings() {
  for (auto& p : ev_pending_) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& t = timings_[p.name];
      t.ms += ms;
      t.count += 1;
      t.flops += p.flops;
      t.bytes += p.bytes;
    }
    ev_pool_.push_back(p.a);
    ev_pool_.push_back(p.b);
  }
  ev_pending_.clear();
}

bool Net::fetch_logical(int tid, std::vector<float>& host, int dims[4], hipStream_t s, std::string& err) {
  if (tid < 0) tid = out_tid_;
  if (tid <= 0 || tid >= plan_.ntensors || !cur_) { err = "bad tensor id"; return false; }
  if (!cur_->exists[tid]) { err = "tensor is fused away under this binding (never written to device memory)"; return false; }
  const TensorDesc& t = tensors_[tid];
  dims[0] = t.n; dims[1] = t.h; dims[2] = t.w; dims[3] = t.c;
  const long M = t.pixels();
  if (t.lvl >= 0) { dims[0] = 1; dims[1] = 1; dims[2] = (int)M; }  // ragged: the lines' [h][w] blocks one after the other
  host.resize((size_t)M * t.c);
  if (t.plain) {
    HIP_OK(hipMemcpyAsync(host.data(), arena_ + t.offset, host.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_OK(g_stream_sync(s));
    return true;
  }
  float* tmp = nullptr;
  HIP_OK(g_malloc(&tmp, host.size() * sizeof(float)));
  launch_c8i_to_plain(arena_ + t.offset, tmp, M, t.c, t.cs, s, t.f16);  // (an f16 tensor arrives converted up: exact)
  hipError_t e = hipMemcpyAsync(host.data(), tmp, host.size() * sizeof(float), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = g_stream_sync(s);
  (void)g_free(tmp);
  if (e != hipSuccess) { err = hipGetErrorString(e); return false; }
  return true;
}

}  // namespace ocr
