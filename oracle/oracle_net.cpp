// TEST INFRASTRUCTURE — CPU oracle, not part of the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// fp32 restatement of the three networks that the reference hands to Paddle Inference
// (/root/reference/src/ocr_det.cpp:116-132, ocr_cls.cpp:67-84, ocr_rec.cpp:76-92), executed over
// the build's fused layer table (cpp-paddle-ocr_amd/plans/*.plan).  Paddle Inference itself is a
// third-party binary that is absent from /root/reference; operator semantics
// follow SURVEY.md Appendix A.  PARITY UNPINNED against real Paddle: no golden logits exist in
// the reference; this file is cross-checked against an independent unfused torch-CPU
// interpretation of the same .pdmodel (oracle/graph_ref.py) and, for cls, on the real weights.
//
// Arithmetic contract ("canonical order") — the HIP kernels are built to reproduce
// these roundings exactly, which is what makes boxes / CTC ids bit-identical:
//   * every contraction is ONE f32 fmaf chain from 0 in ascending k = ((kh*KW)+kw)*Cin + ci
//     (zero padding contributes fmaf(0,w,acc)); bias and the rest follow as separate roundings;
//   * a*x+b stages are mul then add (two roundings); division and sqrt are IEEE;
//   * the PP-LCNetV3 "learnable affine block" chain  bias | *s0 | +a0 [| hswish | *s1 | +a1]  behind a conv is FOLDED when
//     the parameters are resolved (round 5, fold_lab below; what every inference engine does to scale / shift ops around a
//     conv): the weights carry s0, one bias vector carries s0*b + a0, the hard-swish keeps its product u = y*clamp(y+3,0,6)
//     and its 1/6 travels with s1 into ONE fma(u, s1/6, a1) - or, when the only reader is a 1x1 conv, into that conv's
//     weights and bias (a 1x1 conv has no padding: exact in value).  Other hard-swish ops keep the IEEE division;
//   * exp is ocr_expf below (Cephes-style, fmaf Horner), never libm;
//   * reductions: GAP = row-sequential then column-sequential; LN/attention sequential;
//     row softmax: groups of 128 columns, two interleaved chains per group, groups folded in order
//     with exp(m_g - M) rescaling (see the softmax op below).
// Round 6 - the hand-written server plans of BASELINE configs[4] (cpp-paddle-ocr_amd/plans/srv_*.plan, tools/make_server_plans.py:
// NOT reference artifacts) add: `pool` with padding (max pools skip what lies outside), `act:gelu` (ocr_erff below), `addpos`
// (a per-position parameter added to every image), `attn` on a token grid with a local window (SVTR's Local mixer: the
// additive -inf mask of rec_svtrnet.py restated as "keys outside the window do not take part").
// Layout here is plain NHWC; the device uses an octet-interleaved channel order, which does not
// change any of the above.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

inline float ocr_expf(float x) {
  x = fminf(fmaxf(x, -87.0f), 88.0f);
  float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500E-4f;
  p = fmaf(p, r, 1.3981999507E-3f);
  p = fmaf(p, r, 8.3334519073E-3f);
  p = fmaf(p, r, 4.1665795894E-2f);
  p = fmaf(p, r, 1.6666665459E-1f);
  p = fmaf(p, r, 5.0000001201E-1f);
  float r2 = r * r;
  float y = fmaf(p, r2, r);
  y = y + 1.0f;
  int ni = (int)n;
  uint32_t bits = (uint32_t)(ni + 127) << 23;
  float sc;
  memcpy(&sc, &bits, 4);
  return y * sc;
}

// erf for the exact GELU (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7), every step spelled out: the device's f32 build
// performs the same operations in the same order (csrc/srv_kernels.hip, srv_erff)
inline float ocr_erff(float x) {
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

enum StageKind { S_BIAS, S_SMUL, S_SADD, S_BN, S_ACT, S_MULC, S_ADDT, S_ADDUP, S_SFMA, S_ADDPOS };
enum ActKind { A_RELU, A_HSWISH, A_HSIG, A_SWISH, A_SIGMOID, A_HSW6, A_GELU };

struct Stage {
  int kind = 0, act = 0;
  float p0 = 0, p1 = 0;
  std::string n0, n1, n2, n3;  // param names
  int tid = -1, up = 1;
  std::vector<float> v0, v1;   // resolved per-channel vectors
};

struct Op {
  std::string kind;
  std::map<std::string, std::string> kv;
  std::vector<Stage> ep;
  int geti(const char* k, int d = 0) const {
    auto it = kv.find(k);
    return it == kv.end() ? d : atoi(it->second.c_str());
  }
  float getf(const char* k) const { return strtof(kv.at(k).c_str(), nullptr); }
  std::vector<int> getlist(const char* k) const {
    std::vector<int> r;
    std::stringstream ss(kv.at(k));
    std::string t;
    while (std::getline(ss, t, ',')) r.push_back(atoi(t.c_str()));
    return r;
  }
};

struct Tensor {
  int n = 0, h = 0, w = 0, c = 0;
  std::vector<float> d;
  void alloc(int n_, int h_, int w_, int c_) {
    n = n_; h = h_; w = w_; c = c_;
    d.assign((size_t)n * h * w * c, 0.f);
  }
  float* at(int in, int ih, int iw) { return d.data() + (((size_t)in * h + ih) * w + iw) * c; }
  const float* at(int in, int ih, int iw) const { return d.data() + (((size_t)in * h + ih) * w + iw) * c; }
};

struct Param {
  std::vector<int> dims;
  std::vector<float> d;
};

struct Net {
  std::vector<Op> ops;
  int ntensors = 0;
  std::map<std::string, Param> params;
  std::vector<Tensor> t;
  std::string err;
  bool resolved = false;
  // transposed weights cache: op index -> [K][Cout]
  std::map<int, std::vector<float>> wt;
  // depthwise weights after the LAB fold: op index -> [C][kh*kw] (the parameter's own layout)
  std::map<int, std::vector<float>> wdw;
  std::vector<Op> ops_plan;  // the stage lists as parsed (resolve() rewrites `ops` from these)
};

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> r;
  std::stringstream ss(s);
  std::string t;
  while (std::getline(ss, t, sep)) r.push_back(t);
  return r;
}

bool parse_plan(Net& net, const char* text) {
  std::stringstream ss(text);
  std::string line;
  while (std::getline(ss, line)) {
    if (line.empty() || line[0] == '#') continue;
    auto toks = split(line, ' ');
    if (toks[0] == "plan") {
      for (auto& tk : toks)
        if (tk.rfind("ntensors=", 0) == 0) net.ntensors = atoi(tk.c_str() + 9);
      continue;
    }
    Op op;
    op.kind = toks[0];
    for (size_t i = 1; i < toks.size(); ++i) {
      auto eq = toks[i].find('=');
      std::string k = toks[i].substr(0, eq), v = toks[i].substr(eq + 1);
      if (k == "ep") {
        for (auto& st : split(v, '|')) {
          auto c = st.find(':');
          std::string sk = st.substr(0, c);
          auto args = split(st.substr(c + 1), ',');
          Stage s;
          if (sk == "bias") { s.kind = S_BIAS; s.n0 = args[0]; }
          else if (sk == "smul") { s.kind = S_SMUL; s.n0 = args[0]; }
          else if (sk == "sadd") { s.kind = S_SADD; s.n0 = args[0]; }
          else if (sk == "bn") { s.kind = S_BN; s.n0 = args[0]; s.n1 = args[1]; s.n2 = args[2]; s.n3 = args[3]; s.p0 = strtof(args[4].c_str(), nullptr); }
          else if (sk == "act") {
            s.kind = S_ACT;
            if (args[0] == "relu") s.act = A_RELU;
            else if (args[0] == "hswish") s.act = A_HSWISH;
            else if (args[0] == "hsig") { s.act = A_HSIG; s.p0 = strtof(args[1].c_str(), nullptr); s.p1 = strtof(args[2].c_str(), nullptr); }
            else if (args[0] == "swish") s.act = A_SWISH;
            else if (args[0] == "sigmoid") s.act = A_SIGMOID;
            else if (args[0] == "gelu") s.act = A_GELU;
            else return false;
          }
          else if (sk == "mulc") { s.kind = S_MULC; s.tid = atoi(args[0].c_str()); }
          else if (sk == "addt") { s.kind = S_ADDT; s.tid = atoi(args[0].c_str()); }
          else if (sk == "addup") { s.kind = S_ADDUP; s.tid = atoi(args[0].c_str()); s.up = atoi(args[1].c_str()); }
          else if (sk == "addpos") { s.kind = S_ADDPOS; s.n0 = args[0]; }
          else return false;
          op.ep.push_back(s);
        }
      } else {
        op.kv[k] = v;
      }
    }
    net.ops.push_back(op);
  }
  return net.ntensors > 0;
}

const Param* getp(Net& net, const std::string& name) {
  auto it = net.params.find(name);
  if (it == net.params.end()) {
    net.err = "missing parameter " + name;
    return nullptr;
  }
  return &it->second;
}


// ---- LAB fold (the round-5 arithmetic contract; the product's loader does the same, csrc/net.hip fold_lab) ----
// A conv / depthwise conv whose stage list is  bias b | smul s0 | sadd a0  [| act hswish | smul s1 | sadd a1]:
//   w'   = (float)((double)w * s0 * s_in)                      every weight, one rounding
//   b'_o = (float)(s0 * (b_o + a_in * sum_k w_ok) + a0)        in double, k ascending, one rounding
//   stages: bias b' [| act hsw6 (u = y * clamp(y + 3, 0, 6)) [| sfma s6 = (float)(s1 / 6.0), a1  (fmaf(u, s6, a1))]]
// (s_in, a_in) = (1, 0), or - ABSORPTION - the (s6, a1) of a depthwise conv with the full chain whose output's only reader is
// this op, a 1x1 stride-1 conv that itself carries the chain: the depthwise conv then stops after hsw6 and its tensor holds u.
struct LabChain { bool on = false, act = false; float s0 = 1, a0 = 0, s1 = 1, a1 = 0; };
bool lab_chain(Net& net, const Op& op, LabChain& L) {
  L = LabChain();
  const auto& e = op.ep;
  if (op.kind != "conv" && op.kind != "dw") return true;
  const bool p3 = e.size() >= 3 && e[0].kind == S_BIAS && e[1].kind == S_SMUL && e[2].kind == S_SADD;
  const bool p6 = p3 && e.size() == 6 && e[3].kind == S_ACT && e[3].act == A_HSWISH && e[4].kind == S_SMUL && e[5].kind == S_SADD;
  if (!(p6 || (p3 && e.size() == 3))) return true;
  const Param *s0 = getp(net, e[1].n0), *a0 = getp(net, e[2].n0);
  if (!s0 || !a0) return false;
  L.on = true; L.s0 = s0->d[0]; L.a0 = a0->d[0];
  if (p6) {
    const Param *s1 = getp(net, e[4].n0), *a1 = getp(net, e[5].n0);
    if (!s1 || !a1) return false;
    L.act = true; L.s1 = s1->d[0]; L.a1 = a1->d[0];
  }
  return true;
}
inline float lab_s6(float s1) { return (float)((double)s1 / 6.0); }

bool resolve(Net& net) {
  if (net.ops_plan.empty()) net.ops_plan = net.ops;
  net.ops = net.ops_plan;
  net.wt.clear(); net.wdw.clear();
  const size_t nops = net.ops.size();
  // ---- which ops carry the chain, and which depthwise convs hand their (s6, a1) to the 1x1 conv that reads them
  std::vector<LabChain> lab(nops);
  std::vector<int> uses(net.ntensors, 0), absorbs(nops, -1);  // absorbs[conv] = the depthwise op whose affine it takes over
  std::vector<char> handed(nops, 0);
  int out_tid = -1;
  for (size_t oi = 0; oi < nops; ++oi) {
    const Op& op = net.ops[oi];
    if (!lab_chain(net, op, lab[oi])) return false;
    if (op.kind == "output") { out_tid = op.geti("i"); continue; }
    if (op.kind == "concat") { for (int t : op.getlist("i")) uses[t]++; }
    else if (op.kv.count("i")) uses[op.geti("i")]++;
    for (const Stage& s : op.ep) if (s.tid >= 0) uses[s.tid]++;
  }
  for (size_t oi = 0; oi < nops; ++oi) {
    const Op& d = net.ops[oi];
    if (d.kind != "dw" || !lab[oi].act || d.geti("o") == out_tid || uses[d.geti("o")] != 1) continue;
    for (size_t oj = oi + 1; oj < nops; ++oj) {
      const Op& c = net.ops[oj];
      if (c.kind == "concat" || c.kind == "output" || !c.kv.count("i") || c.geti("i") != d.geti("o")) continue;
      if (c.kind == "conv" && lab[oj].on && c.geti("kh") == 1 && c.geti("kw") == 1 && c.geti("sh") == 1 && c.geti("sw") == 1 &&
          c.geti("ph") == 0 && c.geti("pw") == 0) { absorbs[oj] = (int)oi; handed[oi] = 1; }
      break;
    }
  }
  for (size_t oi = 0; oi < nops; ++oi) {
    Op& op = net.ops[oi];
    if (lab[oi].on) {  // rewrite the stage list; the folded bias is resolved here, the folded weights below
      const LabChain& L = lab[oi];
      const Param* b = getp(net, op.ep[0].n0); if (!b) return false;
      const Param* w = getp(net, op.kv["w"]); if (!w) return false;
      const size_t C = b->d.size();
      const size_t per = w->d.size() / C;  // weights per output channel (conv: ci*kh*kw, o-major; dw: kh*kw)
      double s_in = 1.0, a_in = 0.0;
      if (absorbs[oi] >= 0) { s_in = (double)lab_s6(lab[absorbs[oi]].s1); a_in = (double)lab[absorbs[oi]].a1; }
      Stage sb; sb.kind = S_BIAS; sb.v0.resize(C);
      for (size_t o = 0; o < C; ++o) {
        double sum = 0.0;
        if (absorbs[oi] >= 0) for (size_t k = 0; k < per; ++k) sum = sum + (double)w->d[o * per + k];
        const double t = a_in * sum;
        const double u = (double)b->d[o] + t;
        const double v = (double)L.s0 * u;
        sb.v0[o] = (float)(v + (double)L.a0);
      }
      std::vector<Stage> ne;
      ne.push_back(sb);
      if (L.act) {
        Stage sa; sa.kind = S_ACT; sa.act = A_HSW6; ne.push_back(sa);
        if (!handed[oi]) { Stage sf; sf.kind = S_SFMA; sf.p0 = lab_s6(L.s1); sf.p1 = L.a1; ne.push_back(sf); }
      }
      op.ep = ne;
      if (op.kind == "dw") {
        std::vector<float>& wd = net.wdw[(int)oi];
        wd.resize(w->d.size());
        for (size_t i = 0; i < wd.size(); ++i) { const double t = (double)w->d[i] * (double)L.s0; wd[i] = (float)(t * s_in); }
      }
    }
    const double w_s0 = lab[oi].on ? (double)lab[oi].s0 : 1.0;
    const double w_sin = absorbs[oi] >= 0 ? (double)lab_s6(lab[absorbs[oi]].s1) : 1.0;
    const bool w_fold = lab[oi].on;
    for (Stage& s : op.ep) {
      if (s.kind == S_BIAS && !s.v0.empty()) continue;  // (the folded bias above)
      if (s.kind == S_BIAS || s.kind == S_ADDPOS) {
        auto p = getp(net, s.n0); if (!p) return false;
        s.v0 = p->d;
      } else if (s.kind == S_SMUL || s.kind == S_SADD) {
        auto p = getp(net, s.n0); if (!p) return false;
        s.p0 = p->d[0];
      } else if (s.kind == S_BN) {
        auto g = getp(net, s.n0), b = getp(net, s.n1), m = getp(net, s.n2), v = getp(net, s.n3);
        if (!g || !b || !m || !v) return false;
        size_t C = g->d.size();
        s.v0.resize(C); s.v1.resize(C);
        for (size_t c = 0; c < C; ++c) {
          float inv = 1.0f / sqrtf(v->d[c] + s.p0);
          float sc = g->d[c] * inv;
          float mi = m->d[c] * inv;
          float ms = mi * g->d[c];
          s.v0[c] = sc;
          s.v1[c] = b->d[c] - ms;
        }
      }
    }
    if (op.kind == "conv") {
      auto p = getp(net, op.kv["w"]); if (!p) return false;
      int co = p->dims[0], ci = p->dims[1], kh = p->dims[2], kw = p->dims[3];
      std::vector<float>& w = net.wt[(int)oi];
      w.resize((size_t)kh * kw * ci * co);
      for (int o = 0; o < co; ++o)
        for (int c = 0; c < ci; ++c)
          for (int y = 0; y < kh; ++y)
            for (int x = 0; x < kw; ++x) {
              float v = p->d[(((size_t)o * ci + c) * kh + y) * kw + x];
              if (w_fold) { const double t = (double)v * w_s0; v = (float)(t * w_sin); }
              w[((size_t)(y * kw + x) * ci + c) * co + o] = v;
            }
    }
  }
  net.resolved = true;
  return true;
}

inline float act_apply(const Stage& s, float y) {
  switch (s.act) {
    case A_RELU: return fmaxf(y, 0.0f);
    case A_HSWISH: { float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f); float u = y * t; return u / 6.0f; }
    case A_HSIG: { float t = y * s.p0; t = t + s.p1; return fminf(fmaxf(t, 0.0f), 1.0f); }
    case A_SWISH: { float e = ocr_expf(-y); float d = 1.0f + e; return y / d; }
    case A_SIGMOID: { float e = ocr_expf(-y); float d = 1.0f + e; return 1.0f / d; }
    case A_GELU: { const float hx = 0.5f * y; const float z = y * 0.70710678118654752f; const float e1 = 1.0f + ocr_erff(z); return hx * e1; }
    case A_HSW6: { float t = fminf(fmaxf(y + 3.0f, 0.0f), 6.0f); return y * t; }  // the hard-swish's product; its 1/6 is folded (fold_lab)
  }
  return y;
}

// apply the epilogue to one output pixel (all channels)
inline void epilogue(Net& net, const Op& op, float* y, int C, int n, int h, int w) {
  for (const Stage& s : op.ep) {
    switch (s.kind) {
      case S_BIAS: for (int c = 0; c < C; ++c) y[c] = y[c] + s.v0[c]; break;
      case S_SMUL: for (int c = 0; c < C; ++c) y[c] = s.p0 * y[c]; break;
      case S_SADD: for (int c = 0; c < C; ++c) y[c] = y[c] + s.p0; break;
      case S_SFMA: for (int c = 0; c < C; ++c) y[c] = fmaf(y[c], s.p0, s.p1); break;
      case S_BN: for (int c = 0; c < C; ++c) { float t = y[c] * s.v0[c]; y[c] = t + s.v1[c]; } break;
      case S_ACT: for (int c = 0; c < C; ++c) y[c] = act_apply(s, y[c]); break;
      case S_MULC: { const float* g = net.t[s.tid].at(n, 0, 0); for (int c = 0; c < C; ++c) y[c] = y[c] * g[c]; } break;
      case S_ADDT: { const float* g = net.t[s.tid].at(n, h, w); for (int c = 0; c < C; ++c) y[c] = y[c] + g[c]; } break;
      case S_ADDUP: { const float* g = net.t[s.tid].at(n, h / s.up, w / s.up); for (int c = 0; c < C; ++c) y[c] = y[c] + g[c]; } break;
      case S_ADDPOS: { const float* g = s.v0.data() + ((size_t)h * net.t[op.geti("o")].w + w) * C; for (int c = 0; c < C; ++c) y[c] = y[c] + g[c]; } break;
    }
  }
}

bool run(Net& net, const float* x_nhwc, int N, int H, int W) {
  if (!net.resolved && !resolve(net)) return false;
  net.t.assign(net.ntensors, Tensor());
  net.t[0].alloc(N, H, W, 3);
  memcpy(net.t[0].d.data(), x_nhwc, sizeof(float) * (size_t)N * H * W * 3);
  for (size_t oi = 0; oi < net.ops.size(); ++oi) {
    const Op& op = net.ops[oi];
    const std::string& k = op.kind;
    if (k == "output") continue;
    int o = op.geti("o");
    if (k == "conv") {
      const Tensor& in = net.t[op.geti("i")];
      int ci = op.geti("cin"), co = op.geti("cout"), kh = op.geti("kh"), kw = op.geti("kw");
      int sh = op.geti("sh"), sw = op.geti("sw"), ph = op.geti("ph"), pw = op.geti("pw");
      int oh = (in.h + 2 * ph - kh) / sh + 1, ow = (in.w + 2 * pw - kw) / sw + 1;
      Tensor& out = net.t[o];
      out.alloc(in.n, oh, ow, co);
      const float* wt = net.wt[(int)oi].data();
#pragma omp parallel for collapse(2) schedule(static)
      for (int n = 0; n < in.n; ++n)
        for (int y = 0; y < oh; ++y) {
          std::vector<float> acc(co);
          for (int x = 0; x < ow; ++x) {
            for (int c = 0; c < co; ++c) acc[c] = 0.f;
            for (int ky = 0; ky < kh; ++ky)
              for (int kx = 0; kx < kw; ++kx) {
                int iy = y * sh - ph + ky, ix = x * sw - pw + kx;
                bool inb = iy >= 0 && iy < in.h && ix >= 0 && ix < in.w;
                const float* src = inb ? in.at(n, iy, ix) : nullptr;
                const float* wk = wt + (size_t)(ky * kw + kx) * ci * co;
                for (int c = 0; c < ci; ++c) {
                  float a = inb ? src[c] : 0.0f;
                  const float* wr = wk + (size_t)c * co;
#pragma omp simd
                  for (int q = 0; q < co; ++q) acc[q] = fmaf(a, wr[q], acc[q]);
                }
              }
            float* dst = out.at(n, y, x);
            for (int c = 0; c < co; ++c) dst[c] = acc[c];
            epilogue(net, op, dst, co, n, y, x);
          }
        }
    } else if (k == "linear") {
      const Tensor& in = net.t[op.geti("i")];
      int ci = op.geti("cin"), co = op.geti("cout");
      auto p = getp(net, op.kv.at("w")); if (!p) return false;
      const float* wt = p->d.data();  // [in][out]
      Tensor& out = net.t[o];
      out.alloc(in.n, in.h, in.w, co);
      size_t rows = (size_t)in.n * in.h * in.w;
#pragma omp parallel for schedule(static)
      for (long r = 0; r < (long)rows; ++r) {
        const float* src = in.d.data() + (size_t)r * ci;
        float* dst = out.d.data() + (size_t)r * co;
        for (int q = 0; q < co; ++q) dst[q] = 0.f;
        for (int c = 0; c < ci; ++c) {
          float a = src[c];
          const float* wr = wt + (size_t)c * co;
#pragma omp simd
          for (int q = 0; q < co; ++q) dst[q] = fmaf(a, wr[q], dst[q]);
        }
        int n = (int)(r / ((size_t)in.h * in.w));
        int hw = (int)(r % ((size_t)in.h * in.w));
        epilogue(net, op, dst, co, n, hw / in.w, hw % in.w);
      }
    } else if (k == "dw") {
      const Tensor& in = net.t[op.geti("i")];
      int C = op.geti("c"), kh = op.geti("kh"), kw = op.geti("kw");
      int sh = op.geti("sh"), sw = op.geti("sw"), ph = op.geti("ph"), pw = op.geti("pw");
      int oh = (in.h + 2 * ph - kh) / sh + 1, ow = (in.w + 2 * pw - kw) / sw + 1;
      auto p = getp(net, op.kv.at("w")); if (!p) return false;  // [C,1,kh,kw]
      const float* pw_ = net.wdw.count((int)oi) ? net.wdw[(int)oi].data() : p->d.data();  // (after the LAB fold)
      std::vector<float> wt((size_t)kh * kw * C);
      for (int c = 0; c < C; ++c)
        for (int t = 0; t < kh * kw; ++t) wt[(size_t)t * C + c] = pw_[(size_t)c * kh * kw + t];
      Tensor& out = net.t[o];
      out.alloc(in.n, oh, ow, C);
#pragma omp parallel for collapse(2) schedule(static)
      for (int n = 0; n < in.n; ++n)
        for (int y = 0; y < oh; ++y)
          for (int x = 0; x < ow; ++x) {
            float* dst = out.at(n, y, x);
            for (int c = 0; c < C; ++c) dst[c] = 0.f;
            for (int ky = 0; ky < kh; ++ky)
              for (int kx = 0; kx < kw; ++kx) {
                int iy = y * sh - ph + ky, ix = x * sw - pw + kx;
                bool inb = iy >= 0 && iy < in.h && ix >= 0 && ix < in.w;
                const float* src = inb ? in.at(n, iy, ix) : nullptr;
                const float* wr = wt.data() + (size_t)(ky * kw + kx) * C;
#pragma omp simd
                for (int c = 0; c < C; ++c) dst[c] = fmaf(inb ? src[c] : 0.0f, wr[c], dst[c]);
              }
            epilogue(net, op, dst, C, n, y, x);
          }
    } else if (k == "deconv") {
      const Tensor& in = net.t[op.geti("i")];
      int ci = op.geti("cin"), co = op.geti("cout");
      auto p = getp(net, op.kv.at("w")); if (!p) return false;  // [cin,cout,2,2]
      Tensor& out = net.t[o];
      out.alloc(in.n, in.h * 2, in.w * 2, co);
#pragma omp parallel for collapse(2) schedule(static)
      for (int n = 0; n < in.n; ++n)
        for (int y = 0; y < in.h; ++y)
          for (int x = 0; x < in.w; ++x) {
            const float* src = in.at(n, y, x);
            for (int dy = 0; dy < 2; ++dy)
              for (int dx = 0; dx < 2; ++dx) {
                float* dst = out.at(n, 2 * y + dy, 2 * x + dx);
                for (int q = 0; q < co; ++q) {
                  float acc = 0.f;
                  for (int c = 0; c < ci; ++c) acc = fmaf(src[c], p->d[(((size_t)c * co + q) * 2 + dy) * 2 + dx], acc);
                  dst[q] = acc;
                }
                epilogue(net, op, dst, co, n, 2 * y + dy, 2 * x + dx);
              }
          }
    } else if (k == "gap") {
      const Tensor& in = net.t[op.geti("i")];
      Tensor& out = net.t[o];
      out.alloc(in.n, 1, 1, in.c);
      float cnt = (float)(in.h * in.w);
      for (int n = 0; n < in.n; ++n)
        for (int c = 0; c < in.c; ++c) {
          float tot = 0.f;
          for (int y = 0; y < in.h; ++y) {
            float rs = 0.f;
            for (int x = 0; x < in.w; ++x) rs = rs + in.at(n, y, x)[c];
            tot = tot + rs;
          }
          out.at(n, 0, 0)[c] = tot / cnt;
        }
    } else if (k == "sefc") {
      const Tensor& in = net.t[op.geti("i")];
      int C = op.geti("c"), R = op.geti("cr");
      auto w1 = getp(net, op.kv.at("w1")), b1 = getp(net, op.kv.at("b1"));
      auto w2 = getp(net, op.kv.at("w2")), b2 = getp(net, op.kv.at("b2"));
      if (!w1 || !b1 || !w2 || !b2) return false;
      float slope = op.getf("slope"), offset = op.getf("offset");
      Tensor& out = net.t[o];
      out.alloc(in.n, 1, 1, C);
      for (int n = 0; n < in.n; ++n) {
        const float* m = in.at(n, 0, 0);
        std::vector<float> hbuf(R);
        for (int j = 0; j < R; ++j) {
          float acc = 0.f;
          for (int c = 0; c < C; ++c) acc = fmaf(m[c], w1->d[(size_t)j * C + c], acc);
          acc = acc + b1->d[j];
          hbuf[j] = fmaxf(acc, 0.0f);
        }
        float* g = out.at(n, 0, 0);
        for (int c = 0; c < C; ++c) {
          float acc = 0.f;
          for (int j = 0; j < R; ++j) acc = fmaf(hbuf[j], w2->d[(size_t)c * R + j], acc);
          acc = acc + b2->d[c];
          float t = acc * slope;
          t = t + offset;
          g[c] = fminf(fmaxf(t, 0.0f), 1.0f);
        }
      }
    } else if (k == "ew") {
      const Tensor& in = net.t[op.geti("i")];
      Tensor& out = net.t[o];
      out.alloc(in.n, in.h, in.w, in.c);
#pragma omp parallel for collapse(2) schedule(static)
      for (int n = 0; n < in.n; ++n)
        for (int y = 0; y < in.h; ++y)
          for (int x = 0; x < in.w; ++x) {
            float* dst = out.at(n, y, x);
            memcpy(dst, in.at(n, y, x), sizeof(float) * in.c);
            epilogue(net, op, dst, in.c, n, y, x);
          }
    } else if (k == "concat") {
      auto ids = op.getlist("i"), ups = op.getlist("up");
      const Tensor& last = net.t[ids.back()];
      int oh = last.h * ups.back(), ow = last.w * ups.back();
      int C = op.geti("c");
      Tensor& out = net.t[o];
      out.alloc(last.n, oh, ow, C);
      for (int n = 0; n < out.n; ++n)
        for (int y = 0; y < oh; ++y)
          for (int x = 0; x < ow; ++x) {
            float* dst = out.at(n, y, x);
            int off = 0;
            for (size_t j = 0; j < ids.size(); ++j) {
              const Tensor& s = net.t[ids[j]];
              memcpy(dst + off, s.at(n, y / ups[j], x / ups[j]), sizeof(float) * s.c);
              off += s.c;
            }
          }
    } else if (k == "pool") {
      const Tensor& in = net.t[op.geti("i")];
      int kh = op.geti("kh"), kw = op.geti("kw"), sh = op.geti("sh"), sw = op.geti("sw");
      const int ph = op.geti("ph", 0), pw = op.geti("pw", 0);  // (server plans: the ResNet stem's 3x3 s2 p1 max pool)
      bool is_max = op.kv.at("type") == "max";
      int oh = (in.h + 2 * ph - kh) / sh + 1, ow = (in.w + 2 * pw - kw) / sw + 1;  // C++ truncating division (SURVEY §A.2 note)
      Tensor& out = net.t[o];
      out.alloc(in.n, oh, ow, in.c);
#pragma omp parallel for collapse(2) schedule(static)
      for (int n = 0; n < in.n; ++n)
        for (int y = 0; y < oh; ++y)
          for (int x = 0; x < ow; ++x)
            for (int c = 0; c < in.c; ++c) {
              float acc = is_max ? -INFINITY : 0.f;
              int cnt = 0;
              for (int dy = 0; dy < kh; ++dy)
                for (int dx = 0; dx < kw; ++dx) {
                  int iy = y * sh - ph + dy, ix = x * sw - pw + dx;
                  if (iy < 0 || ix < 0 || iy >= in.h || ix >= in.w) continue;  // outside: takes no part (exclusive mean, -inf pad)
                  float v = in.at(n, iy, ix)[c];
                  acc = is_max ? fmaxf(acc, v) : acc + v;
                  ++cnt;
                }
              out.at(n, y, x)[c] = is_max ? acc : acc / (float)cnt;
            }
    } else if (k == "ln") {
      const Tensor& in = net.t[op.geti("i")];
      int C = in.c;
      float eps = op.getf("eps");
      auto g = getp(net, op.kv.at("g")), b = getp(net, op.kv.at("b"));
      if (!g || !b) return false;
      Tensor& out = net.t[o];
      out.alloc(in.n, in.h, in.w, C);
      size_t rows = (size_t)in.n * in.h * in.w;
#pragma omp parallel for schedule(static)
      for (long r = 0; r < (long)rows; ++r) {
        const float* src = in.d.data() + r * C;
        float* dst = out.d.data() + r * C;
        float s = 0.f;
        for (int c = 0; c < C; ++c) s = s + src[c];
        float mean = s / (float)C;
        float v = 0.f;
        for (int c = 0; c < C; ++c) { float xm = src[c] - mean; v = fmaf(xm, xm, v); }
        float var = v / (float)C;
        float rstd = 1.0f / sqrtf(var + eps);
        for (int c = 0; c < C; ++c) {
          float xm = src[c] - mean;
          float t = xm * rstd;
          t = t * g->d[c];
          dst[c] = t + b->d[c];
        }
      }
    } else if (k == "attn") {
      const Tensor& in = net.t[op.geti("i")];
      int heads = op.geti("heads"), hd = op.geti("hd");
      float scale = op.getf("scale");
      int D = heads * hd;
      Tensor& out = net.t[o];
      out.alloc(in.n, in.h, in.w, D);
      int T = in.h * in.w;
      // token grid and local window (server plans, SVTR's Local mixer): key (ky, kx) takes part in query (qy, qx)'s softmax iff
      // |ky - qy| <= lh / 2 and |kx - qx| <= lw / 2 (rec_svtrnet.py adds -inf elsewhere); lh = 0: every key (Global mixer)
      const int gw = op.geti("gw", in.w), lh = op.geti("lh", 0), lw = op.geti("lw", 0);
#pragma omp parallel for collapse(2) schedule(static)
      for (int n = 0; n < in.n; ++n)
        for (int hh = 0; hh < heads; ++hh) {
          std::vector<float> sc(T), q(hd);
          std::vector<char> ok(T);
          const float* base = in.d.data() + (size_t)n * T * 3 * D;
          for (int t = 0; t < T; ++t) {
            const float* qr = base + (size_t)t * 3 * D + hh * hd;
            for (int d = 0; d < hd; ++d) q[d] = qr[d] * scale;
            float m = -INFINITY;
            for (int u = 0; u < T; ++u) {
              ok[u] = 1;
              if (lh > 0) {
                const int dy = u / gw - t / gw, dx = u % gw - t % gw;
                ok[u] = dy >= -(lh / 2) && dy <= lh / 2 && dx >= -(lw / 2) && dx <= lw / 2;
              }
              if (!ok[u]) continue;
              const float* kr = base + (size_t)u * 3 * D + D + hh * hd;
              float acc = 0.f;
              for (int d = 0; d < hd; ++d) acc = fmaf(q[d], kr[d], acc);
              sc[u] = acc;
              m = fmaxf(m, acc);
            }
            float sum = 0.f;
            for (int u = 0; u < T; ++u) if (ok[u]) { sc[u] = ocr_expf(sc[u] - m); sum = sum + sc[u]; }
            for (int u = 0; u < T; ++u) if (ok[u]) sc[u] = sc[u] / sum;
            float* dst = out.d.data() + ((size_t)n * T + t) * D + hh * hd;
            for (int d = 0; d < hd; ++d) {
              float acc = 0.f;
              for (int u = 0; u < T; ++u) if (ok[u]) acc = fmaf(sc[u], base[(size_t)u * 3 * D + 2 * D + hh * hd + d], acc);
              dst[d] = acc;
            }
          }
        }
    } else if (k == "softmax") {
      const Tensor& in = net.t[op.geti("i")];
      int C = in.c;
      Tensor& out = net.t[o];
      out.alloc(in.n, in.h, in.w, C);
      size_t rows = (size_t)in.n * in.h * in.w;
#pragma omp parallel for schedule(static)
      for (long r = 0; r < (long)rows; ++r) {
        const float* src = in.d.data() + (size_t)r * C;
        float* dst = out.d.data() + (size_t)r * C;
        // canonical order: groups of 128 columns; per group its max m_g and
        // s_g = chain0 + chain1, chain h = columns with ((c >> 2) & 1) == h in ascending order of
        // exp(x_c - m_g); row: M = max m_g, S = sum over ascending g of s_g * exp(m_g - M); p = exp(x - M) / S
        const int G = (C + 127) / 128;
        std::vector<float> gm(G), gs(G);
        for (int g = 0; g < G; ++g) {
          const int c0 = g * 128, c1 = std::min(c0 + 128, C);
          float m = -INFINITY;
          for (int c = c0; c < c1; ++c) m = fmaxf(m, src[c]);
          float ch[2] = {0.f, 0.f};
          for (int c = c0; c < c1; ++c) ch[(c >> 2) & 1] = ch[(c >> 2) & 1] + ocr_expf(src[c] - m);
          gm[g] = m;
          gs[g] = ch[0] + ch[1];
        }
        float M = -INFINITY;
        for (int g = 0; g < G; ++g) M = fmaxf(M, gm[g]);
        float S = 0.f;
        for (int g = 0; g < G; ++g) {
          const float t = gs[g] * ocr_expf(gm[g] - M);
          S = S + t;
        }
        for (int c = 0; c < C; ++c) dst[c] = ocr_expf(src[c] - M) / S;
      }
    } else {
      net.err = "unknown plan op " + k;
      return false;
    }
  }
  return true;
}

}  // namespace

extern "C" {

void* oracle_net_create(const char* plan_text) {
  Net* n = new Net();
  if (!parse_plan(*n, plan_text)) { delete n; return nullptr; }
  return n;
}
void oracle_net_destroy(void* h) { delete (Net*)h; }
void oracle_net_set_param(void* h, const char* name, const float* data, int ndims, const int* dims) {
  Net* n = (Net*)h;
  Param p;
  size_t cnt = 1;
  for (int i = 0; i < ndims; ++i) { p.dims.push_back(dims[i]); cnt *= dims[i]; }
  p.d.assign(data, data + cnt);
  n->params[name] = std::move(p);
  n->resolved = false;
}
// x: NHWC f32 [N,H,W,3].  returns 0 ok.
int oracle_net_run(void* h, const float* x, int N, int H, int W) {
  Net* n = (Net*)h;
  return run(*n, x, N, H, W) ? 0 : 1;
}
const char* oracle_net_error(void* h) { return ((Net*)h)->err.c_str(); }
// tensor tap (NHWC); returns element count, fills dims[4]
long oracle_net_tensor(void* h, int tid, int* dims, const float** data) {
  Net* n = (Net*)h;
  if (tid < 0) {  // output
    for (auto& op : n->ops) if (op.kind == "output") tid = op.geti("i");
  }
  Tensor& t = n->t[tid];
  dims[0] = t.n; dims[1] = t.h; dims[2] = t.w; dims[3] = t.c;
  *data = t.d.data();
  return (long)t.d.size();
}
float oracle_expf(float x) { return ocr_expf(x); }
float oracle_erff(float x) { return ocr_erff(x); }

}  // extern "C"
