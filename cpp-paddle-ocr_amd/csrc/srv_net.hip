// Plan executor of the server networks: see srv_net.h.  Host code only (the kernels are srv_kernels.hip).
#include "srv_net.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <mutex>
#include <sstream>

#include "hip_guard.h"

namespace ocr {

namespace {
std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> r;
  std::stringstream ss(s);
  std::string t;
  while (std::getline(ss, t, sep)) r.push_back(t);
  return r;
}
inline int up8(int c) { return (c + 7) & ~7; }

// Weight image of srv_gemm_kernel: [K tile][row n, Npad of them][8 granules] with granule g of row n at slot g ^ ((n >> 1) & 7) -
// the LDS image of a tile is a straight copy of 128 bytes per row, and a fragment read finds (row, granule) conflict-free.
template <typename T>
std::vector<T> weight_image(const std::vector<float>& wnk, int ncols, int K, int npad, int& nkt) {
  constexpr int KG = 16 / (int)sizeof(T), BK = 8 * KG;
  nkt = (K + BK - 1) / BK;
  std::vector<T> img((size_t)nkt * npad * BK, (T)0.f);
  for (int n = 0; n < ncols; ++n)
    for (int k = 0; k < K; ++k) {
      const int kt = k / BK, kk = k % BK, g = kk / KG, e = kk % KG;
      const int slot = g ^ ((n >> 1) & 7);
      img[(((size_t)kt * npad + n) * 8 + slot) * KG + e] = (T)wnk[(size_t)n * K + k];
    }
  return img;
}
}  // namespace

SrvNet::~SrvNet() {
  for (void* p : dev_allocs_) (void)g_free(p);
  if (arena_) (void)g_free(arena_);
  for (auto& p : ev_pending_) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
}

void* SrvNet::upload_bytes(const void* p, size_t n) {
  void* d = nullptr;
  if (g_malloc(&d, std::max<size_t>(n, 16)) != hipSuccess) return nullptr;
  if (n && g_memcpy(d, p, n, hipMemcpyHostToDevice) != hipSuccess) { (void)g_free(d); return nullptr; }
  dev_allocs_.push_back(d);
  return d;
}
const float* SrvNet::upload_f32(const std::vector<float>& v) { return (const float*)upload_bytes(v.data(), v.size() * sizeof(float)); }

bool SrvNet::parse(const char* text, std::string& err) {
  std::stringstream ss(text);
  std::string line;
  while (std::getline(ss, line)) {
    if (line.empty() || line[0] == '#') continue;
    auto toks = split(line, ' ');
    if (toks[0] == "plan") {
      for (auto& t : toks)
        if (t.rfind("ntensors=", 0) == 0) ntensors_ = atoi(t.c_str() + 9);
      continue;
    }
    Op op;
    op.kind = toks[0];
    static const char* kinds[] = {"conv", "linear", "deconv", "pool", "concat", "ew", "ln", "attn", "output"};
    if (std::find_if(std::begin(kinds), std::end(kinds), [&](const char* k) { return op.kind == k; }) == std::end(kinds)) {
      err = "server plan: unknown op " + op.kind;
      return false;
    }
    for (size_t i = 1; i < toks.size(); ++i) {
      const auto eq = toks[i].find('=');
      if (eq == std::string::npos) { err = "server plan: bad token " + toks[i]; return false; }
      const std::string k = toks[i].substr(0, eq), v = toks[i].substr(eq + 1);
      if (k == "ep") {
        for (auto& st : split(v, '|')) {
          const auto c = st.find(':');
          Stage s;
          s.kind = st.substr(0, c);
          auto args = split(st.substr(c + 1), ',');
          if (s.kind == "bias" || s.kind == "addpos") s.a0 = args[0];
          else if (s.kind == "bn") { s.a0 = args[0]; s.a1 = args[1]; s.a2 = args[2]; s.a3 = args[3]; s.f = strtof(args[4].c_str(), nullptr); }
          else if (s.kind == "act") s.a0 = args[0];
          else if (s.kind == "addt") s.tid = atoi(args[0].c_str());
          else if (s.kind == "addup") { s.tid = atoi(args[0].c_str()); s.up = atoi(args[1].c_str()); }
          else { err = "server plan: unknown stage " + s.kind; return false; }
          op.ep.push_back(s);
        }
      } else if (k == "i" && op.kind == "concat") {
        for (auto& t : split(v, ',')) op.ins.push_back(atoi(t.c_str()));
      } else if (k == "up") {
        for (auto& t : split(v, ',')) op.ups.push_back(atoi(t.c_str()));
      } else {
        op.kv[k] = v;
      }
    }
    if (op.kind == "output") out_tid_ = op.geti("i");
    ops_.push_back(op);
  }
  if (ntensors_ <= 0 || out_tid_ < 0) { err = "server plan: missing header or output"; return false; }
  return true;
}

// Resolves an op's parameters into device images.  The epilogue a GEMM launch can carry is, in this order, each part optional:
// bias | batch norm (scale, shift: the operations of the oracle's parameter resolution) | addt or addup | act.
bool SrvNet::prepare_op(Op& op, const WeightMap& W, std::string& err) {
  auto need = [&](const std::string& n) -> const HostTensor* {
    auto it = W.find(n);
    if (it == W.end()) { err = "missing parameter " + n; return nullptr; }
    return &it->second;
  };
  const bool gemm = op.kind == "conv" || op.kind == "linear" || op.kind == "deconv";
  if (gemm) {
    const HostTensor* w = need(op.kv["w"]);
    if (!w) return false;
    const int cin = op.geti("cin"), cout = op.geti("cout");
    const int cin_s = up8(cin);
    op.cin_s = cin_s;
    int K = 0;
    std::vector<float> wnk;
    if (op.kind == "conv") {
      const int kh = op.geti("kh"), kw = op.geti("kw");
      if ((int)w->numel() != cout * cin * kh * kw) { err = "parameter " + op.kv["w"] + " has the wrong size"; return false; }
      K = kh * kw * cin_s;
      op.ncols = cout;
      wnk.assign((size_t)cout * K, 0.f);
      for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c)
          for (int y = 0; y < kh; ++y)
            for (int x = 0; x < kw; ++x) wnk[(size_t)o * K + (size_t)(y * kw + x) * cin_s + c] = w->data[(((size_t)o * cin + c) * kh + y) * kw + x];
    } else if (op.kind == "linear") {
      if ((int)w->numel() != cin * cout) { err = "parameter " + op.kv["w"] + " has the wrong size"; return false; }
      K = cin_s;
      op.ncols = cout;
      wnk.assign((size_t)cout * K, 0.f);
      for (int c = 0; c < cin; ++c)
        for (int o = 0; o < cout; ++o) wnk[(size_t)o * K + c] = w->data[(size_t)c * cout + o];
    } else {  // deconv 2x2 s2: [Cin][Cout][2][2]
      if ((int)w->numel() != cin * cout * 4) { err = "parameter " + op.kv["w"] + " has the wrong size"; return false; }
      K = cin_s;
      if (cout == 1) {  // the DB head's last layer: deconv_map_kernel, weights [4 taps][Cs]
        std::vector<float> w4((size_t)4 * cin_s, 0.f);
        for (int c = 0; c < cin; ++c)
          for (int t = 0; t < 4; ++t) w4[(size_t)t * cin_s + c] = w->data[(size_t)c * 4 + t];
        op.p0 = upload_f32(w4);
        if (!op.p0) { err = "device allocation failed"; return false; }
        op.ncols = 0;
      } else {
        if (cout % 8) { err = "deconv: output channels must be a multiple of 8"; return false; }
        op.ncols = 4 * cout;
        wnk.assign((size_t)4 * cout * K, 0.f);
        for (int c = 0; c < cin; ++c)
          for (int o = 0; o < cout; ++o)
            for (int t = 0; t < 4; ++t) wnk[((size_t)t * cout + o) * K + c] = w->data[((size_t)c * cout + o) * 4 + t];
      }
    }
    if (op.ncols > 0) op.npad = (op.ncols + 255) & ~255;
    // ---- epilogue
    int phase = 0;  // bias 1, bn 2, add 3, act 4: strictly increasing
    const int reps = op.kind == "deconv" && cout > 1 ? 4 : 1;  // per-channel vectors repeat over the four taps of a transposed conv
    const int vlen = std::max(op.npad, 8);
    std::vector<float> h_bias, h_scale, h_shift;  // host copies (the f16 build folds them, below)
    for (const Stage& s : op.ep) {
      int ph = 0;
      if (s.kind == "bias") {
        ph = 1;
        const HostTensor* b = need(s.a0);
        if (!b) return false;
        if (cout == 1 && op.kind == "deconv") { op.fbias = b->data[0]; }
        else {
          h_bias.assign(vlen, 0.f);
          for (int t = 0; t < reps; ++t)
            for (int o = 0; o < cout; ++o) h_bias[(size_t)t * cout + o] = b->data[o];
        }
      } else if (s.kind == "bn") {
        ph = 2;
        const HostTensor *g = need(s.a0), *b = need(s.a1), *m = need(s.a2), *vv = need(s.a3);
        if (!g || !b || !m || !vv) return false;
        std::vector<float> sc(vlen, 0.f), sh(vlen, 0.f);
        for (int o = 0; o < cout; ++o) {
          const float inv = 1.0f / sqrtf(vv->data[o] + s.f);
          const float scv = g->data[o] * inv;
          const float mi = m->data[o] * inv;
          const float ms = mi * g->data[o];
          for (int t = 0; t < reps; ++t) { sc[(size_t)t * cout + o] = scv; sh[(size_t)t * cout + o] = b->data[o] - ms; }
        }
        h_scale = sc;
        h_shift = sh;
      } else if (s.kind == "addt") { ph = 3; op.res_tid = s.tid; op.res_up = 1; }
      else if (s.kind == "addup") {
        ph = 3;
        if (s.up != 2) { err = "addup: only a factor of 2 is built"; return false; }
        op.res_tid = s.tid; op.res_up = 2;
      } else if (s.kind == "act") {
        ph = 4;
        op.act = s.a0 == "relu" ? srv::SACT_RELU : s.a0 == "gelu" ? srv::SACT_GELU : s.a0 == "hswish" ? srv::SACT_HSWISH : s.a0 == "sigmoid" ? srv::SACT_SIGMOID : -1;
        if (op.act < 0) { err = "server plan: activation " + s.a0 + " is not built"; return false; }
      } else { err = "server plan: stage " + s.kind + " cannot follow a matrix product"; return false; }
      if (ph <= phase) { err = "server plan: epilogue of " + op.kv["w"] + " is not in the order bias | bn | add | act"; return false; }
      phase = ph;
    }
    if (op.ncols > 0) {
      // precision "fp16": the batch norm's scale goes into the weights (one rounding to f16 either way) and its shift, with the
      // bias, into ONE vector - (acc + b) s + t = acc s + (b s + t) - so the epilogue is a single add.  The f32 twin keeps the
      // oracle's three roundings (bias, scale, shift) and the unscaled weights.
      if (half_ && !h_scale.empty()) {
        const int K_ = (int)(wnk.size() / op.ncols);
        for (int n = 0; n < op.ncols; ++n)
          for (int k = 0; k < K_; ++k) wnk[(size_t)n * K_ + k] *= h_scale[n];
        if (h_bias.empty()) h_bias.assign(vlen, 0.f);
        for (int n = 0; n < op.ncols; ++n) h_bias[n] = h_bias[n] * h_scale[n] + h_shift[n];
        h_scale.clear();
        h_shift.clear();
      }
      // f16 build, k x k convs on whole 64-channel tiles: K order (channel tile, tap, channel in tile) - GemmArgs::korder = 1.
      // Both forms of such a conv (implicit GEMM, halo patch) then accumulate in the same order.  OCR_SRV_KORDER=0: the oracle's order (no halo form)
      static const bool korder_on = [] { const char* e = getenv("OCR_SRV_KORDER"); return !(e && e[0] == '0'); }();
      if (half_ && korder_on && op.kind == "conv" && op.geti("kh") * op.geti("kw") > 1 && cin_s % 64 == 0) {  // (any stride: the implicit-GEMM form gains
        op.korder = 1;                                                                                     // from the locality alone; the halo form takes stride 1)
        const int taps = op.geti("kh") * op.geti("kw");
        std::vector<float> perm(wnk.size());
        for (int n = 0; n < op.ncols; ++n)
          for (int tap = 0; tap < taps; ++tap)
            for (int c = 0; c < cin_s; ++c)
              perm[(size_t)n * K + (size_t)((c / 64) * taps + tap) * 64 + (c % 64)] = wnk[(size_t)n * K + (size_t)tap * cin_s + c];
        wnk.swap(perm);
      }
      int nkt = 0;
      if (half_) {
        auto img = weight_image<_Float16>(wnk, op.ncols, K, op.npad, nkt);
        op.wimg_bytes = img.size() * sizeof(_Float16);
        op.wimg = upload_bytes(img.data(), op.wimg_bytes);
      } else {
        auto img = weight_image<float>(wnk, op.ncols, K, op.npad, nkt);
        op.wimg_bytes = img.size() * sizeof(float);
        op.wimg = upload_bytes(img.data(), op.wimg_bytes);
      }
      if (!op.wimg) { err = "device allocation failed"; return false; }
    }
    if (!h_bias.empty()) op.bias = upload_f32(h_bias);
    if (!h_scale.empty()) { op.scale = upload_f32(h_scale); op.shift = upload_f32(h_shift); }
    if (op.kind == "deconv" && cout == 1 && (op.scale || op.res_tid >= 0 || op.act != srv::SACT_SIGMOID)) {
      err = "deconv to one channel: only bias | sigmoid is built";
      return false;
    }
    return true;
  }
  if (op.kind == "ln") {
    const HostTensor *g = need(op.kv["g"]), *b = need(op.kv["b"]);
    if (!g || !b) return false;
    op.p0 = upload_f32(g->data);
    op.p1 = upload_f32(b->data);
    return op.p0 && op.p1;
  }
  if (op.kind == "ew") {
    if (op.ep.size() != 1 || op.ep[0].kind != "addpos") { err = "server plan: ew carries only addpos here"; return false; }
    const HostTensor* p = need(op.ep[0].a0);
    if (!p) return false;
    if (op.geti("c") % 8) { err = "addpos: channels must be a multiple of 8"; return false; }
    op.p0 = upload_f32(p->data);
    return op.p0 != nullptr;
  }
  return true;
}

// An `ln` whose result feeds `linear C -> 4C | bias | gelu` (an MLP's fc1): LN(u) W1 + b1 = r (u W1') - r m s + c with W1' = diag(gamma) W1,
// s = the column sums of W1' AS ROUNDED TO f16 (what the matrix instructions multiply), c = beta W1 + b1 (srv_mlp.h MlpArgs::ln_*)
bool SrvNet::prepare_ln_fold(Op& fc1, const Op& ln, const WeightMap& W, std::string& err) {
  auto find = [&](const std::string& n) -> const HostTensor* { auto it = W.find(n); return it == W.end() ? nullptr : &it->second; };
  const HostTensor *w = find(fc1.kv["w"]), *g = find(ln.kv.at("g")), *b = find(ln.kv.at("b")), *b1 = nullptr;
  for (const Stage& st : fc1.ep)
    if (st.kind == "bias") b1 = find(st.a0);
  const int cin = fc1.geti("cin"), cout = fc1.geti("cout");
  if (!w || !g || !b || !b1 || (int)w->numel() != cin * cout || (int)g->numel() != cin || (int)b->numel() != cin || (int)b1->numel() != cout || cin % 8) return true;  // (no fold: the pair runs unfused)
  std::vector<float> wnk((size_t)cout * cin), sv(cout), cv(cout);
  for (int o = 0; o < cout; ++o) {
    double ss = 0, cc = b1->data[o];
    for (int c = 0; c < cin; ++c) {
      const float wv = w->data[(size_t)c * cout + o] * g->data[c];
      wnk[(size_t)o * cin + c] = wv;
      ss += (double)(float)(_Float16)wv;
      cc += (double)w->data[(size_t)c * cout + o] * (double)b->data[c];
    }
    sv[o] = (float)ss;
    cv[o] = (float)cc;
  }
  int nkt = 0;
  auto img = weight_image<_Float16>(wnk, cout, cin, fc1.npad, nkt);
  fc1.wimg_ln_bytes = img.size() * sizeof(_Float16);
  fc1.wimg_ln = upload_bytes(img.data(), fc1.wimg_ln_bytes);
  fc1.ln_s = upload_f32(sv);
  fc1.ln_c = upload_f32(cv);
  if (!fc1.wimg_ln || !fc1.ln_s || !fc1.ln_c) { err = "device allocation failed"; return false; }
  return true;
}

bool SrvNet::load(const char* plan_text, const WeightMap& weights, bool half, std::string& err) {
  half_ = half;
  if (!parse(plan_text, err)) return false;
  for (Op& op : ops_)
    if (!prepare_op(op, weights, err)) return false;
  for (size_t oi = 0; half_ && oi + 1 < ops_.size(); ++oi) {
    const Op& l = ops_[oi];
    Op& f1 = ops_[oi + 1];
    if (l.kind == "ln" && f1.kind == "linear" && f1.act == srv::SACT_GELU && f1.bias && !f1.scale && f1.res_tid < 0 && f1.geti("i") == l.geti("o") &&
        f1.geti("cout") == 4 * f1.geti("cin") && !prepare_ln_fold(f1, l, weights, err)) return false;
  }
  return true;
}

// OCR_SRV_TUNE_FILE=<path>: the tile configurations found on the device are kept in a text file ("<precision> <launch name> <id>"
// per line) and read back by later processes - a service does not re-time its layers at every start, and a profiler run of the
// bench shows the network's launches only (tools/run_profile_cfg5.sh: the tuning launches of one run were 90 % of its trace).
namespace {
std::mutex g_tune_mu;
std::map<std::string, int>& tune_file_cache() {
  static std::map<std::string, int> m;
  static bool loaded = false;
  if (!loaded) {
    loaded = true;
    if (const char* fn = getenv("OCR_SRV_TUNE_FILE")) {
      if (FILE* f = fopen(fn, "r")) {
        char prec[16], name[256];
        int id;
        while (fscanf(f, "%15s %255s %d", prec, name, &id) == 3) m[std::string(prec) + " " + name] = id;
        fclose(f);
      }
    }
  }
  return m;
}
}  // namespace

int SrvNet::tune(const srv::GemmArgs& a, const std::string& key, hipStream_t s) {
  auto it = tuned_.find(key);
  if (it != tuned_.end()) return it->second;
  const std::string fkey = std::string(half_ ? "fp16 " : "fp32 ") + key;
  static const bool do_tune = [] { const char* e = getenv("OCR_SRV_TUNE"); return !(e && e[0] == '0'); }();
  static const int forced = [] { const char* e = getenv("OCR_SRV_CFG"); return e && *e ? atoi(e) : -1; }();
  int best = -1;
  if (forced >= 0 && srv::gemm_config_ok(a, half_, forced)) best = forced;  // (a forced configuration goes before the file's)
  if (best < 0) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto& fc = tune_file_cache();
    auto fi = fc.find(fkey);
    if (fi != fc.end() && srv::gemm_config_ok(a, half_, fi->second)) { tuned_[key] = fi->second; return fi->second; }
  }
  if (best < 0 && !do_tune) {
    // shape heuristic (what the timed choices of the two server networks look like): 256 x 256 tiles where the product is wide and there
    // are enough of them to fill the chip, 128 x 128 for the middle, 128 x 64 for thin layers; forms with one candidate (the halo form
    // behind a folded concat) fall through to the timing loop below
    const long tiles_256 = ((a.M + 255) / 256) * ((a.Ncols + 255) / 256);
    const int pref[] = {a.Ncols <= 64 ? 3 : (a.Ncols >= 512 && tiles_256 >= 256 ? 12 : 1), 1, 3};
    for (int c : pref)
      if (srv::gemm_config_ok(a, half_, c)) { best = c; break; }
  }
  if (best < 0) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::string err;
    auto time_of = [&](int c, int reps) {
      (void)hipEventRecord(e0, s);
      for (int rep = 0; rep < reps; ++rep) (void)srv::launch_gemm(a, half_, c, s, err);
      (void)hipEventRecord(e1, s);
      if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); return 1e30f; }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      return ms / (float)reps;
    };
    // two launches of every candidate, then the three fastest again with six: the best candidates are often within a few per cent of
    // each other, which is also a box's launch-to-launch noise
    std::vector<std::pair<float, int>> cand;
    for (int c = 0; c < srv::gemm_num_configs(); ++c) {
      if (!srv::gemm_config_ok(a, half_, c)) continue;
      if (!srv::launch_gemm(a, half_, c, s, err)) continue;  // warm
      cand.push_back({time_of(c, 2), c});
    }
    std::sort(cand.begin(), cand.end());
    float best_ms = 1e30f;
    for (size_t i = 0; i < cand.size() && i < 3; ++i) {
      const float ms = time_of(cand[i].second, 6);
      if (ms < best_ms) { best_ms = ms; best = cand[i].second; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
  tuned_[key] = best;
  if (best >= 0 && forced < 0 && do_tune) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto& fc = tune_file_cache();
    if (!fc.count(fkey)) {
      fc[fkey] = best;
      if (const char* fn = getenv("OCR_SRV_TUNE_FILE"))
        if (FILE* f = fopen(fn, "a")) { fprintf(f, "%s %d\n", fkey.c_str(), best); fclose(f); }
    }
  }
  return best;
}

bool SrvNet::bind(int N, int H, int W, hipStream_t s, std::string& err) {
  launches_.clear();
  flops_ = bytes_ = 0;
  ctc_slots_ = ctc_step_ = 0;
  const int pk = ntensors_;  // the packed input [N][H][W][8] T: an extra tensor behind the plan's
  tensors_.assign(ntensors_ + 1, SrvTensor());
  auto set = [&](int tid, int n, int h, int w, int c, bool f32 = false, int cs = -1) {
    SrvTensor& t = tensors_[tid];
    t.n = n; t.h = h; t.w = w; t.c = c; t.cs = cs > 0 ? cs : up8(c); t.f32 = f32;
  };
  set(0, N, H, W, 3, true, 3);
  set(pk, N, H, W, 3);
  const size_t esz = half_ ? 2 : 4;
  // ---- shapes
  for (Op& op : ops_) {
    if (op.kind == "output") continue;
    const int o = op.geti("o");
    if (op.kind == "concat") {
      const SrvTensor& last = tensors_[op.ins.back()];
      set(o, last.n, last.h * op.ups.back(), last.w * op.ups.back(), op.geti("c"));
      continue;
    }
    const SrvTensor& in = tensors_[op.geti("i")];
    if (in.n == 0) { err = "server plan: tensor " + std::to_string(op.geti("i")) + " is read before it is written"; return false; }
    if (op.kind == "conv") {
      const int oh = (in.h + 2 * op.geti("ph") - op.geti("kh")) / op.geti("sh") + 1, ow = (in.w + 2 * op.geti("pw") - op.geti("kw")) / op.geti("sw") + 1;
      set(o, in.n, oh, ow, op.geti("cout"));
    } else if (op.kind == "linear") {
      const bool is_out = o == out_tid_;
      set(o, in.n, in.h, in.w, op.geti("cout"), is_out, is_out ? up8(op.geti("cout")) : -1);
    } else if (op.kind == "deconv") {
      const int co = op.geti("cout");
      set(o, in.n, in.h * 2, in.w * 2, co, co == 1, co == 1 ? 1 : -1);
    } else if (op.kind == "pool") {
      const int oh = (in.h + 2 * op.geti("ph") - op.geti("kh")) / op.geti("sh") + 1, ow = (in.w + 2 * op.geti("pw") - op.geti("kw")) / op.geti("sw") + 1;
      set(o, in.n, oh, ow, in.c);
    } else if (op.kind == "attn") {
      set(o, in.n, in.h, in.w, op.geti("heads") * op.geti("hd"));
    } else {
      set(o, in.n, in.h, in.w, in.c);
    }
  }
  // ---- SVTR's MLP pairs as one launch (f16 build, production mode; srv_mlp.h): `linear C -> 4C | gelu` whose only reader is
  // `linear 4C -> C | + the first linear's input`: the hidden tensor then never exists (no arena slot, no launch of its own)
  std::vector<char> mlp_head(ops_.size(), 0), mlp_tail(ops_.size(), 0), ht_head(ops_.size(), 0), ht_tail(ops_.size(), 0), cat_head(ops_.size(), 0), ln_abs(ops_.size(), 0);
  {
    // OCR_SRV_MLP=0: never; =all: every width; default: C <= 256 (measured, tools/micro/srv_mlp_probe + tools/srv_bench.py: 1.46 ms
    // against 1.60 for the two launches at C = 192, 1.22 against 1.20 at C = 256 with 1.5 / 1.0 GB less HBM traffic per launch;
    // at C = 512 the 128 accumulator registers of the 128-token tile spill - 8.2 ms against 1.77 - and the pair stays two launches)
    static const int mlp_max_c = [] { const char* e = getenv("OCR_SRV_MLP"); return e && e[0] == '0' ? 0 : (e && e[0] == 'a' ? 1 << 30 : 256); }();
    const bool mlp_on = mlp_max_c > 0;
    std::vector<int> readers(ntensors_ + 1, 0);
    for (const Op& op : ops_) {
      if (op.kind == "output") { readers[op.geti("i")] += 2; continue; }
      if (op.kind == "concat") for (int t : op.ins) readers[t]++;
      else readers[op.geti("i")]++;
      if (op.res_tid >= 0) readers[op.res_tid]++;
    }
    for (size_t oi = 0; half_ && mlp_on && !keep_all_ && oi + 1 < ops_.size(); ++oi) {
      const Op &f1 = ops_[oi], &f2 = ops_[oi + 1];
      if (f1.kind != "linear" || f2.kind != "linear" || f1.act != srv::SACT_GELU || f1.res_tid >= 0 || !f1.bias || !f2.bias) continue;
      if (f2.geti("i") != f1.geti("o") || readers[f1.geti("o")] != 1 || f2.res_tid != f1.geti("i") || f2.res_up != 1 || f2.act != srv::SACT_NONE) continue;
      const int c = f1.geti("cin");
      if (c > mlp_max_c) continue;
      if (f1.geti("cout") != 4 * c || f2.geti("cin") != 4 * c || f2.geti("cout") != c || f2.geti("o") == out_tid_) continue;
      std::string e;
      if (!srv::launch_mlp(nullptr, 0, nullptr, 0, 0, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, c, nullptr, true, e)) continue;
      mlp_head[oi] = 1;
      mlp_tail[oi + 1] = 1;
    }
    // ---- the LayerNorm IN FRONT of a fused MLP absorbed into it (srv_mlp.h MlpArgs::ln_*): `ln` read by exactly the pair - fc1's input and
    // fc2's residual.  The normalised tensor is then never written and the LayerNorm launch is gone.  OPT-IN (OCR_SRV_MLPLN=1): measured, the
    // kernel pays back most of what the launch cost (-0.3 ms of a 76 ms step) and the arithmetic is no longer LayerNorm-then-MLP - built as the
    // first piece of the LayerNorm-free execution DESIGN.md section 8 sizes, not as a default
    static const bool mlpln_on = [] { const char* e = getenv("OCR_SRV_MLPLN"); return e && e[0] == '1'; }();
    for (size_t oi = 0; half_ && mlpln_on && !keep_all_ && oi + 2 < ops_.size(); ++oi) {
      const Op &l = ops_[oi], &f1 = ops_[oi + 1];
      if (l.kind != "ln" || !mlp_head[oi + 1] || !f1.wimg_ln || f1.geti("i") != l.geti("o") || readers[l.geti("o")] != 2 || l.geti("o") == out_tid_) continue;
      ln_abs[oi] = 1;
    }
    // ---- the DB head's two transposed convs as one launch (f16 build, production mode; srv_kernels.hip head_tail_kernel):
    // `deconv 64 -> 64 | bias | relu` (batch norm folded) whose only reader is `deconv 64 -> 1 | bias | sigmoid`.  OCR_SRV_HEAD=0: two launches
    static const bool head_on = [] { const char* e = getenv("OCR_SRV_HEAD"); return !(e && e[0] == '0'); }();
    for (size_t oi = 0; half_ && head_on && !keep_all_ && oi + 1 < ops_.size(); ++oi) {
      const Op &d1 = ops_[oi], &d2 = ops_[oi + 1];
      if (d1.kind != "deconv" || d2.kind != "deconv" || d1.geti("cout") != 64 || d1.cin_s != 64 || d2.geti("cout") != 1 || d2.cin_s != 64) continue;
      if (d1.act != srv::SACT_RELU || !d1.bias || d1.scale || d1.res_tid >= 0 || d1.npad != 256 || !d2.p0) continue;
      if (d2.geti("i") != d1.geti("o") || readers[d1.geti("o")] != 1) continue;
      std::string e;
      if (!srv::launch_head_tail(nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0, 0, nullptr, true, e)) continue;
      ht_head[oi] = 1;
      ht_tail[oi + 1] = 1;
    }
    // ---- the DB neck's `concat up=8,4,2,1` folded into the 3x3 conv that is its only reader (f16 build, production mode): with the
    // K order (channel tile, tap) channel tile j of that conv IS source j, and the halo form fetches tile j's patch from source j at
    // (y >> sh, x >> sh) (GemmArgs::cat_*) - the 256-channel tensor (0.94 GB at batch 32) is neither written nor read.  OCR_SRV_CAT=0: a concat launch
    static const bool cat_on = [] { const char* e = getenv("OCR_SRV_CAT"); return !(e && e[0] == '0'); }();
    for (size_t oi = 0; half_ && cat_on && !keep_all_ && oi + 1 < ops_.size(); ++oi) {
      const Op &c = ops_[oi], &v = ops_[oi + 1];
      if (c.kind != "concat" || v.kind != "conv" || !v.korder || v.geti("i") != c.geti("o") || readers[c.geti("o")] != 1) continue;
      if (v.geti("kh") != 3 || v.geti("kw") != 3 || v.geti("sh") != 1 || v.geti("sw") != 1 || v.geti("ph") != 1 || v.geti("pw") != 1 || v.res_tid >= 0 || v.scale) continue;
      const int ns = (int)c.ins.size();
      bool ok = ns >= 1 && ns <= 4 && v.cin_s == 64 * ns;
      for (int j = 0; ok && j < ns; ++j) ok = c.ups[j] >= 1 && (c.ups[j] & (c.ups[j] - 1)) == 0;  // (shapes are checked at bind: 64 channels each)
      if (ok) cat_head[oi] = 1;
    }
  }
  // ---- arena: every tensor a slot of its own (keep_all) or first-fit reuse by last reader
  std::vector<int> last_use(ntensors_ + 1, -1);
  for (size_t oi = 0; oi < ops_.size(); ++oi) {
    const Op& op = ops_[oi];
    if (op.kind == "output") continue;
    if (op.kind == "concat") for (int t : op.ins) last_use[t] = (int)oi;
    else last_use[op.geti("i") == 0 ? pk : op.geti("i")] = (int)oi;
    if (op.res_tid >= 0) last_use[op.res_tid] = (int)oi;
  }
  last_use[out_tid_] = 1 << 30;
  // a fused pair runs as ONE launch that writes the SECOND op's tensor: what the first op reads must outlive that tensor's allocation
  // (and is released then - the arena loop below skips a fused head, with it the releases that fall on its index)
  for (size_t oi = 0; oi + 1 < ops_.size(); ++oi)
    if (ht_head[oi] || mlp_head[oi] || cat_head[oi])
      for (int t = 0; t <= ntensors_; ++t)
        if (last_use[t] == (int)oi) last_use[t] = (int)oi + 1;
  for (size_t oi = 0; oi + 2 < ops_.size(); ++oi)
    if (ln_abs[oi]) {  // (after the loop above: the raw sum is read by the launch that writes fc2's tensor)
      const int u = ops_[oi].geti("i") == 0 ? ntensors_ : ops_[oi].geti("i");
      last_use[u] = std::max(last_use[u], (int)oi + 2);
    }
  {
    struct Blk { size_t off, size; };
    std::vector<Blk> free_list;
    size_t top = 0;
    auto alloc = [&](size_t bytes) {
      bytes = (bytes + 255) & ~(size_t)255;
      int bi = -1;
      for (size_t i = 0; i < free_list.size(); ++i)
        if (free_list[i].size >= bytes && (bi < 0 || free_list[i].size < free_list[bi].size)) bi = (int)i;
      if (bi >= 0) {
        const size_t off = free_list[bi].off;
        free_list[bi].off += bytes;
        free_list[bi].size -= bytes;
        if (!free_list[bi].size) free_list.erase(free_list.begin() + bi);
        return off;
      }
      const size_t off = top;
      top += bytes;
      return off;
    };
    auto release = [&](size_t off, size_t bytes) {
      bytes = (bytes + 255) & ~(size_t)255;
      free_list.push_back({off, bytes});
      std::sort(free_list.begin(), free_list.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
      for (size_t i = 0; i + 1 < free_list.size();)
        if (free_list[i].off + free_list[i].size == free_list[i + 1].off) { free_list[i].size += free_list[i + 1].size; free_list.erase(free_list.begin() + i + 1); }
        else ++i;
    };
    tensors_[pk].offset = alloc(tensors_[pk].bytes(half_));
    for (size_t oi = 0; oi < ops_.size(); ++oi) {
      const Op& op = ops_[oi];
      if (op.kind == "output") continue;
      const int o = op.geti("o");
      if (mlp_head[oi] || ht_head[oi] || cat_head[oi] || ln_abs[oi]) { tensors_[o].n = 0; continue; }  // the hidden tensor of a fused MLP / the head's 64-channel map: never written
      tensors_[o].offset = alloc(tensors_[o].bytes(half_));
      if (keep_all_) continue;
      for (int t = 1; t <= ntensors_; ++t)
        if (last_use[t] == (int)oi && tensors_[t].n) release(tensors_[t].offset, tensors_[t].bytes(half_));
    }
    if (top > arena_cap_) {
      if (arena_) (void)g_free(arena_);
      arena_ = nullptr;
      arena_cap_ = 0;
      if (g_malloc(&arena_, top) != hipSuccess) { (void)hipGetLastError(); err = "activation arena: hipMalloc of " + std::to_string(top >> 20) + " MB failed"; return false; }
      arena_cap_ = top;
    }
  }
  auto ptr = [&](int tid) { return (void*)(arena_ + tensors_[tid].offset); };
  // ---- launches
  {
    const long px = (long)N * H * W;
    Launch L;
    L.name = "pack_input";
    L.bytes = (double)px * (12 + 8 * esz);
    void* dst = ptr(pk);
    const bool hf = half_;
    L.fn = [this, dst, px, hf](hipStream_t st, std::string&) { srv::launch_pack_input(x_in_, dst, px, hf, st); return true; };
    launches_.push_back(L);
  }
  int gi = 0;
  for (size_t oi = 0; oi < ops_.size(); ++oi) {
    Op& op = ops_[oi];
    if (op.kind == "output") continue;
    const int o = op.geti("o");
    const SrvTensor& ot = tensors_[o];
    const bool hf = half_;
    Launch L;
    char nm[160];
    if (op.kind == "concat" && cat_head[oi]) continue;  // gathered by the conv behind it
    if (op.kind == "concat") {
      const void* src[4] = {nullptr, nullptr, nullptr, nullptr};
      int ups[4] = {1, 1, 1, 1};
      const int ns = (int)op.ins.size();
      if (ns > 4) { err = "concat: at most four sources"; return false; }
      const int cs = tensors_[op.ins[0]].cs;
      for (int j = 0; j < ns; ++j) {
        if (tensors_[op.ins[j]].cs != cs || tensors_[op.ins[j]].c != cs) { err = "concat: sources must have equal, unpadded channel counts"; return false; }
        src[j] = ptr(op.ins[j]);
        ups[j] = op.ups[j];
      }
      snprintf(nm, sizeof nm, "%zu.concat_%dx%d@%dx%dx%d", oi, ns, cs, ot.n, ot.h, ot.w);
      L.name = nm;
      L.bytes = 2.0 * ot.bytes(half_);
      void* dst = ptr(o);
      const int n_ = ot.n, oh = ot.h, ow = ot.w;
      std::array<const void*, 4> sa{src[0], src[1], src[2], src[3]};
      std::array<int, 4> ua{ups[0], ups[1], ups[2], ups[3]};
      L.fn = [=](hipStream_t st, std::string&) { srv::launch_concat_up(sa.data(), ua.data(), ns, cs, dst, n_, oh, ow, hf, st); return true; };
      launches_.push_back(L);
      continue;
    }
    if (mlp_tail[oi] || ht_tail[oi] || ln_abs[oi]) continue;  // launched with its head (an absorbed LayerNorm: inside the MLP behind it)
    const int itid = op.geti("i") == 0 ? pk : op.geti("i");
    const bool cat_in = oi > 0 && cat_head[oi - 1] && ops_[oi - 1].geti("o") == itid;
    SrvTensor in_cat;
    if (cat_in) {  // (the concatenation has no tensor: its shape from its sources)
      const Op& c = ops_[oi - 1];
      const SrvTensor& sl = tensors_[c.ins.back()];
      in_cat.n = sl.n; in_cat.h = sl.h * c.ups.back(); in_cat.w = sl.w * c.ups.back();
      in_cat.c = in_cat.cs = 64 * (int)c.ins.size();
    }
    const SrvTensor& in = cat_in ? in_cat : tensors_[itid];
    if (ht_head[oi]) {
      const Op& d2 = ops_[oi + 1];
      const SrvTensor& yt = tensors_[d2.geti("o")];
      snprintf(nm, sizeof nm, "%zu.head_tail_64_64_1@%dx%dx%d", oi, yt.n, yt.h, yt.w);
      L.name = nm;
      L.flops = 2.0 * in.pixels() * 64.0 * 256.0 + 2.0 * in.pixels() * 4.0 * 4.0 * 64.0;
      L.bytes = (double)in.bytes(half_) + (double)yt.bytes(half_);
      const void* xs = ptr(itid);
      float* dst = (float*)ptr(d2.geti("o"));
      const void* w1 = op.wimg;
      const float *b1 = op.bias, *w4 = d2.p0;
      const float fb = d2.fbias;
      const int n_ = in.n, h_ = in.h, w_ = in.w;
      L.fn = [=](hipStream_t st, std::string& e) { return srv::launch_head_tail(xs, w1, b1, w4, fb, dst, n_, h_, w_, st, false, e); };
      launches_.push_back(L);
      continue;
    }
    if (mlp_head[oi]) {
      const Op& f2 = ops_[oi + 1];
      const SrvTensor& yt = tensors_[f2.geti("o")];
      const bool lnin = oi > 0 && ln_abs[oi - 1];  // its input's LayerNorm absorbed: it reads the raw sum
      const int xtid = lnin ? (ops_[oi - 1].geti("i") == 0 ? pk : ops_[oi - 1].geti("i")) : itid;
      const SrvTensor& xt = tensors_[xtid];
      const int c = xt.c;
      snprintf(nm, sizeof nm, "%zu.mlp%s_%d_%d_%d@%dx%dx%d", oi, lnin ? "_ln" : "", c, 4 * c, c, yt.n, yt.h, yt.w);
      L.name = nm;
      const double M_ = (double)xt.pixels();
      L.flops = 2.0 * 2.0 * M_ * c * 4.0 * c;
      L.bytes = (double)xt.bytes(half_) + (double)yt.bytes(half_) + 2.0 * 4.0 * c * c * esz;
      const void* xs = ptr(xtid);
      void* dst = ptr(f2.geti("o"));
      const unsigned long long xb = xt.bytes(half_);
      const void *w1 = lnin ? op.wimg_ln : op.wimg, *w2 = f2.wimg;
      const unsigned long long w1b = lnin ? op.wimg_ln_bytes : op.wimg_bytes, w2b = f2.wimg_bytes;
      const int n1 = op.npad, n2 = f2.npad;
      const float *b1 = op.bias, *b2 = f2.bias;
      const long M = xt.pixels();
      if (lnin) {
        const Op& l = ops_[oi - 1];
        srv::MlpLn ln;
        ln.g = l.p0; ln.b = l.p1; ln.s = op.ln_s; ln.c = op.ln_c; ln.eps = l.getf("eps");
        L.fn = [=](hipStream_t st, std::string& e) { return srv::launch_mlp_ln(xs, xb, w1, w1b, n1, w2, w2b, n2, ln, b2, dst, M, c, st, false, e); };
      } else {
        L.fn = [=](hipStream_t st, std::string& e) { return srv::launch_mlp(xs, xb, w1, w1b, n1, w2, w2b, n2, b1, b2, dst, M, c, st, false, e); };
      }
      launches_.push_back(L);
      continue;
    }
    const bool gemm = (op.kind == "conv" || op.kind == "linear" || op.kind == "deconv") && op.ncols > 0;
    if (gemm) {
      srv::GemmArgs a;
      a.x = cat_in ? ptr(ops_[oi - 1].ins[0]) : ptr(itid);
      a.x_bytes = cat_in ? tensors_[ops_[oi - 1].ins[0]].bytes(half_) : in.bytes(half_);
      if (cat_in) {
        const Op& c = ops_[oi - 1];
        a.cat_n = (int)c.ins.size();
        for (int j = 0; j < a.cat_n; ++j) {
          const SrvTensor& sj = tensors_[c.ins[j]];
          if (sj.cs != 64 || sj.c != 64 || sj.n != in.n || sj.h * c.ups[j] != in.h || sj.w * c.ups[j] != in.w) { err = "server plan: folded concat: source shapes"; return false; }
          a.cat_x[j] = ptr(c.ins[j]);
          a.cat_bytes[j] = sj.bytes(half_);
          int sh_ = 0;
          while ((1 << sh_) < c.ups[j]) ++sh_;
          a.cat_sh[j] = sh_;
        }
      }
      a.w = op.wimg;
      a.w_bytes = op.wimg_bytes;
      a.y = ptr(o);
      a.y_bytes = ot.bytes(half_);
      a.N = in.n; a.H = in.h; a.W = in.w; a.Cin = in.cs;
      if (in.cs != op.cin_s) { err = "server plan: op " + std::to_string(oi) + " reads a tensor of " + std::to_string(in.c) + " channels"; return false; }
      a.Npad = op.npad;
      a.Ncols = op.ncols;
      a.Cs_out = ot.cs;
      const int KG = half_ ? 8 : 4, BK = 8 * KG;
      if (op.kind == "conv") {
        a.KH = op.geti("kh"); a.KW = op.geti("kw"); a.SH = op.geti("sh"); a.SW = op.geti("sw"); a.PH = op.geti("ph"); a.PW = op.geti("pw");
        a.OH = ot.h; a.OW = ot.w;
      } else if (op.kind == "linear") {
        a.OH = in.h; a.OW = in.w;
      } else {
        a.OH = in.h; a.OW = in.w;  // GEMM columns = INPUT pixels of the transposed conv
        a.deconv = 1;
        a.CoutD = op.geti("cout");
      }
      a.M = (long)in.n * a.OH * a.OW;
      a.K = a.KH * a.KW * a.Cin;
      a.nkt = (a.K + BK - 1) / BK;
      a.x1 = (a.KH == 1 && a.KW == 1 && a.SH == 1 && a.SW == 1 && a.PH == 0 && a.PW == 0 && a.K % BK == 0) ? 1 : 0;
      a.cin_shift = -1;
      if (!a.x1 && a.Cin % BK != 0) {  // a K tile spans taps: every lane finds (tap, channel) of its granule - by shift for the
        int sh_ = 0;                   // powers of two (8, 32), by division otherwise (96: SVTR's second patch-embedding conv)
        while ((1 << sh_) < a.Cin) ++sh_;
        a.cin_shift = (1 << sh_) == a.Cin ? sh_ : -2;
      }
      a.bias = op.bias; a.scale = op.scale; a.shift = op.shift;
      a.act = op.act;
      a.korder = op.korder;
      {
        // tile order (GemmArgs::group_m): a weight slab beyond ~1.5 MB does not stay in an XCD's 4 MB L2 beside the pixel panels and the
        // output stream - it would be fetched again for every pixel tile (PMC on the CTC head, 5 MB of weights: 1.77 GB fetched per launch
        // against 68 MB of operands).  OCR_SRV_GROUP_M=0 / n overrides
        static const int gm_env = [] { const char* e = getenv("OCR_SRV_GROUP_M"); return e && *e ? atoi(e) : -1; }();
        const double slab = (double)op.ncols * (double)a.K * esz;
        a.group_m = gm_env >= 0 ? gm_env : (slab > 1.5e6 ? 8 : 0);
      }
      if (op.res_tid >= 0) {
        const SrvTensor& rt = tensors_[op.res_tid];
        a.res = ptr(op.res_tid);
        a.res_bytes = rt.bytes(half_);
        a.res_up = op.res_up;
        const bool okshape = op.res_up == 1 ? (rt.h == ot.h && rt.w == ot.w) : (rt.h * 2 == ot.h && rt.w * 2 == ot.w);
        if (!okshape || rt.cs != ot.cs || rt.n != ot.n || op.kind == "deconv") { err = "server plan: residual of op " + std::to_string(oi) + " has another shape"; return false; }
      }
      a.out_f32 = ot.f32 ? 1 : 0;
      const bool ctc_here = ctc_ && half_ && !keep_all_ && o == out_tid_ && ot.f32 && op.kind == "linear" && op.res_tid < 0 && !op.scale && op.act == srv::SACT_NONE &&
                            (size_t)((op.ncols + 63) / 64) * 16 <= (size_t)ot.cs * 4;  // (the partials of a row fit the row's logits slot)
      if (ctc_here) {
        a.ctc_part = (float*)ptr(o);
        a.ctc_slots = (op.ncols + 63) / 64;
      }
      snprintf(nm, sizeof nm, "%zu.%s%dx%d_%d_%d_s%d%s@%dx%dx%d", oi, op.kind.c_str(), a.KH, a.KW, in.c, op.geti("cout"), a.SH * 10 + a.SW,
               op.res_tid >= 0 ? "_res" : "", ot.n, ot.h, ot.w);
      const std::string lname = std::string(nm) + (ctc_here ? "_ctc" : "") + (cat_in ? "_cat" : "");  // (also the tuning key: these forms have their own candidates)
      const int cfg = tune(a, lname, s);
      if (cfg < 0) { err = std::string("no tile configuration runs ") + lname; return false; }
      L.name = lname + "[" + srv::gemm_config_name(cfg) + "]";
      if (ctc_here) { ctc_slots_ = a.ctc_slots; ctc_step_ = std::max(1, srv::gemm_config_bn(cfg) / 64); }
      const double klog = (double)a.KH * a.KW * in.c;
      L.flops = 2.0 * (double)a.M * klog * (double)op.ncols;
      double in_bytes = (double)in.bytes(half_);
      if (cat_in) {  // (what is read: the sources at their own resolutions)
        in_bytes = 0;
        for (int j = 0; j < a.cat_n; ++j) in_bytes += (double)a.cat_bytes[j];
      }
      L.bytes = in_bytes + (double)ot.bytes(half_) + (double)op.ncols * klog * esz + (op.res_tid >= 0 ? (double)tensors_[op.res_tid].bytes(half_) / (op.res_up == 2 ? 1.0 : 1.0) : 0.0);
      L.fn = [a, hf, cfg](hipStream_t st, std::string& e) { return srv::launch_gemm(a, hf, cfg, st, e); };
      launches_.push_back(L);
      ++gi;
      continue;
    }
    if (op.kind == "deconv") {  // -> the probability map
      snprintf(nm, sizeof nm, "%zu.deconv_map_%d@%dx%dx%d", oi, in.c, ot.n, ot.h, ot.w);
      L.name = nm;
      L.flops = 2.0 * in.pixels() * 4 * in.c;
      L.bytes = (double)in.bytes(half_) + (double)ot.bytes(half_);
      const void* xs = ptr(itid);
      float* dst = (float*)ptr(o);
      const float* w4 = op.p0;
      const float fb = op.fbias;
      const int n_ = in.n, h_ = in.h, w_ = in.w, cs = in.cs;
      L.fn = [=](hipStream_t st, std::string&) { srv::launch_deconv_to_map(xs, w4, fb, dst, n_, h_, w_, cs, hf, st); return true; };
    } else if (op.kind == "pool") {
      snprintf(nm, sizeof nm, "%zu.pool_%s%dx%d_%d@%dx%dx%d", oi, op.kv["type"].c_str(), op.geti("kh"), op.geti("kw"), in.c, ot.n, ot.h, ot.w);
      L.name = nm;
      L.bytes = (double)in.bytes(half_) + (double)ot.bytes(half_);
      const void* xs = ptr(itid);
      void* dst = ptr(o);
      const int n_ = in.n, h_ = in.h, w_ = in.w, cs = in.cs, oh = ot.h, ow = ot.w;
      const int kh = op.geti("kh"), kw = op.geti("kw"), sh = op.geti("sh"), sw = op.geti("sw"), ph = op.geti("ph"), pw = op.geti("pw");
      const bool mx = op.kv["type"] == "max";
      L.fn = [=](hipStream_t st, std::string&) { srv::launch_pool(xs, dst, n_, h_, w_, cs, oh, ow, kh, kw, sh, sw, ph, pw, mx, hf, st); return true; };
    } else if (op.kind == "ew") {
      snprintf(nm, sizeof nm, "%zu.addpos_%d@%dx%dx%d", oi, in.c, ot.n, ot.h, ot.w);
      L.name = nm;
      L.bytes = 2.0 * ot.bytes(half_);
      const void* xs = ptr(itid);
      void* dst = ptr(o);
      const float* pos = op.p0;
      const long px = in.pixels();
      const int hw = in.h * in.w, cs = in.cs;
      L.fn = [=](hipStream_t st, std::string&) { srv::launch_addpos(xs, pos, dst, px, hw, cs, hf, st); return true; };
    } else if (op.kind == "ln") {
      if (in.c != in.cs) { err = "layer norm: channel count must be a multiple of 8"; return false; }
      snprintf(nm, sizeof nm, "%zu.ln_%d@%dx%dx%d", oi, in.c, ot.n, ot.h, ot.w);
      L.name = nm;
      L.bytes = 2.0 * ot.bytes(half_);
      const void* xs = ptr(itid);
      void* dst = ptr(o);
      const float *g = op.p0, *b = op.p1;
      const long rows = in.pixels();
      const int c = in.c;
      const float eps = op.getf("eps");
      L.fn = [=](hipStream_t st, std::string&) { srv::launch_layernorm(xs, g, b, dst, rows, c, eps, hf, st); return true; };
    } else if (op.kind == "attn") {
      const int heads = op.geti("heads"), hd = op.geti("hd"), T = in.h * in.w;
      const int gh = op.geti("gh", in.h), gw = op.geti("gw", in.w), lh = op.geti("lh", 0), lw = op.geti("lw", 0);
      if (gh != in.h || gw != in.w) { err = "attention: the plan's token grid is " + std::to_string(gh) + "x" + std::to_string(gw) + ", the tensor's " + std::to_string(in.h) + "x" + std::to_string(in.w); return false; }
      snprintf(nm, sizeof nm, "%zu.attn_%s_h%d@%dx%dx%d", oi, lh > 0 ? "local" : "global", heads, ot.n, ot.h, ot.w);
      L.name = nm;
      // keys that take part: the window (clipped at the grid's border) or every token
      double keys = T;
      if (lh > 0) {
        double ky = 0, kx = 0;
        for (int y = 0; y < gh; ++y) ky += std::min(gh - 1, y + lh / 2) - std::max(0, y - lh / 2) + 1;
        for (int x = 0; x < gw; ++x) kx += std::min(gw - 1, x + lw / 2) - std::max(0, x - lw / 2) + 1;
        keys = (ky / gh) * (kx / gw);
      }
      L.flops = 4.0 * (double)in.n * heads * T * keys * hd;
      L.bytes = (double)in.bytes(half_) + (double)ot.bytes(half_);
      const void* xs = ptr(itid);
      void* dst = ptr(o);
      const float scale = op.getf("scale");
      const int n_ = in.n;
      L.fn = [=](hipStream_t st, std::string& e) { return srv::launch_attention(xs, dst, n_, T, heads, hd, scale, gh, gw, lh, lw, hf, st, e); };
    } else {
      err = "server plan: op " + op.kind + " has no kernel";
      return false;
    }
    launches_.push_back(L);
  }
  for (auto& L : launches_) { flops_ += L.flops; bytes_ += L.bytes; }
  bound_n_ = N; bound_h_ = H; bound_w_ = W;
  return true;
}

bool SrvNet::run(const float* x, int N, int H, int W, hipStream_t s, std::string& err) {
  if (N <= 0 || H <= 0 || W <= 0) { err = "bad shape"; return false; }
  if (N != bound_n_ || H != bound_h_ || W != bound_w_) {
    if (!bind(N, H, W, s, err)) { bound_n_ = -1; return false; }
  }
  x_in_ = x;
  for (auto& L : launches_) {
    hipEvent_t a = nullptr, b = nullptr;
    if (timing_) {
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      (void)hipEventRecord(a, s);
    }
    if (!L.fn(s, err)) { err = L.name + ": " + err; return false; }
    if (timing_) {
      (void)hipEventRecord(b, s);
      ev_pending_.push_back({a, b, L.name, L.flops, L.bytes});
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { err = std::string("server network launch: ") + hipGetErrorString(e); return false; }
  return true;
}

void SrvNet::collect_timings() {
  for (auto& p : ev_pending_) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      KernelTiming& t = timings_[p.name];
      t.ms += ms; t.count += 1; t.flops += p.flops; t.bytes += p.bytes;
    } else (void)hipGetLastError();
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  ev_pending_.clear();
}

bool SrvNet::fetch_logical(int tid, std::vector<float>& host, int dims[4], hipStream_t s, std::string& err) {
  if (tid < 0) tid = out_tid_;
  if (tid <= 0 || tid > ntensors_ || bound_n_ < 0 || tensors_[tid].n == 0) { err = "no such tensor"; return false; }
  const SrvTensor& t = tensors_[tid];
  dims[0] = t.n; dims[1] = t.h; dims[2] = t.w; dims[3] = t.c;
  const size_t cnt = (size_t)t.pixels() * t.c;
  host.resize(cnt);
  float* tmp = nullptr;
  if (g_malloc(&tmp, cnt * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); err = "hipMalloc failed"; return false; }
  if (t.f32) {
    if (t.cs == t.c) {
      (void)hipMemcpyAsync(tmp, tensor_ptr(tid), cnt * sizeof(float), hipMemcpyDeviceToDevice, s);
    } else {
      (void)hipMemcpy2DAsync(tmp, (size_t)t.c * 4, tensor_ptr(tid), (size_t)t.cs * 4, (size_t)t.c * 4, (size_t)t.pixels(), hipMemcpyDeviceToDevice, s);
    }
  } else {
    srv::launch_to_f32(tensor_ptr(tid), tmp, t.pixels(), t.cs, t.c, half_, s);
  }
  hipError_t e = g_stream_sync(s);
  if (e == hipSuccess) e = g_memcpy(host.data(), tmp, cnt * sizeof(float), hipMemcpyDeviceToHost);
  (void)g_free(tmp);
  if (e != hipSuccess) { err = std::string("fetch: ") + hipGetErrorString(e); return false; }
  return true;
}

}  // namespace ocr
