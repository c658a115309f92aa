#include "pd_format.h"

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sys/stat.h>

namespace ocr {

namespace {

struct Span {
  const uint8_t* p;
  size_t n;
};

bool read_file(const std::string& path, std::vector<uint8_t>& buf) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  buf.resize(sz);
  size_t got = sz ? fread(buf.data(), 1, sz, f) : 0;
  fclose(f);
  return got == (size_t)sz;
}

bool varint(Span& s, uint64_t& v) {
  v = 0;
  int shift = 0;
  while (s.n) {
    uint8_t c = *s.p++;
    --s.n;
    v |= (uint64_t)(c & 0x7f) << shift;
    if (c < 0x80) return true;
    shift += 7;
    if (shift > 63) return false;
  }
  return false;
}

// next field of a message; for length-delimited fields `sub` spans the payload
bool next_field(Span& s, int& field, int& wire, uint64_t& val, Span& sub) {
  uint64_t key;
  if (!varint(s, key)) return false;
  field = (int)(key >> 3);
  wire = (int)(key & 7);
  if (wire == 0) return varint(s, val);
  if (wire == 2) {
    uint64_t len;
    if (!varint(s, len) || len > s.n) return false;
    sub = {s.p, (size_t)len};
    s.p += len;
    s.n -= len;
    return true;
  }
  size_t w = wire == 5 ? 4 : wire == 1 ? 8 : 0;
  if (!w || s.n < w) return false;
  val = 0;
  memcpy(&val, s.p, w);
  s.p += w;
  s.n -= w;
  return true;
}

}  // namespace

bool file_exists(const std::string& path) {
  struct stat st;
  return stat(path.c_str(), &st) == 0;
}

bool pdmodel_persistable_names(const std::string& path, std::vector<std::string>& names, std::string& err) {
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) {
    err = "cannot read " + path;
    return false;
  }
  names.clear();
  Span prog{buf.data(), buf.size()};
  int f, w;
  uint64_t v;
  Span sub{};
  bool first_block = true;
  while (prog.n) {
    if (!next_field(prog, f, w, v, sub)) { err = "malformed ProgramDesc"; return false; }
    if (f != 1 || w != 2 || !first_block) continue;  // blocks = 1
    first_block = false;
    Span blk = sub;
    while (blk.n) {
      Span var{};
      if (!next_field(blk, f, w, v, var)) { err = "malformed BlockDesc"; return false; }
      if (f != 3 || w != 2) continue;  // vars = 3
      std::string name;
      bool persistable = false;
      while (var.n) {
        Span x{};
        if (!next_field(var, f, w, v, x)) { err = "malformed VarDesc"; return false; }
        if (f == 1 && w == 2) name.assign((const char*)x.p, x.n);
        if (f == 3 && w == 0) persistable = v != 0;
      }
      if (persistable && name != "feed" && name != "fetch") names.push_back(name);
    }
  }
  std::sort(names.begin(), names.end());
  if (names.empty()) { err = "no persistable variables in " + path; return false; }
  return true;
}

bool pdmodel_graph_signature(const std::string& path, int& nops, unsigned long long& fnv, std::string& err) {
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) {
    err = "cannot read " + path;
    return false;
  }
  nops = 0;
  fnv = 1469598103934665603ull;
  auto mix = [&](const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; ++i) { fnv ^= p[i]; fnv *= 1099511628211ull; }
  };
  Span prog{buf.data(), buf.size()};
  int f, w;
  uint64_t v;
  Span sub{};
  bool first_block = true;
  while (prog.n) {
    if (!next_field(prog, f, w, v, sub)) { err = "malformed ProgramDesc"; return false; }
    if (f != 1 || w != 2 || !first_block) continue;  // blocks = 1
    first_block = false;
    Span blk = sub;
    while (blk.n) {
      Span op{};
      if (!next_field(blk, f, w, v, op)) { err = "malformed BlockDesc"; return false; }
      if (f != 4 || w != 2) continue;  // ops = 4
      while (op.n) {
        Span x{};
        if (!next_field(op, f, w, v, x)) { err = "malformed OpDesc"; return false; }
        if (f == 3 && w == 2) {  // type = 3
          if (nops) { const uint8_t sep = ';'; mix(&sep, 1); }
          mix(x.p, x.n);
          ++nops;
        }
      }
    }
  }
  if (!nops) { err = "no ops in " + path; return false; }
  return true;
}

bool pdiparams_read(const std::string& path, const std::vector<std::string>& names, WeightMap& out, std::string& err) {
  std::vector<uint8_t> buf;
  if (!read_file(path, buf)) {
    err = "cannot read " + path;
    return false;
  }
  size_t i = 0;
  auto need = [&](size_t k) { return i + k <= buf.size(); };
  for (const std::string& name : names) {
    if (!need(12)) { err = "truncated pdiparams at " + name; return false; }
    uint64_t lod_levels;
    memcpy(&lod_levels, &buf[i + 4], 8);
    i += 12;
    for (uint64_t l = 0; l < lod_levels; ++l) {
      if (!need(8)) { err = "truncated lod"; return false; }
      uint64_t nb;
      memcpy(&nb, &buf[i], 8);
      i += 8 + nb;
    }
    if (!need(8)) { err = "truncated tensor header"; return false; }
    int32_t dlen;
    memcpy(&dlen, &buf[i + 4], 4);
    i += 8;
    if (dlen < 0 || !need((size_t)dlen)) { err = "bad TensorDesc length"; return false; }
    Span desc{&buf[i], (size_t)dlen};
    i += dlen;
    HostTensor t;
    int dtype = -1;
    while (desc.n) {
      int f, w;
      uint64_t v;
      Span sub{};
      if (!next_field(desc, f, w, v, sub)) { err = "malformed TensorDesc"; return false; }
      if (f == 1 && w == 0) dtype = (int)v;
      if (f == 2 && w == 0) t.dims.push_back((int)(int64_t)v);
      if (f == 2 && w == 2) {
        while (sub.n) {
          uint64_t d;
          if (!varint(sub, d)) { err = "malformed dims"; return false; }
          t.dims.push_back((int)(int64_t)d);
        }
      }
    }
    if (dtype != 5) { err = "parameter " + name + " is not FP32"; return false; }
    size_t n = 1;
    for (int d : t.dims) n *= (size_t)d;
    if (!need(n * 4)) { err = "truncated data for " + name; return false; }
    t.data.resize(n);
    memcpy(t.data.data(), &buf[i], n * 4);
    i += n * 4;
    out[name] = std::move(t);
  }
  if (i != buf.size()) { err = "pdiparams has trailing bytes (graph/params mismatch)"; return false; }
  return true;
}

}  // namespace ocr
