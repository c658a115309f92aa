// OCRIPCService for Linux: the JSON request/reply protocol of the reference's Windows named-pipe service
// (/root/reference/include/paddle_ocr/ocr_ipc_service.h:19-100, /root/reference/src/ocr_ipc_service.cpp:203-448)
// over a Unix-domain stream socket; a message (the pipe's PIPE_TYPE_MESSAGE unit) is framed as a 4-byte
// little-endian length followed by that many bytes of JSON, in both directions.
//
//   {"command":"recognize","image_path":"..."}            image file: PNG, JPEG, binary PPM (P6), BMP
//   {"command":"recognize","image_data":"<base64>"}       the same file contents, base64
//   {"command":"status"}    -> {"success":true,"status":"{\"running\":..,\"total_requests\":..,...}"}
//   {"command":"shutdown"}  -> {"success":true,"message":"Shutdown command received, stopping service..."}
// Replies to `recognize` are the worker's result JSON (ocr_worker.cpp:155-190) unchanged; every failure is
// {"success":false,"error":"..."} with the reference's messages.  Requests of 1 MiB - 1 bytes or more are
// answered with the reference's "Data too large for buffer" error (ocr_ipc_service.cpp:222-238).
//
// Differences, by necessity: the transport (socket path instead of \\.\pipe\ocr_service); replies are compact
// JSON (jsoncpp's default writer indents); cv::imread/imdecode are replaced by the decoders below - PNG through
// the system's libpng16 (its simplified API, loaded with dlopen: the image ships the .so but no headers), PPM
// and BMP natively, JPEG with the decoder of jpeg_decode.h (a restatement of libjpeg's default pipeline,
// sequential and progressive, checked bit for bit against libjpeg-turbo through PIL).
// There is no CPU worker pool: cpu_workers is accepted and ignored, gpu_workers = 0 leaves `recognize`
// answering with an error (status / shutdown still work - that is what the CPU-only tests drive).
#pragma once
#include <dlfcn.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "jpeg_decode.h"
#include "paddle_ocr_hip.h"

namespace PaddleOCR {
namespace ipc {

// ---- minimal JSON reader: top-level object, string members (what the protocol uses) ----
struct JsonObject {
  std::vector<std::pair<std::string, std::string>> strings;  // key -> string value
  std::string get(const std::string& k, const std::string& dflt = "") const {
    for (auto& kv : strings) if (kv.first == k) return kv.second;
    return dflt;
  }
};
class JsonParser {
 public:
  explicit JsonParser(const std::string& s) : s_(s) {}
  bool parse(JsonObject& out, std::string& err) {
    ws();
    if (!eat('{')) return fail(err, "expected '{'");
    ws();
    if (eat('}')) return tail(err);
    for (;;) {
      std::string key, val;
      ws();
      if (!string(key)) return fail(err, "expected member name");
      ws();
      if (!eat(':')) return fail(err, "expected ':'");
      ws();
      if (peek() == '"') {
        if (!string(val)) return fail(err, "bad string");
        out.strings.emplace_back(key, val);
      } else if (!skip_value()) {
        return fail(err, "bad value");
      }
      ws();
      if (eat(',')) continue;
      if (eat('}')) return tail(err);
      return fail(err, "expected ',' or '}'");
    }
  }

 private:
  bool tail(std::string& err) { ws(); return i_ == s_.size() ? true : fail(err, "trailing characters"); }
  bool fail(std::string& err, const char* what) {
    std::ostringstream o;
    o << "* Line 1, Column " << (i_ + 1) << "\n  Syntax error: " << what << "\n";
    err = o.str();
    return false;
  }
  char peek() const { return i_ < s_.size() ? s_[i_] : '\0'; }
  bool eat(char c) { if (peek() == c) { ++i_; return true; } return false; }
  void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\t' || s_[i_] == '\n' || s_[i_] == '\r')) ++i_; }
  static void utf8(std::string& o, unsigned cp) {
    if (cp < 0x80) o += (char)cp;
    else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
  }
  bool hex4(unsigned& v) {
    v = 0;
    for (int k = 0; k < 4; ++k) {
      const char c = peek();
      unsigned d;
      if (c >= '0' && c <= '9') d = c - '0'; else if (c >= 'a' && c <= 'f') d = c - 'a' + 10; else if (c >= 'A' && c <= 'F') d = c - 'A' + 10; else return false;
      v = v * 16 + d;
      ++i_;
    }
    return true;
  }
  bool string(std::string& out) {
    if (!eat('"')) return false;
    while (i_ < s_.size()) {
      const char c = s_[i_++];
      if (c == '"') return true;
      if (c != '\\') { out += c; continue; }
      if (i_ >= s_.size()) return false;
      const char e = s_[i_++];
      switch (e) {
        case '"': out += '"'; break; case '\\': out += '\\'; break; case '/': out += '/'; break;
        case 'b': out += '\b'; break; case 'f': out += '\f'; break; case 'n': out += '\n'; break;
        case 'r': out += '\r'; break; case 't': out += '\t'; break;
        case 'u': {
          unsigned cp;
          if (!hex4(cp)) return false;
          if (cp >= 0xD800 && cp <= 0xDBFF && peek() == '\\') {  // surrogate pair
            const size_t save = i_;
            unsigned lo;
            ++i_;
            if (eat('u') && hex4(lo) && lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            else i_ = save;
          }
          utf8(out, cp);
        } break;
        default: return false;
      }
    }
    return false;
  }
  bool skip_value() {  // number, literal, array or object: skipped with nesting and strings respected
    int depth = 0;
    const size_t start = i_;
    while (i_ < s_.size()) {
      const char c = s_[i_];
      if (c == '"') { std::string tmp; if (!string(tmp)) return false; continue; }
      if (c == '{' || c == '[') ++depth;
      else if (c == '}' || c == ']') { if (depth == 0) break; --depth; }
      else if (c == ',' && depth == 0) break;
      ++i_;
    }
    return i_ > start;
  }
  const std::string& s_;
  size_t i_ = 0;
};

inline void json_escape_to(std::string& o, const std::string& s) { detail::json_escape(o, s); }
inline std::string error_reply(const std::string& msg) {
  std::string o = "{\"error\":";
  json_escape_to(o, msg);
  o += ",\"success\":false}";
  return o;
}

// ---- base64 (RFC 4648, padding optional, whitespace skipped); false on a foreign character ----
inline bool base64_decode(const std::string& in, std::vector<uint8_t>& out) {
  unsigned acc = 0;
  int bits = 0;
  for (const unsigned char c : in) {
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A'; else if (c >= 'a' && c <= 'z') v = c - 'a' + 26; else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+') v = 62; else if (c == '/') v = 63; else if (c == '=') break;
    else if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue; else return false;
    acc = (acc << 6) | (unsigned)v;
    bits += 6;
    if (bits >= 8) { bits -= 8; out.push_back((uint8_t)((acc >> bits) & 0xFF)); }
  }
  return true;
}

// ---- image decoders: file contents -> packed BGR (what cv::imread / cv::imdecode return) ----
// Every decoder refuses headers that would make the service allocate gigabytes from a sub-megabyte request
// (the same 64 Mpixel cap as jpeg_decode.h).
constexpr long kMaxDecodedPixels = 64L << 20;
inline bool decode_ppm(const std::vector<uint8_t>& d, Image& im) {
  if (d.size() < 11 || d[0] != 'P' || d[1] != '6') return false;
  size_t i = 2;
  long v[3];
  for (int k = 0; k < 3; ++k) {
    for (;;) {  // whitespace and comments
      while (i < d.size() && (d[i] == ' ' || d[i] == '\n' || d[i] == '\r' || d[i] == '\t')) ++i;
      if (i < d.size() && d[i] == '#') { while (i < d.size() && d[i] != '\n') ++i; continue; }
      break;
    }
    if (i >= d.size() || d[i] < '0' || d[i] > '9') return false;
    long x = 0;
    while (i < d.size() && d[i] >= '0' && d[i] <= '9') { x = x * 10 + (d[i] - '0'); if (x > 1000000) return false; ++i; }
    v[k] = x;
  }
  ++i;  // the single whitespace after maxval
  const long w = v[0], h = v[1];
  if (w <= 0 || h <= 0 || w * h > kMaxDecodedPixels || v[2] != 255 || d.size() < i + (size_t)w * h * 3) return false;
  im.rows = (int)h; im.cols = (int)w;
  im.pixels.resize((size_t)w * h * 3);
  for (size_t p = 0; p < (size_t)w * h; ++p) {  // RGB -> BGR
    im.pixels[3 * p] = d[i + 3 * p + 2]; im.pixels[3 * p + 1] = d[i + 3 * p + 1]; im.pixels[3 * p + 2] = d[i + 3 * p];
  }
  return true;
}
inline bool decode_bmp(const std::vector<uint8_t>& d, Image& im) {
  if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
  auto u32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
  const uint32_t off = u32(10), hdr = u32(14);
  const int32_t w = (int32_t)u32(18), hs = (int32_t)u32(22);
  const unsigned bpp = d[28] | (d[29] << 8), comp = u32(30);
  if (hdr < 40 || w <= 0 || hs == 0 || (bpp != 24 && bpp != 32) || (comp != 0 && !(bpp == 32 && comp == 3))) return false;
  const int h = hs < 0 ? -hs : hs;
  if (hs == INT32_MIN || (long)w * h > kMaxDecodedPixels) return false;
  const size_t stride = ((size_t)w * (bpp / 8) + 3) & ~(size_t)3;
  if (d.size() < off + stride * h) return false;
  im.rows = h; im.cols = w;
  im.pixels.resize((size_t)w * h * 3);
  for (int y = 0; y < h; ++y) {
    const uint8_t* src = d.data() + off + stride * (hs < 0 ? y : h - 1 - y);  // bottom-up unless the height is negative
    for (int x = 0; x < w; ++x) memcpy(&im.pixels[((size_t)y * w + x) * 3], src + (size_t)x * (bpp / 8), 3);  // BMP is BGR(A)
  }
  return true;
}
// libpng16 simplified API (png.h: png_image, PNG_IMAGE_VERSION 1, PNG_FORMAT_BGR = 0x12), resolved at run time
struct PngImage {
  void* opaque;
  uint32_t version, width, height, format, flags, colormap_entries, warning_or_error;
  char message[64];
};
inline bool decode_png(const std::vector<uint8_t>& d, Image& im) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (d.size() < 8 || memcmp(d.data(), sig, 8)) return false;
  static void* lib = dlopen("libpng16.so.16", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return false;
  typedef int (*BeginFn)(PngImage*, const void*, size_t);
  typedef int (*FinishFn)(PngImage*, const void*, void*, int32_t, void*);
  typedef void (*FreeFn)(PngImage*);
  static BeginFn begin = (BeginFn)dlsym(lib, "png_image_begin_read_from_memory");
  static FinishFn finish = (FinishFn)dlsym(lib, "png_image_finish_read");
  static FreeFn pfree = (FreeFn)dlsym(lib, "png_image_free");
  if (!begin || !finish || !pfree) return false;
  PngImage pi;
  memset(&pi, 0, sizeof pi);
  pi.version = 1;
  if (!begin(&pi, d.data(), d.size())) return false;
  pi.format = 0x12;  // PNG_FORMAT_BGR: 8-bit, colour, blue first, alpha removed (composited on black like a
                     // missing background; cv::imread(IMREAD_COLOR) drops alpha without compositing - differs
                     // only for translucent pixels)
  if (pi.width == 0 || pi.height == 0 || pi.width > 100000 || pi.height > 100000 ||
      (long)pi.width * (long)pi.height > kMaxDecodedPixels) { pfree(&pi); return false; }
  im.rows = (int)pi.height; im.cols = (int)pi.width;
  im.pixels.resize((size_t)pi.width * pi.height * 3);
  if (!finish(&pi, nullptr, im.pixels.data(), 0, nullptr)) { pfree(&pi); im = Image(); return false; }
  return true;
}
// device_pixels: stop after the entropy decoding and leave the pixel half to the GPU worker (OCR_DEVICE_JPEG=0 or
// device_pixels = false: everything on this host thread)
inline bool decode_jpeg(const std::vector<uint8_t>& d, Image& im, bool device_pixels = false) {
  if (d.size() < 4 || d[0] != 0xFF || d[1] != 0xD8) return false;
  jpeg::Decoder dec;
  static const char* env = getenv("OCR_DEVICE_JPEG");
  if (device_pixels && !(env && env[0] == '0')) {
    auto c = std::make_shared<jpeg::Coefs>();
    if (!dec.decode_coefficients(d.data(), d.size(), *c)) { im = Image(); return false; }
    im.pixels.clear();
    im.rows = c->rows; im.cols = c->cols;
    im.jpeg = std::move(c);
    return true;
  }
  if (!dec.decode(d.data(), d.size(), im.pixels, im.rows, im.cols)) { im = Image(); return false; }
  return true;
}
inline bool decode_image(const std::vector<uint8_t>& bytes, Image& im, bool device_pixels = false) {
  return decode_png(bytes, im) || decode_jpeg(bytes, im, device_pixels) || decode_ppm(bytes, im) || decode_bmp(bytes, im);
}
inline bool read_file(const std::string& path, std::vector<uint8_t>& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  return true;
}

}  // namespace ipc

class OCRIPCService {
 public:
  static const int PIPE_OUTPUT_BUFFER_SIZE = 65536;   // ocr_ipc_service.h:86
  static const int READ_BUFFER_SIZE = 1048576;        // ocr_ipc_service.h:88

  OCRIPCService(const std::string& model_dir, const std::string& socket_path = "/tmp/ocr_service.sock", int gpu_workers = 0,
                int cpu_workers = 1)
      : model_dir_(model_dir), socket_path_(socket_path), gpu_workers_(gpu_workers), cpu_workers_(cpu_workers) {}
  ~OCRIPCService() { stop(); }

  bool start() {
    if (running_) return true;
    if (gpu_workers_ > 0) {
      gpu_worker_pool_.reset(new GPUWorkerPool(model_dir_, gpu_workers_));
      gpu_worker_pool_->start();
    }
    listen_fd_ = socket(AF_UNIX, SOCK_STREAM, 0);
    if (listen_fd_ < 0) return false;
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    if (socket_path_.size() >= sizeof addr.sun_path) return false;
    strcpy(addr.sun_path, socket_path_.c_str());
    unlink(socket_path_.c_str());
    if (bind(listen_fd_, (sockaddr*)&addr, sizeof addr) != 0 || listen(listen_fd_, 16) != 0) { close(listen_fd_); listen_fd_ = -1; return false; }
    running_ = true;
    accept_thread_ = std::thread(&OCRIPCService::acceptLoop, this);
    return true;
  }
  void stop() {
    std::lock_guard<std::mutex> stop_lock(stop_mutex_);  // a second caller waits for the first to finish
    if (!running_.exchange(false)) return;
    if (listen_fd_ >= 0) { shutdown(listen_fd_, SHUT_RDWR); close(listen_fd_); listen_fd_ = -1; }
    if (accept_thread_.joinable() && accept_thread_.get_id() != std::this_thread::get_id()) accept_thread_.join();
    {
      std::lock_guard<std::mutex> lock(client_threads_mutex_);
      for (int fd : client_fds_) shutdown(fd, SHUT_RDWR);
    }
    for (;;) {  // client threads remove themselves; wait for them
      std::thread t;
      {
        std::lock_guard<std::mutex> lock(client_threads_mutex_);
        if (client_threads_.empty()) break;
        t = std::move(client_threads_.back());
        client_threads_.pop_back();
      }
      if (t.joinable()) t.join();
    }
    if (gpu_worker_pool_) gpu_worker_pool_->stop();
    unlink(socket_path_.c_str());
  }
  bool isRunning() const { return running_; }
  void waitUntilStopped() {
    while (running_) std::this_thread::sleep_for(std::chrono::milliseconds(20));
    std::lock_guard<std::mutex> stop_lock(stop_mutex_);  // ... and for whoever is stopping it to be done
  }

  // processIPCRequest (ocr_ipc_service.cpp:310-423)
  std::string processIPCRequest(const std::string& request_json) {
    try {
      ipc::JsonObject req;
      std::string errors;
      if (!ipc::JsonParser(request_json).parse(req, errors)) return ipc::error_reply("Invalid JSON: " + errors);
      const std::string command = req.get("command");
      if (command == "recognize") {
        const std::string image_path = req.get("image_path"), image_base64 = req.get("image_data");
        Image image;
        std::string error_msg;
        if (!image_path.empty()) {
          std::vector<uint8_t> bytes;
          if (!ipc::read_file(image_path, bytes) || !ipc::decode_image(bytes, image, gpu_worker_pool_ != nullptr) || image.empty())
            error_msg = "Failed to load image from path: " + image_path;
        } else if (!image_base64.empty()) {
          std::vector<uint8_t> bytes;
          if (!ipc::base64_decode(image_base64, bytes)) error_msg = "Base64 decode error: invalid character";
          else if (!ipc::decode_image(bytes, image, gpu_worker_pool_ != nullptr) || image.empty()) error_msg = "Failed to decode base64 image data";
        } else {
          error_msg = "Missing image_path or image_data";
        }
        if (!error_msg.empty()) return ipc::error_reply(error_msg);
        if (!gpu_worker_pool_) return ipc::error_reply("No GPU workers configured (this build has no CPU path)");
        const int request_id = request_counter_.fetch_add(1);
        auto request = std::make_shared<OCRRequest>(request_id, std::move(image));
        total_requests_.fetch_add(1);
        std::string result = gpu_worker_pool_->submitRequest(request).get();
        if (result.find("\"success\":true") != std::string::npos) successful_requests_.fetch_add(1);
        const size_t p = result.find("\"processing_time_ms\":");
        if (p != std::string::npos) {
          std::lock_guard<std::mutex> lock(stats_mutex_);
          total_processing_time_ += atof(result.c_str() + p + 21);
        }
        return result;
      } else if (command == "status") {
        std::string o = "{\"status\":";
        ipc::json_escape_to(o, getStatusInfo());
        o += ",\"success\":true}";
        return o;
      } else if (command == "shutdown") {
        std::thread([this]() {  // let the reply go out first (ocr_ipc_service.cpp:381-403)
          std::this_thread::sleep_for(std::chrono::milliseconds(50));
          this->stop();
        }).detach();
        return "{\"message\":\"Shutdown command received, stopping service...\",\"success\":true}";
      }
      return ipc::error_reply("Unknown command: " + command);
    } catch (const std::exception& e) {
      return ipc::error_reply(e.what());
    }
  }

  // getStatusInfo (ocr_ipc_service.cpp:438-448)
  std::string getStatusInfo() const {
    double avg = 0.0;
    {
      std::lock_guard<std::mutex> lock(stats_mutex_);
      avg = total_requests_.load() > 0 ? total_processing_time_ / total_requests_.load() : 0.0;
    }
    char buf[256];
    snprintf(buf, sizeof buf, "{\"average_processing_time_ms\":%.17g,\"running\":%s,\"successful_requests\":%d,\"total_requests\":%d}", avg,
             running_.load() ? "true" : "false", successful_requests_.load(), total_requests_.load());
    return buf;
  }

 private:
  static constexpr size_t kMaxClients = 256;
  void acceptLoop() {
    while (running_) {
      const int fd = accept(listen_fd_, nullptr, nullptr);
      if (fd < 0) { if (!running_) break; continue; }
      {
        // a bound on concurrent clients (each holds a thread and a 1 MiB request buffer): beyond it the connection is
        // closed at once, which a client sees as "service busy" and retries
        std::lock_guard<std::mutex> lock(client_threads_mutex_);
        if (client_fds_.size() >= kMaxClients) { close(fd); continue; }
      }
      const int big = 2 * READ_BUFFER_SIZE;
      setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &big, sizeof big);
      setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &big, sizeof big);
      std::lock_guard<std::mutex> lock(client_threads_mutex_);
      client_fds_.push_back(fd);
      client_threads_.emplace_back(&OCRIPCService::handleClientConnection, this, fd);
    }
  }
  static bool read_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) {
      const ssize_t r = recv(fd, c, n, 0);
      if (r <= 0) return false;
      c += r;
      n -= (size_t)r;
    }
    return true;
  }
  static bool write_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) {
      const ssize_t r = send(fd, c, n, MSG_NOSIGNAL);
      if (r <= 0) return false;
      c += r;
      n -= (size_t)r;
    }
    return true;
  }
  // handleClientConnection (ocr_ipc_service.cpp:203-308): one message in, one message out
  void handleClientConnection(int fd) {
    std::vector<char> buffer(READ_BUFFER_SIZE);
    while (running_) {
      uint32_t len = 0;
      if (!read_all(fd, &len, 4)) break;  // peer closed
      std::string response;
      if (len >= (uint32_t)READ_BUFFER_SIZE - 1) {
        // the pipe version reads at most READ_BUFFER_SIZE - 1 bytes and rejects a message that fills them
        response = ipc::error_reply("Data too large for buffer (max 1MB). Consider using file path transmission.");
        uint32_t left = len;  // drain the oversized message so the stream stays in step
        bool ok = true;
        while (left && ok) { const uint32_t n = left < (uint32_t)READ_BUFFER_SIZE ? left : (uint32_t)READ_BUFFER_SIZE; ok = read_all(fd, buffer.data(), n); left -= n; }
        if (!ok) break;
      } else {
        if (!read_all(fd, buffer.data(), len)) break;
        response = len ? processIPCRequest(std::string(buffer.data(), (size_t)len)) : std::string();
        if (!len) continue;  // "Received 0 bytes": ignored (ocr_ipc_service.cpp:213-217)
      }
      const uint32_t rl = (uint32_t)response.size();
      if (!write_all(fd, &rl, 4) || !write_all(fd, response.data(), response.size())) break;
    }
    close(fd);
    std::lock_guard<std::mutex> lock(client_threads_mutex_);
    for (size_t i = 0; i < client_fds_.size(); ++i)
      if (client_fds_[i] == fd) { client_fds_.erase(client_fds_.begin() + i); break; }
    for (size_t i = 0; i < client_threads_.size(); ++i)
      if (client_threads_[i].get_id() == std::this_thread::get_id()) {
        client_threads_[i].detach();
        client_threads_.erase(client_threads_.begin() + i);
        break;
      }
  }

  std::string model_dir_, socket_path_;
  int gpu_workers_, cpu_workers_;
  std::atomic<bool> running_{false};
  int listen_fd_ = -1;
  std::thread accept_thread_;
  std::vector<std::thread> client_threads_;
  std::vector<int> client_fds_;
  std::mutex client_threads_mutex_, stop_mutex_;
  std::unique_ptr<GPUWorkerPool> gpu_worker_pool_;
  std::atomic<int> request_counter_{1}, total_requests_{0}, successful_requests_{0};
  mutable std::mutex stats_mutex_;
  double total_processing_time_ = 0.0;
};

}  // namespace PaddleOCR
