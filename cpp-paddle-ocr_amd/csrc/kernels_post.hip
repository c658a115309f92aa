// DB post-processing on the GPU (replaces /root/reference/src/ocr_det.cpp:137-165 and
// /root/reference/src/postprocess_op.cpp:20-362 incl. the ClipperOffset calls into
// /root/reference/src/clipper.cpp:3628-4021, and the OpenCV calls they make: threshold, dilate,
// findContours(RETR_LIST, CHAIN_APPROX_SIMPLE), minAreaRect, boxPoints, fillPoly, mean).
//
// Structure (all HBM-bound integer/byte work; no GEMM):
//   1. bitmap            fused into the DB-head tail kernel, or bitmap_kernel for caller maps
//   2. CCL               union-find with atomicMin over 8-connected foreground AND 4-connected
//                        background in one label image: a border of the reference's raster scan
//                        starts exactly at the minimum-index pixel of a foreground component
//                        (outer border) or of a background component that does not touch the
//                        frame (hole border) — so border starts are found without the sequential
//                        scan-and-mark of Suzuki-Abe.
//   3. ordered compaction of the starts (one workgroup per image, wave ballots) -> the last
//      `max_candidates` discovered, in the reference's reverse discovery order
//   4. per border: one lane follows it (read-only bitmap, so borders are independent), emitting the
//      CHAIN_APPROX_SIMPLE vertices as sortable keys
//   5. per border, one wave: bitonic sort of the keys (LDS), Sklansky hull, float rotating
//      calipers, box score as a wave-parallel masked mean with the fillPoly raster rule in closed
//      form, Clipper round-join offset, second minAreaRect, integer box, FilterTagDetRes
//   6. ordered compaction of surviving boxes
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>

#include "kernels_post.h"
#include "lds_attr.h"
#include "rt_options.h"

// development probe (OCR_POST_STOP, only in -DOCR_DEV_PROBES builds): the per-border stage returns after phase n
#ifdef OCR_DEV_PROBES
#define OCR_PROBE_STOP(a, n) ((a).probe_stop == (n))
#else
#define OCR_PROBE_STOP(a, n) false
#endif

namespace ocr {

// ------------------------------------------------------------------ 1. bitmap for caller-supplied maps
__global__ void __launch_bounds__(256) bitmap_kernel(const float* __restrict__ prob, uint8_t* __restrict__ bm, long total,
                                                     int ithresh) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int u8 = (int)(unsigned char)(prob[i] * 255.0f);
  bm[i] = u8 > ithresh ? 1 : 0;
}
// 2x2 rect dilate, anchor (1,1): dst(y,x) = max src(y-1..y, x-1..x)
__global__ void __launch_bounds__(256) dilate2_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int N, int H,
                                                      int W) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * H * W) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  uint8_t m = src[i];
  if (x > 0) m |= src[i - 1];
  if (y > 0) m |= src[i - W];
  if (x > 0 && y > 0) m |= src[i - W - 1];
  dst[i] = m;
}
void launch_bitmap(const float* prob, uint8_t* bm, long total, int ithresh, hipStream_t s) {
  hipLaunchKernelGGL(bitmap_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, prob, bm, total, ithresh);
}
void launch_dilate2(const uint8_t* src, uint8_t* dst, int N, int H, int W, hipStream_t s) {
  const long total = (long)N * H * W;
  hipLaunchKernelGGL(dilate2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, N, H, W);
}

// ------------------------------------------------------------------ 2. connected components
__device__ __forceinline__ int uf_find(const int* L, int a) {
  int p = L[a];
  while (p != a) { a = p; p = L[a]; }
  return a;
}
__device__ __forceinline__ void uf_unite(int* L, int a, int b) {
  bool done;
  do {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a < b) { const int old = atomicMin(&L[b], a); done = (old == b); b = old; }
    else if (b < a) { const int old = atomicMin(&L[a], b); done = (old == a); a = old; }
    else done = true;
  } while (!done);
}
// Initial labels = start pixel of the horizontal run (same value as the left neighbours) a pixel lies in:
// one wave walks one image row in 64-pixel segments (ballot + carry), so the union-find below only
// has to join RUNS, not pixels.
// (grid: rows / 4 x images; `img` != null: a batch of mixed sizes, H x W is the largest)
__global__ void __launch_bounds__(256) ccl_rows_kernel(const uint8_t* __restrict__ bm, int* __restrict__ L,
                                                       uint8_t* __restrict__ touch, int N, int H, int W, const PostImg* __restrict__ img) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  long base0 = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base0 = img[n].off; }
  if (row >= H) return;
  const long base = base0 + (long)row * W;
  int carry = 0;
  for (int x0 = 0; x0 < W; x0 += 64) {
    const int x = x0 + lane;
    const bool in = x < W;
    const uint8_t v = in ? bm[base + x] : 2;
    const uint8_t pv = (in && x > 0) ? bm[base + x - 1] : 3;
    const bool st = in && (v != pv);
    const unsigned long long bal = __ballot(st);
    const unsigned long long below = bal & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int rs = below ? x0 + (63 - __clzll(below)) : carry;
    if (in) {
      L[base + x] = (int)(base + rs);
      touch[base + x] = 0;
    }
    carry = __shfl(rs, 63);
  }
}
__device__ __forceinline__ bool run_start(const uint8_t* bm, long i, int x) { return x == 0 || bm[i - 1] != bm[i]; }
// (one wave per image row, walking it in 64-pixel segments as ccl_rows_kernel does: a flat pixel index costs two
// 64-bit divisions per thread, most of this kernel's instructions)
__global__ void __launch_bounds__(256) ccl_merge_kernel(const uint8_t* __restrict__ bm, int* __restrict__ L, int N, int H,
                                                        int W, const PostImg* __restrict__ img) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  long base0 = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base0 = img[n].off; }
  if (row >= H || row == 0) return;  // the first row of an image has nothing above it
  const long base = base0 + (long)row * W;
  for (int x0 = 0; x0 < W; x0 += 64) {
    const int x = x0 + lane;
    if (x >= W) break;
    const long i = base + x;
    const uint8_t v = bm[i], up = bm[i - W];
    if (up == v) {
      // two vertically overlapping runs are joined once, at the leftmost column of the overlap
      if (run_start(bm, i, x) || run_start(bm, i - W, x)) uf_unite(L, (int)i, (int)(i - W));
    } else if (v) {
      // 8-connectivity of the foreground: diagonal-only contacts
      if (x > 0 && bm[i - W - 1]) uf_unite(L, (int)i, (int)(i - W - 1));
      if (x < W - 1 && bm[i - W + 1]) uf_unite(L, (int)i, (int)(i - W + 1));
    }
  }
}
// After the merge a pixel is a component's representative iff L[i] == i, and that is all the border-start search asks
// of the labels (is_start): no pass that rewrites every pixel's label to its root is needed.  What remains of it is the
// frame test of the BACKGROUND components (a hole border exists only for a background component that does not reach the
// frame): one thread per frame pixel marks the root of its component.
__global__ void __launch_bounds__(256) ccl_frame_kernel(const uint8_t* __restrict__ bm, const int* __restrict__ L,
                                                        uint8_t* __restrict__ touch, int N, int H, int W, const PostImg* __restrict__ img) {
  const int n = blockIdx.y;
  long base0 = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base0 = img[n].off; }
  const int per = 2 * W + 2 * H;  // top row, bottom row, left column, right column (corners twice: harmless)
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= per) return;
  int x, y;
  if (k < W) { x = k; y = 0; }
  else if (k < 2 * W) { x = k - W; y = H - 1; }
  else if (k < 2 * W + H) { x = 0; y = k - 2 * W; }
  else { x = W - 1; y = k - 2 * W - H; }
  const long i = base0 + (long)y * W + x;
  if (!bm[i]) touch[uf_find(L, (int)i)] = 1;  // background component reaches the frame
}
void launch_ccl(const uint8_t* bm, int* L, uint8_t* touch, int N, int H, int W, const PostImg* img, hipStream_t s) {
  // (images are the grid's y dimension, as in the kernels below: a batch stays far below its 65535 limit)
  hipLaunchKernelGGL(ccl_rows_kernel, dim3((unsigned)((H + 3) / 4), N), dim3(256), 0, s, bm, L, touch, N, H, W, img);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3((unsigned)((H + 3) / 4), N), dim3(256), 0, s, bm, L, N, H, W, img);
  hipLaunchKernelGGL(ccl_frame_kernel, dim3((unsigned)((2 * W + 2 * H + 255) / 256), N), dim3(256), 0, s, bm, L, touch, N, H, W, img);
}

// ------------------------------------------------------------------ 3. border starts, reference order
__device__ __forceinline__ bool is_start(const uint8_t* bm, const int* L, const uint8_t* touch, long base, int i) {
  const long g = base + i;
  if (L[g] != (int)g) return false;
  return bm[g] ? true : !touch[g];
}
// Ordered compaction in three small kernels so that every CU takes part (a single workgroup per image
// left 3/4 of the chip idle): per-chunk counts, a per-image scan of the chunk counts, ranked writes.
#define START_CHUNKS 128
__global__ void __launch_bounds__(256) starts_count_kernel(const uint8_t* __restrict__ bm, const int* __restrict__ L,
                                                           const uint8_t* __restrict__ touch, int H, int W,
                                                           int* __restrict__ chunk_cnt, const PostImg* __restrict__ img) {
  __shared__ int wsum[4];
  const int n = blockIdx.y, c = blockIdx.x;
  long base = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base = img[n].off; }
  const int per = H * W, chunk = (per + START_CHUNKS - 1) / START_CHUNKS;
  const int lo = c * chunk, hi = min(per, lo + chunk);
  int cnt = 0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) cnt += is_start(bm, L, touch, base, i) ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) chunk_cnt[n * START_CHUNKS + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ void __launch_bounds__(64) starts_scan_kernel(int* __restrict__ chunk_cnt, int max_cand, int* __restrict__ ncont_all,
                                                         int* __restrict__ ncont) {
  const int n = blockIdx.x;
  if (threadIdx.x != 0) return;
  int run = 0;
  for (int c = 0; c < START_CHUNKS; ++c) {
    const int v = chunk_cnt[n * START_CHUNKS + c];
    chunk_cnt[n * START_CHUNKS + c] = run;  // exclusive offset
    run += v;
  }
  ncont_all[n] = run;
  ncont[n] = run < max_cand ? run : max_cand;
}
__global__ void __launch_bounds__(256) starts_write_kernel(const uint8_t* __restrict__ bm, const int* __restrict__ L,
                                                           const uint8_t* __restrict__ touch, int H, int W, int max_cand,
                                                           const int* __restrict__ chunk_off, const int* __restrict__ ncont_all,
                                                           int* __restrict__ starts /*[N][max_cand]*/, const PostImg* __restrict__ img) {
  __shared__ int wsum[4];
  __shared__ int running;
  const int n = blockIdx.y, c = blockIdx.x;
  long base = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base = img[n].off; }
  const int per = H * W, chunk = (per + START_CHUNKS - 1) / START_CHUNKS;
  const int lo = c * chunk, hi = min(per, lo + chunk);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int total = ncont_all[n];
  if (threadIdx.x == 0) running = chunk_off[n * START_CHUNKS + c];
  __syncthreads();
  for (int i0 = lo; i0 < hi; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool st = i < hi && is_start(bm, L, touch, base, i);
    const unsigned long long bal = __ballot(st);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < wv; ++k) woff += wsum[k];
    const int rank = running + woff + before;  // raster-order rank = discovery order of the reference scan
    if (st) {
      const int pos = total - 1 - rank;        // findContours returns borders in reverse discovery order
      if (pos < max_cand) starts[(long)n * max_cand + pos] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) running += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
}
void launch_starts(const uint8_t* bm, const int* L, const uint8_t* touch, int N, int H, int W, int max_cand, int* ncont_all,
                   int* ncont, int* starts, int* chunk_cnt, const PostImg* img, hipStream_t s) {
  hipLaunchKernelGGL(starts_count_kernel, dim3(START_CHUNKS, N), dim3(256), 0, s, bm, L, touch, H, W, chunk_cnt, img);
  hipLaunchKernelGGL(starts_scan_kernel, dim3(N), dim3(64), 0, s, chunk_cnt, max_cand, ncont_all, ncont);
  hipLaunchKernelGGL(starts_write_kernel, dim3(START_CHUNKS, N), dim3(256), 0, s, bm, L, touch, H, W, max_cand, chunk_cnt,
                     ncont_all, starts, img);
}

// ------------------------------------------------------------------ 4. border following (icvFetchContour)
__constant__ int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__constant__ int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

#define KEY_OFF 16384
__device__ __forceinline__ unsigned long long make_key(int x, int y, unsigned idx) {
  return ((unsigned long long)(unsigned)(x + KEY_OFF) << 48) | ((unsigned long long)(unsigned)(y + KEY_OFF) << 32) | idx;
}
__device__ __forceinline__ int key_x(unsigned long long k) { return (int)(k >> 48) - KEY_OFF; }
__device__ __forceinline__ int key_y(unsigned long long k) { return (int)((k >> 32) & 0xffff) - KEY_OFF; }
__device__ __forceinline__ unsigned key_i(unsigned long long k) { return (unsigned)(k & 0xffffffffu); }

// 8-neighbour occupancy of (cx, cy) as a bit mask (bit d = direction d of kDx/kDy): the eight byte
// loads are independent, so one step of the border walk costs one memory round trip instead of up to
// eight dependent ones.
__device__ __forceinline__ unsigned nbr_mask(const uint8_t* __restrict__ bm, int H, int W, int cx, int cy) {
  unsigned m = 0;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    const int x = cx + kDx[d], y = cy + kDy[d];
    const bool in = x >= 0 && y >= 0 && x < W && y < H;
    const uint8_t v = bm[in ? (long)y * W + x : 0];
    m |= (in && v) ? (1u << d) : 0u;
  }
  return m;
}

// follows one border; emits CHAIN_APPROX_SIMPLE vertices (if keys != null) and returns their number
template <class NB>
__device__ int trace_border_t(NB nb, int H, int W, int ox, int oy, bool is_hole, unsigned long long* keys, int cap = INT_MAX) {
  int count = 0;
  int s_end, s;
  s_end = s = is_hole ? 0 : 4;
  // first neighbour clockwise from s_end: s = s_end-1, s_end-2, ... (mod 8), at most 8 probes
  const unsigned m0 = nb(ox, oy);
  int found = -1;
#pragma unroll
  for (int k = 1; k <= 8; ++k) {
    const int d = (s_end - k) & 7;
    if (found < 0 && (k == 8 || ((m0 >> d) & 1u))) found = d;  // k == 8 lands on s_end itself (loop exit s == s_end)
  }
  s = found;
  if (s == s_end && !((m0 >> s) & 1u)) {  // isolated pixel
    if (keys) keys[0] = make_key(ox, oy, 0);
    return 1;
  }
  if (s == s_end) {
    // the do-while stopped because s wrapped to s_end, but that neighbour is set: the reference treats
    // `s == s_end` as the single-pixel case regardless (icvFetchContour) — keep that behaviour
    if (keys) keys[0] = make_key(ox, oy, 0);
    return 1;
  }
  const int i1x = ox + kDx[s], i1y = oy + kDy[s];
  int cx = ox, cy = oy;
  int prev_s = s ^ 4;
  // a closed border of a W x H bitmap has at most 4*W*H steps; the bound keeps every lane finite
  const long max_steps = 4L * W * H + 16;
  for (long step = 0; step < max_steps; ++step) {
    const unsigned m = nb(cx, cy);
    // first set direction counter-clockwise after s: s+1, s+2, ... (the previous pixel guarantees a hit)
    const unsigned rot = ((m | (m << 8)) >> ((s + 1) & 7)) & 0xffu;
    const int j = rot ? __ffs(rot) - 1 : 7;
    s = (s + 1 + j) & 7;
    const int nx = cx + kDx[s], ny = cy + kDy[s];
    if (s != prev_s) {
      if (keys && count < cap) keys[count] = make_key(cx, cy, (unsigned)count);
      ++count;
      prev_s = s;
    }
    const bool last = (nx == ox && ny == oy && cx == i1x && cy == i1y);
    cx = nx;
    cy = ny;
    if (last) break;
    s = (s + 4) & 7;
  }
  return count;
}

__device__ int trace_border(const uint8_t* __restrict__ bm, int H, int W, int ox, int oy, bool is_hole,
                            unsigned long long* keys) {
  return trace_border_t([&](int cx, int cy) { return nbr_mask(bm, H, W, cx, cy); }, H, W, ox, oy, is_hole, keys);
}

// cross-lane traffic through global scratch goes past the (non-coherent) vector L1
#define GLD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// The bitmap of one image as bits in LDS, with a one-pixel zero frame: bit (x+1) of row (y+1).  A step of the
// border walk then costs a few LDS reads (~100 clocks) instead of a global round trip (~1 us).
struct LdsBits {
  const unsigned* bits;
  int stride;  // words per row
  __device__ __forceinline__ unsigned row3(int Y, int b) const {  // bits b, b+1, b+2 of row Y
    const unsigned* r = bits + Y * stride + (b >> 5);
    const unsigned long long v = ((unsigned long long)r[1] << 32) | r[0];
    return (unsigned)(v >> (b & 31)) & 7u;
  }
  __device__ __forceinline__ unsigned operator()(int cx, int cy) const {
    const unsigned up = row3(cy, cx), mid = row3(cy + 1, cx), dn = row3(cy + 2, cx);  // x-1..x+1 -> shifted bits cx..cx+2
    return ((mid >> 2) & 1u) | (((up >> 2) & 1u) << 1) | (((up >> 1) & 1u) << 2) | ((up & 1u) << 3) | ((mid & 1u) << 4) |
           ((dn & 1u) << 5) | (((dn >> 1) & 1u) << 6) | (((dn >> 2) & 1u) << 7);
  }
};
__host__ __device__ inline int lds_bits_stride(int W) { return (W + 2 + 31) / 32 + 1; }

__device__ __forceinline__ void contour_origin(const uint8_t* bm, long base, int start, int W, int& ox, int& oy, bool& hole) {
  hole = bm[base + start] == 0;
  const int sx = start % W, sy = start / W;
  ox = hole ? sx - 1 : sx;
  oy = sy;
}

// pass A: count vertices per border.  one lane per border.
__global__ void __launch_bounds__(64) trace_count_kernel(const uint8_t* __restrict__ bm, int H, int W, int max_cand,
                                                         const int* __restrict__ ncont, const int* __restrict__ starts,
                                                         int* __restrict__ npts, const PostImg* __restrict__ img) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= ncont[n]) return;
  long base = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base = img[n].off; }
  int ox, oy;
  bool hole;
  contour_origin(bm, base, starts[(long)n * max_cand + c], W, ox, oy, hole);
  npts[(long)n * max_cand + c] = trace_border(bm + base, H, W, ox, oy, hole, nullptr);
}
__device__ __forceinline__ int pow2_ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}
// per image: exclusive scan of pow2-rounded sizes -> offsets; overflow flag
__global__ void __launch_bounds__(64) trace_offsets_kernel(int max_cand, const int* __restrict__ ncont,
                                                           const int* __restrict__ npts, int* __restrict__ poff, int pool_cap,
                                                           int* __restrict__ status, int* __restrict__ pool_need) {
  const int n = blockIdx.x;
  if (threadIdx.x != 0) return;
  int off = 0;
  long need = 0;
  const int nc = ncont[n];
  for (int c = 0; c < nc; ++c) {
    const int sz = npts[(long)n * max_cand + c];
    const int p2 = sz > 2 ? pow2_ceil(sz) : 0;  // borders with <= 2 vertices are dropped by the reference
    need += p2;
    if (off + p2 > pool_cap) { atomicOr(status, POST_ERR_POOL); poff[(long)n * max_cand + c] = -1; continue; }
    poff[(long)n * max_cand + c] = off;
    off += p2;
  }
  pool_need[n] = need > INT_MAX ? INT_MAX : (int)need;  // the host grows the pool to this and runs the pass again
}
// pass B: store keys (padded with +inf to a power of two)
__global__ void __launch_bounds__(64) trace_store_kernel(const uint8_t* __restrict__ bm, int H, int W, int max_cand,
                                                         const int* __restrict__ ncont, const int* __restrict__ starts,
                                                         const int* __restrict__ npts, const int* __restrict__ poff,
                                                         unsigned long long* __restrict__ pool, int pool_cap, const PostImg* __restrict__ img) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= ncont[n]) return;
  const long ci = (long)n * max_cand + c;
  const int sz = npts[ci];
  if (sz <= 2 || poff[ci] < 0) return;
  long base = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base = img[n].off; }
  int ox, oy;
  bool hole;
  contour_origin(bm, base, starts[ci], W, ox, oy, hole);
  unsigned long long* keys = pool + (long)n * pool_cap + poff[ci];
  trace_border(bm + base, H, W, ox, oy, hole, keys);
  const int p2 = pow2_ceil(sz);
  for (int i = sz; i < p2; ++i) keys[i] = ~0ull;
}

// Border following with the image's bitmap packed into LDS (one workgroup per image; maps up to ~1.2 Mpixel
// fit the 160 KB).  One launch does what trace_count / trace_offsets / trace_store do with the global bitmap:
// count the vertices of every border, lay the borders out in the image's key pool, walk them again to store.
constexpr int kTraceThreads = 1024;  // the bitmap -> LDS fill is latency-bound: 16 waves keep 4x the loads in flight
__global__ void __launch_bounds__(kTraceThreads) trace_lds_kernel(const uint8_t* __restrict__ bm, int H, int W, int max_cand,
                                                        const int* __restrict__ ncont, const int* __restrict__ starts,
                                                        int* __restrict__ npts, int* __restrict__ poff,
                                                        unsigned long long* __restrict__ pool, int pool_cap,
                                                        int* __restrict__ iscratch, int slice_limit, int* __restrict__ status,
                                                        int* __restrict__ pool_need, const PostImg* __restrict__ img) {
  extern __shared__ unsigned s_bits[];
  const int n = blockIdx.x, tid = threadIdx.x;
  long base = (long)n * H * W;
  if (img) { H = img[n].h; W = img[n].w; base = img[n].off; }  // (LDS is sized for the largest image of the batch)
  const int stride = lds_bits_stride(W);
  const int nc = ncont[n];
  if (nc == 0) { if (tid == 0) pool_need[n] = 0; return; }
  for (int i = tid; i < (H + 2) * stride; i += kTraceThreads) {
    const int Y = i / stride, w = i - Y * stride;
    unsigned word = 0;
    if (Y >= 1 && Y <= H) {
      const uint8_t* row = bm + base + (long)(Y - 1) * W;
      // shifted bit position b = x + 1  ->  word w holds x = 32w - 1 .. 32w + 30:
      // bit 0 from the byte before the 32-byte group, bits 1..31 from the group's first 31 bytes
      const int x0 = 32 * w;
      if (x0 - 1 >= 0 && x0 - 1 < W && row[x0 - 1]) word |= 1u;
      if ((W & 15) == 0 && x0 + 32 <= W) {  // two 16-byte loads (rows are 16-byte aligned when W % 16 == 0)
        const uint4 q0 = *(const uint4*)(row + x0), q1 = *(const uint4*)(row + x0 + 16);
        const unsigned v[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int b = 4 * k + j + 1;
            if (b < 32 && ((v[k] >> (8 * j)) & 0xffu)) word |= 1u << b;
          }
      } else {
        for (int b = 1; b < 32; ++b) {
          const int x = x0 + b - 1;
          if (x < W && row[x]) word |= 1u << b;
        }
      }
    }
    s_bits[i] = word;
  }
  __syncthreads();
  const LdsBits nb{s_bits, stride};
  // sizes and pool offsets of the image's borders stay in LDS between the passes (behind the bitmap): the serial scan
  // and the second pass then read LDS instead of global values other lanes have just written
  int* s_npts = (int*)(s_bits + (H + 2) * stride);
  int* s_poff = s_npts + max_cand;
  // ONE walk per border: its vertices go to a provisional slice of the image's hull scratch (idle until the per-border
  // stage; 2 * pool_cap key slots shared evenly by the image's borders) while they are counted, and are copied into the
  // pool once the offsets are known.  A border that outgrows its slice is walked a second time, as before.
  unsigned long long* prov = (unsigned long long*)(iscratch + (long)n * pool_cap * 4);
  const long slice = 2L * pool_cap / nc;
  const int cap_each = slice > slice_limit ? slice_limit : (int)slice;
  for (int c = tid; c < nc; c += kTraceThreads) {
    const long ci = (long)n * max_cand + c;
    int ox, oy;
    bool hole;
    contour_origin(bm, base, starts[ci], W, ox, oy, hole);
    const int sz = trace_border_t(nb, H, W, ox, oy, hole, prov + (long)c * cap_each, cap_each);
    npts[ci] = sz;
    s_npts[c] = sz;
  }
  __threadfence_block();
  __syncthreads();
  if (tid == 0) {  // exclusive scan of the pow2-rounded sizes (trace_offsets_kernel)
    int off = 0;
    long need = 0;
    for (int c = 0; c < nc; ++c) {
      const int sz = s_npts[c];
      const int p2 = sz > 2 ? pow2_ceil(sz) : 0;  // borders with <= 2 vertices are dropped by the reference
      need += p2;
      if (off + p2 > pool_cap) { atomicOr(status, POST_ERR_POOL); s_poff[c] = -1; continue; }
      s_poff[c] = off;
      off += p2;
    }
    pool_need[n] = need > INT_MAX ? INT_MAX : (int)need;  // the host grows the pool to this and runs the pass again
  }
  __syncthreads();
  for (int c = 0; c < nc; ++c) {  // every border: the workgroup copies its keys (padded with +inf to a power of two)
    const int sz = s_npts[c], po = s_poff[c];
    if (tid == 0) poff[(long)n * max_cand + c] = po;
    if (sz <= 2 || po < 0 || sz > cap_each) continue;
    unsigned long long* keys = pool + (long)n * pool_cap + po;
    const unsigned long long* src = prov + (long)c * cap_each;
    const int p2 = pow2_ceil(sz);
    for (int i = tid; i < p2; i += kTraceThreads) keys[i] = i < sz ? GLD(&src[i]) : ~0ull;
  }
  for (int c = tid; c < nc; c += kTraceThreads) {  // the borders that outgrew their slice
    const int sz = s_npts[c], po = s_poff[c];
    if (sz <= 2 || po < 0 || sz <= cap_each) continue;
    const long ci = (long)n * max_cand + c;
    int ox, oy;
    bool hole;
    contour_origin(bm, base, starts[ci], W, ox, oy, hole);
    unsigned long long* keys = pool + (long)n * pool_cap + po;
    trace_border_t(nb, H, W, ox, oy, hole, keys);
    const int p2 = pow2_ceil(sz);
    for (int i = sz; i < p2; ++i) keys[i] = ~0ull;
  }
}

// ------------------------------------------------------------------ 5. per-border geometry (one wave each)
struct P2f { float x, y; };
struct RRect { float cx, cy, w, h, angle; };

template <typename T> __device__ __forceinline__ int sgn(T v) { return (v > 0) - (v < 0); }

// Sklansky scan over sorted points; X/Y accessors by sorted position
template <class Acc, class DotT>
__device__ int sklansky(const Acc& P, int start, int end, int* stack, int nsign, int sign2) {
  int incr = end > start ? 1 : -1;
  int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
  int stacksize = 3;
  if (start == end || (P.x(start) == P.x(end) && P.y(start) == P.y(end))) {
    stack[0] = start;
    return 1;
  }
  stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
  end += incr;
  while (pnext != end) {
    auto cury = P.y(pcur), nexty = P.y(pnext);
    auto by = nexty - cury;
    if (sgn(by) != nsign) {
      auto ax = P.x(pcur) - P.x(pprev);
      auto bx = P.x(pnext) - P.x(pcur);
      auto ay = cury - P.y(pprev);
      DotT convexity = (DotT)ay * bx - (DotT)ax * by;
      if (sgn(convexity) == sign2 && (ax != 0 || ay != 0)) {
        pprev = pcur; pcur = pnext; pnext += incr;
        stack[stacksize] = pnext;
        stacksize++;
      } else {
        if (pprev == start) {
          pcur = pnext; stack[1] = pcur; pnext += incr; stack[2] = pnext;
        } else {
          stack[stacksize - 2] = pnext;
          pcur = pprev;
          pprev = stack[stacksize - 4];
          stacksize--;
        }
      }
    } else {
      pnext += incr;
      stack[stacksize - 1] = pnext;
    }
  }
  return --stacksize;
}

// cv::convexHull(points, hull, clockwise=false, returnPoints) over SORTED keys (x, y, original index).
// Writes hull vertices (as floats) to hull[]; returns their number (or -1 on capacity overflow).
struct KeyAcc {
  const unsigned long long* k;
  __device__ int x(int i) const { return key_x(k[i]); }
  __device__ int y(int i) const { return key_y(k[i]); }
  __device__ int orig(int i) const { return (int)key_i(k[i]); }
};
template <class Acc, class DotT>
__device__ int convex_hull_sorted(const Acc& P, int total, int* stack, int* hullbuf, P2f* hull, int hull_cap) {
  if (total == 0) return 0;
  int miny_ind = 0, maxy_ind = 0;
  for (int i = 1; i < total; ++i) {
    auto y = P.y(i);
    if (P.y(miny_ind) > y) miny_ind = i;
    if (P.y(maxy_ind) < y) maxy_ind = i;
  }
  int nout = 0;
  if (P.x(0) == P.x(total - 1) && P.y(0) == P.y(total - 1)) {
    hullbuf[nout++] = P.orig(0);  // all points identical: index 0 after sort is the smallest original index
    hull[0].x = (float)P.x(0);
    hull[0].y = (float)P.y(0);
    return 1;
  }
  // NB: hullbuf holds ORIGINAL indices (the reference's `pointer[...] - data0`); hsorted (second half
  // of the buffer) remembers the sorted positions so coordinates can be fetched again afterwards.
  int* hsorted = hullbuf + total;
  int* tl_stack = stack;
  int tl_count = sklansky<Acc, DotT>(P, 0, maxy_ind, tl_stack, -1, 1);
  int* tr_stack = stack + tl_count;
  int tr_count = sklansky<Acc, DotT>(P, total - 1, maxy_ind, tr_stack, -1, -1);
  { int* t = tl_stack; tl_stack = tr_stack; tr_stack = t; int c = tl_count; tl_count = tr_count; tr_count = c; }  // !clockwise
  for (int i = 0; i < tl_count - 1; ++i) { hsorted[nout] = tl_stack[i]; hullbuf[nout++] = P.orig(tl_stack[i]); }
  for (int i = tr_count - 1; i > 0; --i) { hsorted[nout] = tr_stack[i]; hullbuf[nout++] = P.orig(tr_stack[i]); }
  const int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
  int* bl_stack = stack;
  int bl_count = sklansky<Acc, DotT>(P, 0, miny_ind, bl_stack, 1, -1);
  int* br_stack = stack + bl_count;
  int br_count = sklansky<Acc, DotT>(P, total - 1, miny_ind, br_stack, 1, 1);
  if (stop_idx >= 0) {
    const int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
    if (check_idx == stop_idx || (check_idx >= 0 && P.x(check_idx) == P.x(stop_idx) && P.y(check_idx) == P.y(stop_idx))) {
      bl_count = bl_count < 2 ? bl_count : 2;
      br_count = br_count < 2 ? br_count : 2;
    }
  }
  for (int i = 0; i < bl_count - 1; ++i) { hsorted[nout] = bl_stack[i]; hullbuf[nout++] = P.orig(bl_stack[i]); }
  for (int i = br_count - 1; i > 0; --i) { hsorted[nout] = br_stack[i]; hullbuf[nout++] = P.orig(br_stack[i]); }
  // cyclic shift so that original indices ascend/descend (only the start vertex changes)
  int shift = 0;
  if (nout >= 3) {
    int min_idx = 0, max_idx = 0, lt = 0, i;
    for (i = 1; i < nout; ++i) {
      const int idx = hullbuf[i];
      lt += hullbuf[i - 1] < idx;
      if (lt > 1 && lt <= i - 2) break;
      if (idx < hullbuf[min_idx]) min_idx = i;
      if (idx > hullbuf[max_idx]) max_idx = i;
    }
    const int mmdist = abs(max_idx - min_idx);
    if ((mmdist == 1 || mmdist == nout - 1) && (lt <= 1 || lt >= nout - 2)) {
      const int ascending = (max_idx + 1) % nout == min_idx;
      const int i0 = ascending ? min_idx : max_idx;
      int j = i0;
      if (i0 > 0) {
        for (i = 0; i < nout; ++i) {
          const int curr_idx = hullbuf[j];
          const int next_j = j + 1 < nout ? j + 1 : 0;
          const int next_idx = hullbuf[next_j];
          if (i < nout - 1 && (ascending != (curr_idx < next_idx))) break;
          j = next_j;
        }
        if (i == nout) shift = i0;
      }
    }
  }
  if (nout > hull_cap) return -1;
  for (int i = 0; i < nout; ++i) {
    const int sp = hsorted[(i + shift) % nout];
    hull[i].x = (float)P.x(sp);
    hull[i].y = (float)P.y(sp);
  }
  return nout;
}

// rotatingCalipers(CALIPERS_MINAREARECT); vect/inv are scratch of n entries
__device__ void rotating_calipers(const P2f* points, int n, P2f* vect, float* inv_vect_length, float out[6]) {
  float minarea = FLT_MAX;
  int left = 0, bottom = 0, right = 0, top = 0;
  int seq[4] = {-1, -1, -1, -1};
  float orientation = 0, base_a, base_b = 0;
  float left_x, right_x, top_y, bottom_y;
  P2f pt0 = points[0];
  left_x = right_x = pt0.x;
  top_y = bottom_y = pt0.y;
  for (int i = 0; i < n; ++i) {
    if (pt0.x < left_x) left_x = pt0.x, left = i;
    if (pt0.x > right_x) right_x = pt0.x, right = i;
    if (pt0.y > top_y) top_y = pt0.y, top = i;
    if (pt0.y < bottom_y) bottom_y = pt0.y, bottom = i;
    const P2f pt = points[(i + 1) < n ? i + 1 : 0];
    const double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
    vect[i].x = (float)dx;
    vect[i].y = (float)dy;
    inv_vect_length[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    pt0 = pt;
  }
  {
    double ax = vect[n - 1].x, ay = vect[n - 1].y;
    for (int i = 0; i < n; ++i) {
      const double bx = vect[i].x, by = vect[i].y;
      const double convexity = ax * by - ay * bx;
      if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
      ax = bx; ay = by;
    }
  }
  base_a = orientation;
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
  int best_left = 0, best_bottom = 0;
  float b_a = 0, b_w = 0, b_b = 0, b_h = 0;
  for (int k = 0; k < n; ++k) {
    float dp[4];
    dp[0] = +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y;
    dp[1] = -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y;
    dp[2] = -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y;
    dp[3] = +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y;
    float maxcos = dp[0] * inv_vect_length[seq[0]];
    int main_element = 0;
    for (int i = 1; i < 4; ++i) {
      const float cosalpha = dp[i] * inv_vect_length[seq[i]];
      if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
    }
    {
      const int pindex = seq[main_element];
      const float lead_x = vect[pindex].x * inv_vect_length[pindex];
      const float lead_y = vect[pindex].y * inv_vect_length[pindex];
      switch (main_element) {
        case 0: base_a = lead_x; base_b = lead_y; break;
        case 1: base_a = lead_y; base_b = -lead_x; break;
        case 2: base_a = -lead_x; base_b = -lead_y; break;
        default: base_a = -lead_y; base_b = lead_x; break;
      }
    }
    seq[main_element] += 1;
    seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
    {
      float dx = points[seq[1]].x - points[seq[3]].x;
      float dy = points[seq[1]].y - points[seq[3]].y;
      const float width = dx * base_a + dy * base_b;
      dx = points[seq[2]].x - points[seq[0]].x;
      dy = points[seq[2]].y - points[seq[0]].y;
      const float height = -dx * base_b + dy * base_a;
      const float area = width * height;
      if (area <= minarea) {
        minarea = area;
        best_left = seq[3];
        b_a = base_a; b_w = width; b_b = base_b; b_h = height;
        best_bottom = seq[0];
      }
    }
  }
  const float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
  const float C1 = A1 * points[best_left].x + points[best_left].y * B1;
  const float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
  const float idet = 1.f / (A1 * B2 - A2 * B1);
  const float px = (C1 * B2 - C2 * B1) * idet;
  const float py = (A1 * C2 - A2 * C1) * idet;
  out[0] = px; out[1] = py;
  out[2] = A1 * b_w; out[3] = B1 * b_w;
  out[4] = A2 * b_h; out[5] = B2 * b_h;
}

__device__ RRect min_area_rect_hull(const P2f* h, int n, P2f* vect, float* inv) {
  RRect box{0, 0, 0, 0, 0};
  if (n > 2) {
    float out[6];
    rotating_calipers(h, n, vect, inv, out);
    box.cx = out[0] + (out[2] + out[4]) * 0.5f;
    box.cy = out[1] + (out[3] + out[5]) * 0.5f;
    box.w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
    box.h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
    box.angle = (float)atan2((double)out[3], (double)out[2]);
  } else if (n == 2) {
    box.cx = (h[0].x + h[1].x) * 0.5f;
    box.cy = (h[0].y + h[1].y) * 0.5f;
    const double dx = h[1].x - h[0].x, dy = h[1].y - h[0].y;
    box.w = (float)sqrt(dx * dx + dy * dy);
    box.h = 0;
    box.angle = (float)atan2(dy, dx);
  } else if (n == 1) {
    box.cx = h[0].x;
    box.cy = h[0].y;
  }
  box.angle = (float)(box.angle * 180 / 3.14159265358979323846);
  return box;
}

__device__ void box_points(const RRect& r, P2f pt[4]) {
  const double ang = r.angle * 3.14159265358979323846 / 180.;
  const float b = (float)cos(ang) * 0.5f;
  const float a = (float)sin(ang) * 0.5f;
  pt[0].x = r.cx - a * r.h - b * r.w;
  pt[0].y = r.cy + b * r.h - a * r.w;
  pt[1].x = r.cx + a * r.h - b * r.w;
  pt[1].y = r.cy - b * r.h - a * r.w;
  pt[2].x = 2 * r.cx - pt[0].x;
  pt[2].y = 2 * r.cy - pt[0].y;
  pt[3].x = 2 * r.cx - pt[1].x;
  pt[3].y = 2 * r.cy - pt[1].y;
}

__device__ void get_mini_boxes(const RRect& box, float& ssid, P2f out[4]) {
  ssid = fmaxf(box.w, box.h);
  P2f a[4];
  box_points(box, a);
  for (int i = 1; i < 4; ++i) {  // std::sort on 4 elements == libstdc++ insertion sort
    const P2f v = a[i];
    if (v.x < a[0].x) {
      for (int k = i; k > 0; --k) a[k] = a[k - 1];
      a[0] = v;
    } else {
      int j = i - 1;
      while (v.x < a[j].x) { a[j + 1] = a[j]; --j; }
      a[j + 1] = v;
    }
  }
  P2f idx1, idx2, idx3, idx4;
  if (a[3].y <= a[2].y) { idx2 = a[3]; idx3 = a[2]; } else { idx2 = a[2]; idx3 = a[3]; }
  if (a[1].y <= a[0].y) { idx1 = a[1]; idx4 = a[0]; } else { idx1 = a[0]; idx4 = a[1]; }
  out[0] = idx1; out[1] = idx2; out[2] = idx3; out[3] = idx4;
}

// ---- fillPoly raster rule in closed form (XY_SHIFT = 16 edge fill + 8-connected Bresenham outline)
struct LineRast {  // one outline segment after cv::clipLine, LineIterator(leftToRight)
  int valid, px, py, dx, dy, delta_major, delta_minor, vert, count;
};
struct EdgeRast {  // one non-horizontal polygon edge of the scan fill
  int valid, y0, y1;
  long long x, dx;
};
__device__ bool clip_line(long long w, long long h, long long& x1, long long& y1, long long& x2, long long& y2) {
  const long long right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    long long a;
    if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (long long)((double)(a - y1) * (x2 - x1) / (y2 - y1)); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
    if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (long long)((double)(a - y2) * (x2 - x1) / (y2 - y1)); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) { a = c1 == 1 ? 0 : right; y1 += (long long)((double)(a - x1) * (y2 - y1) / (x2 - x1)); x1 = a; c1 = 0; }
      if (c2) { a = c2 == 1 ? 0 : right; y2 += (long long)((double)(a - x2) * (y2 - y1) / (x2 - x1)); x2 = a; c2 = 0; }
    }
  }
  return (c1 | c2) == 0;
}
__device__ LineRast make_line(int W, int H, int ax, int ay, int bx, int by) {
  LineRast L{};
  long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
  if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
      (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
    if (!clip_line(W, H, x1, y1, x2, y2)) return L;
  }
  int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
  int delta_x = 1, delta_y = 1;
  int px = (int)x1, py = (int)y1;
  if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }
  if (dy < 0) { dy = -dy; delta_y = -1; }
  const bool vert = dy > dx;
  if (vert) { int t = dx; dx = dy; dy = t; t = delta_x; delta_x = delta_y; delta_y = t; }
  L.valid = 1; L.px = px; L.py = py; L.dx = dx; L.dy = dy;
  L.delta_major = delta_x; L.delta_minor = delta_y; L.vert = vert; L.count = dx + 1;
  return L;
}
// is (x,y) one of the pixels LineIterator visits?  (closed form of the Bresenham error recurrence;
// all quantities fit 32 bits: |dx|,|dy|,j < 2^15)
__device__ __forceinline__ bool on_line(const LineRast& L, int x, int y) {
  if (!L.valid) return false;
  const int maj = L.vert ? y : x, mnr = L.vert ? x : y;
  const int maj0 = L.vert ? L.py : L.px, mnr0 = L.vert ? L.px : L.py;
  const int j = (maj - maj0) * L.delta_major;
  if (j < 0 || j >= L.count) return false;
  const int d = (mnr - mnr0) * L.delta_minor;  // minor offset this pixel would need
  if (d < 0 || d > L.dy) return false;
  // m_j = 0 if 2*dy*j - dx <= 0 else ceil((2*dy*j - dx) / (2*dx));  test m_j == d without dividing:
  const int a = 2 * L.dy * j - L.dx;
  if (a <= 0) return d == 0;
  // ceil(a / (2dx)) == d  <=>  (d-1)*2dx < a <= d*2dx
  return (long long)(d - 1) * 2 * L.dx < a && a <= (long long)d * 2 * L.dx;
}

// One polygon edge of the scan fill as CollectPolyEdges builds it (drawing.cpp), for a vertex pair already relative to
// the mask.  `shifted` selects the OpenCV rule (ocr_det_cfg.cv_compat):
//   false (OCR_CV_45, OpenCV 4.0 - 4.5.1): x in 16.16 at the vertices' integer x;
//   true  (OCR_CV_410, 4.5.2 and later):   an edge whose outline segment lies inside the mask moves right by half a pixel;
//         one whose segment had to be clipped is built from the CLIPPED integer end points (as far as cv::clipLine moved
//         them, whatever it returned) and extrapolated back to the edge's first row.
// The spans of a row are then [ (xa + up) >> 16, xb >> 16 ] with up = 65535 (classic: ceil) or 0 (shifted).
__device__ EdgeRast make_edge(int mw, int mh, int ax, int ay, int bx, int by, bool shifted) {
  EdgeRast e{};
  if (ay == by) return e;
  long long c0x = (long long)ax << 16, c0y = ay, c1x = (long long)bx << 16, c1y = by;
  if (shifted) {
    if ((unsigned)ax >= (unsigned)mw || (unsigned)bx >= (unsigned)mw || (unsigned)ay >= (unsigned)mh || (unsigned)by >= (unsigned)mh) {
      long long x1 = ax, y1 = ay, x2 = bx, y2 = by;
      (void)clip_line(mw, mh, x1, y1, x2, y2);
      if (y1 != y2) { c0x = x1 << 16; c0y = y1; c1x = x2 << 16; c1y = y2; }
    } else {
      c0x += 32768;
      c1x += 32768;
    }
  }
  e.valid = 1;
  e.dx = (c1x - c0x) / (c1y - c0y);
  if (ay < by) { e.y0 = ay; e.y1 = by; e.x = c0x + ((long long)ay - c0y) * e.dx; }
  else { e.y0 = by; e.y1 = ay; e.x = c1x + ((long long)by - c1y) * e.dx; }
  return e;
}

// BoxScoreFast: masked mean over the bbox with the mask of fillPoly(int-truncated corners).
// All 64 lanes participate; returns the score in every lane.
__device__ float box_score_fast_wave(const P2f arr[4], const float* __restrict__ pred, int H, int W, int lane, bool shifted) {
  float fxmin = fminf(fminf(arr[0].x, arr[1].x), fminf(arr[2].x, arr[3].x));
  float fxmax = fmaxf(fmaxf(arr[0].x, arr[1].x), fmaxf(arr[2].x, arr[3].x));
  float fymin = fminf(fminf(arr[0].y, arr[1].y), fminf(arr[2].y, arr[3].y));
  float fymax = fmaxf(fmaxf(arr[0].y, arr[1].y), fmaxf(arr[2].y, arr[3].y));
  auto clampi = [](int v, int lo, int hi) { return v > hi ? hi : (v < lo ? lo : v); };
  const int xmin = clampi((int)floorf(fxmin), 0, W - 1), xmax = clampi((int)ceilf(fxmax), 0, W - 1);
  const int ymin = clampi((int)floorf(fymin), 0, H - 1), ymax = clampi((int)ceilf(fymax), 0, H - 1);
  const int mw = xmax - xmin + 1, mh = ymax - ymin + 1;
  int vx[4], vy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { vx[i] = (int)arr[i].x - xmin; vy[i] = (int)arr[i].y - ymin; }
  LineRast Ls[4];
  EdgeRast Es[4];
  int nedges = 0;
  int ey_min = INT_MAX, ey_max = INT_MIN;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = (i + 3) & 3;  // previous vertex: edges run v[count-1] -> v[0] -> ...
    Ls[i] = make_line(mw, mh, vx[p], vy[p], vx[i], vy[i]);
    Es[i] = make_edge(mw, mh, vx[p], vy[p], vx[i], vy[i], shifted);
    if (Es[i].valid) {
      ++nedges;
      ey_min = min(ey_min, Es[i].y0);
      ey_max = max(ey_max, Es[i].y1);
    }
  }
  const bool do_fill = nedges >= 2;
  const long long up = shifted ? 0 : 65535;
  double sum = 0.0;
  int cnt = 0;
  // Row by row: the scan-fill spans of a row are worked out once per row, not once per pixel (the per-pixel form spent
  // 640 of the border stage's 950 us here: a 64-bit division for the pixel position, four 64-bit edge evaluations and a
  // sort for every pixel).  A lane group of W2 lanes owns a row; narrow boxes put 64 / W2 rows side by side.
  const int W2 = mw >= 64 ? 64 : pow2_ceil(mw), rows_per_iter = 64 / W2;
  const int lrow = lane / W2, lx = lane - lrow * W2;
  // An item = U pixels of one row per lane (x = xb + u * W2).  The U probability loads of item i+1 are issued before
  // item i's mask tests and are consumed through a select: a load under the `in` branch is one exposed memory round
  // trip per pixel (330 us of latency for a few thousand pixels per lane).
  constexpr int U = 8;
  const int nxb = (mw + U * W2 - 1) / (U * W2), nyb = (mh + rows_per_iter - 1) / rows_per_iter, nitems = nxb * nyb;
  auto fetch = [&](int item, float (&v)[U]) {
    const int yb = item / nxb, xb = (item - yb * nxb) * U * W2 + lx;
    int y = yb * rows_per_iter + lrow;
    y = y < mh ? y : mh - 1;  // lanes past the last row read a valid one and mask it off below
    const float* prow = pred + (long)(y + ymin) * W + xmin;
#pragma unroll
    for (int u = 0; u < U; ++u) { const int x = xb + u * W2; v[u] = prow[x < mw ? x : mw - 1]; }
  };
  float va[U], vb[U];
  if (nitems > 0) fetch(0, va);
  for (int item = 0; item < nitems; ++item) {
    if (item + 1 < nitems) fetch(item + 1, vb);
    const int yb = item / nxb, xb = (item - yb * nxb) * U * W2 + lx;
    const int y = yb * rows_per_iter + lrow;
    const bool yok = y < mh;
    int s1[2] = {1, 1}, s2[2] = {0, 0};  // filled spans [s1, s2] of this row (empty by default)
    if (do_fill && yok && y >= ey_min && y < ey_max) {
      // active edges on this scanline, their x in 16.16, sorted ascending, filled pairwise
      long long xs[4];
      int na = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (Es[i].valid && Es[i].y0 <= y && y < Es[i].y1) xs[na++] = Es[i].x + (long long)(y - Es[i].y0) * Es[i].dx;
      for (int i = 1; i < na; ++i) {
        const long long v = xs[i];
        int j = i - 1;
        while (j >= 0 && xs[j] > v) { xs[j + 1] = xs[j]; --j; }
        xs[j + 1] = v;
      }
      for (int i = 0; i + 1 < na; i += 2) {
        s1[i >> 1] = (int)((xs[i] + up) >> 16);
        s2[i >> 1] = (int)(xs[i + 1] >> 16);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int x = xb + u * W2;
      const bool ok = yok && x < mw;
      bool in = (x >= s1[0] && x <= s2[0]) || (x >= s1[1] && x <= s2[1]);  // interior first: most of the bbox, no outline test
      if (!in && ok) in = on_line(Ls[0], x, y) || on_line(Ls[1], x, y) || on_line(Ls[2], x, y) || on_line(Ls[3], x, y);
      in = in && ok;
      sum += in ? (double)va[u] : 0.0;  // + 0.0 leaves the sum as it is
      cnt += in;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) va[u] = vb[u];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sum += __shfl_xor(sum, off);
    cnt += __shfl_xor(cnt, off);
  }
  return cnt ? (float)(sum / (double)cnt) : 0.f;
}

// PolygonScoreAcc (score_mode "slow", /root/reference/src/postprocess_op.cpp:170-214): masked mean over the
// border's bounding box with mask = cv::fillPoly(border polygon).  The polygon has hundreds of
// vertices, so the mask is rasterised into a bit image in scratch instead of evaluated per pixel:
//   outline : lanes stride over the edges, each walks its (clipped) Bresenham line, atomicOr of bits
//   interior: counting sort of the scanline crossings by row (count / scan / scatter), then one lane
//             per row sorts its few crossings and sets the spans [ceil(x_even), floor(x_odd)]
//   mean    : lanes stride over the mask words.
// pts: the border's vertices in contour order as (x, y) keys.  scratch: this border's slice of the
// per-image word pool.  Returns the score in every lane, or -1 when the pool is exhausted.
__device__ float polygon_score_wave(const unsigned long long* __restrict__ pts, int npts, 
                                    const float* __restrict__ pred, int H, int W, unsigned* pool, unsigned pool_words,
                                    unsigned* pool_top, int lane, int* s_tmp /* >= 4 ints of LDS */, bool shifted) {
  // bounding box of the vertices (ints already; PolygonScoreAcc floors/ceils floats of ints)
  int bx0 = INT_MAX, bx1 = INT_MIN, by0 = INT_MAX, by1 = INT_MIN;
  for (int i = lane; i < npts; i += 64) {
    const int x = key_x(GLD(&pts[i])), y = key_y(GLD(&pts[i]));
    bx0 = min(bx0, x); bx1 = max(bx1, x); by0 = min(by0, y); by1 = max(by1, y);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    bx0 = min(bx0, __shfl_xor(bx0, off)); bx1 = max(bx1, __shfl_xor(bx1, off));
    by0 = min(by0, __shfl_xor(by0, off)); by1 = max(by1, __shfl_xor(by1, off));
  }
  auto clampi = [](int v, int lo, int hi) { return v > hi ? hi : (v < lo ? lo : v); };
  const int xmin = clampi(bx0, 0, W - 1), xmax = clampi(bx1, 0, W - 1);
  const int ymin = clampi(by0, 0, H - 1), ymax = clampi(by1, 0, H - 1);
  const int mw = xmax - xmin + 1, mh = ymax - ymin + 1;
  const int wpr = (mw + 31) >> 5;  // mask words per row
  // scratch carve-up: mask[mh*wpr] | cnt[mh+1] | crossings[2 words each]
  // crossings total = sum over edges of their clipped height (<= npts * mh, usually ~2*mh)
  // pass 0: count crossings
  long long total_cross = 0;
  for (int i = lane; i < npts; i += 64) {
    const int j = i == 0 ? npts - 1 : i - 1;
    const int y0 = key_y(GLD(&pts[j])) - ymin, y1 = key_y(GLD(&pts[i])) - ymin;
    if (y0 != y1) {
      const int lo = max(min(y0, y1), 0), hi = min(max(y0, y1), mh);
      if (hi > lo) total_cross += hi - lo;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) total_cross += __shfl_xor(total_cross, off);
  // whole 128-byte lines per border: no line of this scratch is shared with another wave
  const unsigned need = ((unsigned)mh * wpr + (unsigned)(mh + 1) + (unsigned)mh + 2u + (unsigned)(2 * total_cross) + 31u) & ~31u;
  if (lane == 0) {
    const unsigned base = atomicAdd(pool_top, need);
    s_tmp[0] = (base + need <= pool_words) ? (int)base : -1;
  }
  __syncthreads();
  const int base = s_tmp[0];
  __syncthreads();
  if (base < 0) return -1.0f;
  unsigned* mask = pool + base;
  int* cnt = (int*)(mask + (size_t)mh * wpr);     // [mh+1] counts -> offsets
  int* fillp = cnt + mh + 1;                      // [mh] scatter cursors
  long long* cross = (long long*)(((size_t)(fillp + mh) + 7) & ~(size_t)7);  // 8-byte aligned (2 spare words reserved)
  for (int i = lane; i < mh * wpr; i += 64) GST(&mask[i], 0u);
  for (int i = lane; i <= mh; i += 64) GST(&cnt[i], 0);
  __threadfence_block();
  __syncthreads();
  // ---- outline + crossing counts
  for (int i = lane; i < npts; i += 64) {
    const int j = i == 0 ? npts - 1 : i - 1;
    const int ax = key_x(GLD(&pts[j])) - xmin, ay = key_y(GLD(&pts[j])) - ymin, bx = key_x(GLD(&pts[i])) - xmin, by = key_y(GLD(&pts[i])) - ymin;
    const LineRast L = make_line(mw, mh, ax, ay, bx, by);
    if (L.valid) {
      int px = L.px, py = L.py, err = L.dx - 2 * L.dy;
      for (int k = 0; k < L.count; ++k) {
        atomicOr(&mask[(size_t)py * wpr + (px >> 5)], 1u << (px & 31));
        const bool stepm = err < 0;
        err += -2 * L.dy + (stepm ? 2 * L.dx : 0);
        if (L.vert) { py += L.delta_major; if (stepm) px += L.delta_minor; }
        else { px += L.delta_major; if (stepm) py += L.delta_minor; }
      }
    }
    if (ay != by) {
      const int lo = max(min(ay, by), 0), hi = min(max(ay, by), mh);
      for (int y = lo; y < hi; ++y) atomicAdd(&cnt[y], 1);
    }
  }
  __threadfence_block();
  __syncthreads();
  if (lane == 0) {  // exclusive scan of the per-row counts (mh <= image height)
    int run = 0;
    for (int y = 0; y < mh; ++y) { const int c = GLD(&cnt[y]); GST(&cnt[y], run); GST(&fillp[y], run); run += c; }
    GST(&cnt[mh], run);
  }
  __threadfence_block();
  __syncthreads();
  // ---- scatter crossings (x in 16.16 at each scanline the edge covers)
  for (int i = lane; i < npts; i += 64) {
    const int j = i == 0 ? npts - 1 : i - 1;
    const int ax = key_x(GLD(&pts[j])) - xmin, ay = key_y(GLD(&pts[j])) - ymin, bx = key_x(GLD(&pts[i])) - xmin, by = key_y(GLD(&pts[i])) - ymin;
    if (ay == by) continue;
    const EdgeRast e = make_edge(mw, mh, ax, ay, bx, by, shifted);
    const int lo = max(e.y0, 0), hi = min(e.y1, mh);
    for (int y = lo; y < hi; ++y) GST(&cross[atomicAdd(&fillp[y], 1)], e.x + (long long)(y - e.y0) * e.dx);
  }
  __threadfence_block();
  __syncthreads();
  // ---- spans: one lane per row
  for (int y = lane; y < mh; y += 64) {
    const int b = GLD(&cnt[y]), e = GLD(&cnt[y + 1]);
    for (int i = b + 1; i < e; ++i) {  // insertion sort (a handful of crossings per row; this lane owns [b, e))
      const long long v = GLD(&cross[i]);
      int k = i - 1;
      while (k >= b) {
        const long long u = GLD(&cross[k]);
        if (u <= v) break;
        GST(&cross[k + 1], u);
        --k;
      }
      GST(&cross[k + 1], v);
    }
    for (int i = b; i + 1 < e; i += 2) {
      int x1 = (int)((GLD(&cross[i]) + (shifted ? 0 : 65535)) >> 16), x2 = (int)(GLD(&cross[i + 1]) >> 16);
      if (x1 < mw && x2 >= 0) {
        x1 = max(x1, 0);
        x2 = min(x2, mw - 1);
        for (int x = x1; x <= x2; ++x) atomicOr(&mask[(size_t)y * wpr + (x >> 5)], 1u << (x & 31));
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // ---- masked mean
  double sum = 0.0;
  int n = 0;
  for (int i = lane; i < mh * wpr; i += 64) {
    unsigned m = GLD(&mask[i]);
    const int y = i / wpr, xb = (i - y * wpr) << 5;
    while (m) {
      const int b = __ffs(m) - 1;
      m &= m - 1;
      sum += (double)pred[(long)(y + ymin) * W + xb + b + xmin];
      ++n;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { sum += __shfl_xor(sum, off); n += __shfl_xor(n, off); }
  return n ? (float)(sum / (double)n) : 0.f;
}

// ---- ClipperOffset (jtRound, etClosedPolygon) + union clean-up, lane 0 only
struct IPt { long long X, Y; };
__device__ __forceinline__ long long clip_round(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }
// the round join's step count and rotation (ClipperOffset::DoOffset, /root/reference/src/clipper.cpp:3800-3818): ocml's f64 acos / sin / cos
// where the reference calls the host library's; ocr_selftest_unclip_trig hands these three values out so that a test can count
// the deltas on which the two libraries differ BEFORE Round() hides it
__device__ __forceinline__ void clipper_round_setup(double delta, double& steps, double& m_sin, double& m_cos) {
  const double pi = 3.141592653589793238, two_pi = pi * 2;
  double y = 0.25;
  if (0.25 > fabs(delta) * 0.25) y = fabs(delta) * 0.25;
  steps = pi / acos(1 - y / fabs(delta));
  if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
  m_sin = sin(two_pi / steps);
  m_cos = cos(two_pi / steps);
  if (delta < 0.0) m_sin = -m_sin;
}
// dest: scratch of `cap` points; returns point count (0 when Execute yields no path, -1 on overflow)
__device__ int clipper_offset_round(const IPt in[4], double delta, IPt* dest, int cap) {
  IPt c[4];
  int highI = 3;
  while (highI > 0 && in[0].X == in[highI].X && in[0].Y == in[highI].Y) highI--;
  int len = 0;
  c[len++] = in[0];
  for (int i = 1; i <= highI; ++i)
    if (c[len - 1].X != in[i].X || c[len - 1].Y != in[i].Y) c[len++] = in[i];
  if (len < 3) return 0;
  {
    double a = 0;
    for (int i = 0, j = len - 1; i < len; ++i) { a += ((double)c[j].X + c[i].X) * ((double)c[j].Y - c[i].Y); j = i; }
    if (!(-a * 0.5 >= 0)) {
      for (int i = 0; i < len / 2; ++i) { IPt t = c[i]; c[i] = c[len - 1 - i]; c[len - 1 - i] = t; }
    }
  }
  int nd = 0;
  const double pi = 3.141592653589793238, two_pi = pi * 2;
  if (delta > -1.0e-20 && delta < 1.0e-20) {
    for (int i = 0; i < len; ++i) dest[nd++] = c[i];
  } else {
    double steps, m_sin, m_cos;
    clipper_round_setup(delta, steps, m_sin, m_cos);
    const double steps_per_rad = steps / two_pi;
    double nX[4], nY[4];
    for (int j = 0; j < len; ++j) {
      const IPt& p1 = c[j];
      const IPt& p2 = c[(j + 1) % len];
      if (p1.X == p2.X && p1.Y == p2.Y) { nX[j] = 0; nY[j] = 0; continue; }
      double Dx = (double)(p2.X - p1.X), dy = (double)(p2.Y - p1.Y);
      const double f = 1 * 1.0 / sqrt(Dx * Dx + dy * dy);
      Dx *= f; dy *= f;
      nX[j] = dy; nY[j] = -Dx;
    }
    int k = len - 1;
    for (int j = 0; j < len; ++j) {
      double sinA = nX[k] * nY[j] - nX[j] * nY[k];
      if (fabs(sinA * delta) < 1.0) {
        const double cosA = nX[k] * nX[j] + nY[j] * nY[k];
        if (cosA > 0) {
          if (nd >= cap) return -1;
          dest[nd++] = {clip_round(c[j].X + nX[k] * delta), clip_round(c[j].Y + nY[k] * delta)};
          continue;
        }
      } else if (sinA > 1.0) sinA = 1.0;
      else if (sinA < -1.0) sinA = -1.0;
      if (sinA * delta < 0) {
        if (nd + 3 > cap) return -1;
        dest[nd++] = {clip_round(c[j].X + nX[k] * delta), clip_round(c[j].Y + nY[k] * delta)};
        dest[nd++] = c[j];
        dest[nd++] = {clip_round(c[j].X + nX[j] * delta), clip_round(c[j].Y + nY[j] * delta)};
      } else {
        const double a = atan2(sinA, nX[k] * nX[j] + nY[k] * nY[j]);
        long long r = clip_round(steps_per_rad * fabs(a));
        const int st = (int)r > 1 ? (int)r : 1;
        if (nd + st + 1 > cap) return -1;
        double X = nX[k], Y = nY[k], X2;
        for (int i = 0; i < st; ++i) {
          dest[nd++] = {clip_round(c[j].X + X * delta), clip_round(c[j].Y + Y * delta)};
          X2 = X;
          X = X * m_cos - m_sin * Y;
          Y = X2 * m_sin + Y * m_cos;
        }
        dest[nd++] = {clip_round(c[j].X + nX[j] * delta), clip_round(c[j].Y + nY[j] * delta)};
      }
      k = j;
    }
  }
  // union clean-up: duplicates / collinear vertices go; ring starts after the last top-most vertex
  bool changed = true;
  while (changed && nd >= 3) {
    changed = false;
    int i = 0;
    while (i < nd && nd >= 3) {
      const IPt a = dest[(i + nd - 1) % nd], b = dest[i], cc = dest[(i + 1) % nd];
      if ((b.X == cc.X && b.Y == cc.Y) || (b.X == a.X && b.Y == a.Y) ||
          (b.Y - a.Y) * (cc.X - b.X) == (b.X - a.X) * (cc.Y - b.Y)) {
        for (int q = i; q + 1 < nd; ++q) dest[q] = dest[q + 1];
        --nd;
        changed = true;
        if (i > 0) --i;
      } else
        ++i;
    }
  }
  if (nd < 3) return 0;
  {  // non-positive winding (ring folded over itself at tiny delta) is not part of a pftPositive union
    double a = 0;
    for (int i = 0, j = nd - 1; i < nd; ++i) { a += ((double)dest[j].X + dest[i].X) * ((double)dest[j].Y - dest[i].Y); j = i; }
    if (!(-a * 0.5 > 0)) return 0;
  }
  long long miny = dest[0].Y;
  for (int i = 1; i < nd; ++i) miny = dest[i].Y < miny ? dest[i].Y : miny;
  int e = 0;
  for (int i = 0; i < nd; ++i)
    if (dest[i].Y == miny && dest[(i + 1) % nd].Y != miny) { e = i; break; }
  // rotate left by e+1 (in place, via reversal)
  const int rsh = (e + 1) % nd;
  auto rev = [&](int lo, int hi) { while (lo < hi) { IPt t = dest[lo]; dest[lo] = dest[hi]; dest[hi] = t; ++lo; --hi; } };
  if (rsh) { rev(0, rsh - 1); rev(rsh, nd - 1); rev(0, nd - 1); }
  return nd;
}

struct FKeyAcc {  // float-valued integer points sorted through the same key packing
  const unsigned long long* k;
  __device__ float x(int i) const { return (float)key_x(k[i]); }
  __device__ float y(int i) const { return (float)key_y(k[i]); }
  __device__ int orig(int i) const { return (int)key_i(k[i]); }
};


// bitonic sort of n2 (power of two) keys by the whole wave; `a` may be LDS or global
__device__ void bitonic_sort_wave(unsigned long long* a, int n2, int lane, bool in_lds) {
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < n2; i += 64) {
        const int l = i ^ j;
        if (l > i) {
          unsigned long long x, y;
          if (in_lds) { x = a[i]; y = a[l]; }
          else {
            x = __hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y = __hip_atomic_load(&a[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          const bool asc = (i & k) == 0;
          if (asc ? (x > y) : (x < y)) {
            if (in_lds) { a[i] = y; a[l] = x; }
            else {
              __hip_atomic_store(&a[i], y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&a[l], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
      }
      __threadfence_block();
      __syncthreads();
    }
  }
}

// DBPostProcessor::UnClip (/root/reference/src/postprocess_op.cpp:39-72) of one box by one lane: distance from the float shoelace
// area and perimeter, the four int-truncated corners through ClipperOffset(jtRound, etClosedPolygon), cv::minAreaRect of what
// Execute returns (an empty solution: RotatedRect(Point2f(0, 0), Size2f(1, 1), 0)).  The caller's LDS scratch: `unclip` / `ukeys`
// UNCLIP_CAP entries, `ustack` UNCLIP_CAP + 2, `uhull` 2 * UNCLIP_CAP, `hull` / `vect` / `inv` HULL_CAP.  Returns 0 or the
// POST_ERR_* bit of the scratch that overflowed; *npoly (optional) = vertices of the offset polygon.
struct UnclipScratch {
  IPt* unclip;
  unsigned long long* ukeys;
  int* ustack;
  int* uhull;
  P2f* hull;
  P2f* vect;
  float* inv;
};
template <int HULL_CAP, int UNCLIP_CAP>
__device__ int unclip_min_rect(const P2f arr[4], float unclip_ratio, const UnclipScratch& S, RRect& pts, int* npoly) {
  float area = 0.0f, dist = 0.0f;
  for (int i = 0; i < 4; ++i) {
    const int j = (i + 1) % 4;
    area += arr[i].x * arr[j].y - arr[i].y * arr[j].x;
    dist += sqrtf((arr[i].x - arr[j].x) * (arr[i].x - arr[j].x) + (arr[i].y - arr[j].y) * (arr[i].y - arr[j].y));
  }
  area = fabsf((float)(area / 2.0));
  const float distance = area * unclip_ratio / dist;
  IPt q[4];
  for (int i = 0; i < 4; ++i) q[i] = {(long long)(int)arr[i].x, (long long)(int)arr[i].y};
  const int un = clipper_offset_round(q, (double)distance, S.unclip, UNCLIP_CAP);
  if (npoly) *npoly = un;
  if (un < 0) return POST_ERR_UNCLIP;
  if (un == 0) {
    pts = RRect{0, 0, 1, 1, 0};
    return 0;
  }
  // minAreaRect(points): sort (x, y, index) by insertion, hull, calipers
  for (int i = 0; i < un; ++i) {
    const unsigned long long k = make_key((int)S.unclip[i].X, (int)S.unclip[i].Y, (unsigned)i);
    int j = i - 1;
    while (j >= 0 && S.ukeys[j] > k) { S.ukeys[j + 1] = S.ukeys[j]; --j; }
    S.ukeys[j + 1] = k;
  }
  FKeyAcc FP{S.ukeys};
  const int hn = convex_hull_sorted<FKeyAcc, float>(FP, un, S.ustack, S.uhull, S.hull, HULL_CAP);
  if (hn < 0) return POST_ERR_HULL;
  pts = min_area_rect_hull(S.hull, hn, S.vect, S.inv);
  return 0;
}

// one wave (workgroup of 64) per border.  The LDS working set decides how many borders a CU holds at once
// (the work per border is a chain of dependent single-lane steps, so that is the throughput):
// <512, 512, 256> is 23 KB - six borders per CU - and covers every border of a 960x960 map short of a
// convex lattice polygon with more than 512 corners; launch_post_large re-runs the stage with
// <4096, 1024, 512> (70 KB) when a border overflowed one of them (POST_ERR_HULL / POST_ERR_UNCLIP).
template <int SORT_LDS_CAP, int HULL_CAP, int UNCLIP_CAP, bool RETRY>
__global__ void __launch_bounds__(64) border_box_kernel(const PostArgs a) {
  __shared__ unsigned long long s_keys[SORT_LDS_CAP];
  __shared__ P2f s_hull[HULL_CAP];
  // one region, two lives: the calipers' scratch (edge vectors, inverse lengths) while a hull is measured,
  // the unclip polygon and its sort keys in between (each is dead when the other is written)
  // (third life: the contour hull's Sklansky stack and index buffer, 4 ints per sorted key, when the sort ran in LDS -
  // in global memory every stack pop of the single-lane scan is a dependent L2 round trip)
  constexpr int kCal = HULL_CAP * (int)(sizeof(P2f) + sizeof(float)), kUn = UNCLIP_CAP * (int)(sizeof(IPt) + sizeof(unsigned long long));
  constexpr int kStk = RETRY ? 0 : SORT_LDS_CAP * 4 * (int)sizeof(int), kCU = kCal > kUn ? kCal : kUn;  // (the retry pass: rare, large contours)
  __shared__ __attribute__((aligned(16))) unsigned char s_region[kCU > kStk ? kCU : kStk];
  P2f* s_vect = (P2f*)s_region;
  float* s_inv = (float*)(s_region + HULL_CAP * sizeof(P2f));
  IPt* s_unclip = (IPt*)s_region;
  unsigned long long* s_ukeys = (unsigned long long*)(s_region + UNCLIP_CAP * sizeof(IPt));
  __shared__ int s_ustack[UNCLIP_CAP + 2];
  __shared__ int s_uhull[2 * UNCLIP_CAP];
  __shared__ float s_box[8];
  __shared__ int s_flag;
  const int n = blockIdx.y, c = blockIdx.x, lane = threadIdx.x;
  const long ci = (long)n * a.max_cand + c;
  int* obox = a.cand_boxes + ci * 8;
  if (c >= a.ncont[n]) return;
  // the image's own map size / source size / ratios in a batch of mixed sizes
  int g_H = a.H, g_W = a.W, g_src_h = a.src_h, g_src_w = a.src_w;
  float g_ratio_h = a.ratio_h, g_ratio_w = a.ratio_w;
  const float* g_pred = a.pred + (long)n * a.H * a.W;
  if (a.img) {
    const PostImg im = a.img[n];
    g_H = im.h; g_W = im.w; g_src_h = im.src_h; g_src_w = im.src_w; g_ratio_h = im.ratio_h; g_ratio_w = im.ratio_w;
    g_pred = a.pred + im.poff;
  }
  if (lane == 0) a.cand_valid[ci] = 0;
  const int total = a.npts[ci];
  if (total <= 2 || a.poff[ci] < 0) return;  // contours[i].size() <= 2 -> continue
  const int n2 = pow2_ceil(total);
  unsigned long long* gkeys = a.pool + (long)n * a.pool_cap + a.poff[ci];
  unsigned long long* keys;
  const bool in_lds = n2 <= SORT_LDS_CAP;
  if (in_lds) {
    for (int i = lane; i < n2; i += 64) s_keys[i] = gkeys[i];
    keys = s_keys;
  } else {
    keys = gkeys;
  }
  __syncthreads();
  bitonic_sort_wave(keys, n2, lane, in_lds);
  if (OCR_PROBE_STOP(a, 1)) return;
  // ---- minAreaRect(contour) -> GetMiniBoxes (lane 0)
  if (lane == 0) {
    s_flag = 0;
    // 4 ints per key: stack[n2+2 <= 2*n2], hullbuf[2*n2]
    int* stack = in_lds && !RETRY ? (int*)s_region : a.iscratch + ((long)n * a.pool_cap + a.poff[ci]) * 4;
    int* hullbuf = stack + 2 * n2;
    KeyAcc P{keys};
    const int hn = convex_hull_sorted<KeyAcc, int>(P, total, stack, hullbuf, s_hull, HULL_CAP);
    if (hn < 0) { atomicOr(a.status, POST_ERR_HULL); s_flag = 0; }
    else if (OCR_PROBE_STOP(a, 2)) { s_flag = 0; }
    else {
      const RRect box = min_area_rect_hull(s_hull, hn, s_vect, s_inv);
      float ssid;
      P2f arr[4];
      get_mini_boxes(box, ssid, arr);
      if (ssid >= 3.0f) {  // min_size = 3
        s_flag = 1;
        for (int i = 0; i < 4; ++i) { s_box[2 * i] = arr[i].x; s_box[2 * i + 1] = arr[i].y; }
      }
    }
  }
  __syncthreads();
  if (!s_flag || OCR_PROBE_STOP(a, 3)) return;
  P2f arr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { arr[i].x = s_box[2 * i]; arr[i].y = s_box[2 * i + 1]; }
  // ---- box score (whole wave)
  float score;
  if (a.slow) {
    // PolygonScoreAcc needs the vertices in contour order: the unsorted keys are still in the pool when
    // the sort ran in LDS; after an in-place (global) sort restore the order by sorting on the index
    if (!in_lds || RETRY) {  // (a first pass that sorted in place may have stopped before restoring the order)
      for (int i = lane; i < n2; i += 64) { const unsigned long long k = GLD(&gkeys[i]); GST(&gkeys[i], (k << 32) | (k >> 32)); }
      __threadfence_block();
      __syncthreads();
      bitonic_sort_wave(gkeys, n2, lane, false);
      for (int i = lane; i < n2; i += 64) { const unsigned long long k = GLD(&gkeys[i]); GST(&gkeys[i], (k << 32) | (k >> 32)); }
      __threadfence_block();
      __syncthreads();
    }
    score = polygon_score_wave(gkeys, total, g_pred, g_H, g_W, a.mask_pool + (size_t)n * a.mask_pool_words,
                               a.mask_pool_words, a.mask_pool_top + n, lane, &s_flag, a.fill_shifted != 0);
    if (score < 0.f) { if (lane == 0) atomicOr(a.status, POST_ERR_POOL); return; }
  } else {
    score = box_score_fast_wave(arr, g_pred, g_H, g_W, lane, a.fill_shifted != 0);
  }
  if (score < a.box_thresh) return;
  if (lane != 0) return;
  if (OCR_PROBE_STOP(a, 4)) return;
  // ---- UnClip
  RRect pts;
  {
    const UnclipScratch us{s_unclip, s_ukeys, s_ustack, s_uhull, s_hull, s_vect, s_inv};
    const int ue = unclip_min_rect<HULL_CAP, UNCLIP_CAP>(arr, a.unclip_ratio, us, pts, nullptr);
    if (ue) { atomicOr(a.status, ue); return; }
  }
  if (OCR_PROBE_STOP(a, 5)) return;
  if (pts.h < 1.001 && pts.w < 1.001) return;
  float ssid;
  P2f clip[4];
  get_mini_boxes(pts, ssid, clip);
  if (ssid < 5.0f) return;  // min_size + 2
  int b[4][2];
  const float fw = (float)g_W, fh = (float)g_H;
  for (int k = 0; k < 4; ++k) {
    float vx = roundf(clip[k].x / fw * fw), vy = roundf(clip[k].y / fh * fh);
    vx = vx > fw ? fw : (vx < 0 ? 0 : vx);
    vy = vy > fh ? fh : (vy < 0 ? 0 : vy);
    b[k][0] = (int)vx;
    b[k][1] = (int)vy;
  }
  // ---- FilterTagDetRes for this box (OrderPointsClockwise, /ratio, clamp, size gate)
  for (int i = 1; i < 4; ++i) {
    const int vx = b[i][0], vy = b[i][1];
    if (vx < b[0][0]) {
      for (int k = i; k > 0; --k) { b[k][0] = b[k - 1][0]; b[k][1] = b[k - 1][1]; }
      b[0][0] = vx; b[0][1] = vy;
    } else {
      int j = i - 1;
      while (vx < b[j][0]) { b[j + 1][0] = b[j][0]; b[j + 1][1] = b[j][1]; --j; }
      b[j + 1][0] = vx; b[j + 1][1] = vy;
    }
  }
  int l0 = 0, l1 = 1, r0 = 2, r1 = 3;
  if (b[l0][1] > b[l1][1]) { const int t = l0; l0 = l1; l1 = t; }
  if (b[r0][1] > b[r1][1]) { const int t = r0; r0 = r1; r1 = t; }
  const int ord[4] = {l0, r0, r1, l1};
  int f[4][2];
  for (int m = 0; m < 4; ++m) {
    int x = (int)((float)b[ord[m]][0] / g_ratio_w);
    int y = (int)((float)b[ord[m]][1] / g_ratio_h);
    x = min(max(x, 0), g_src_w - 1);
    y = min(max(y, 0), g_src_h - 1);
    f[m][0] = x; f[m][1] = y;
  }
  const double dw0 = (double)(f[0][0] - f[1][0]), dw1 = (double)(f[0][1] - f[1][1]);
  const double dh0 = (double)(f[0][0] - f[3][0]), dh1 = (double)(f[0][1] - f[3][1]);
  const int rect_width = (int)sqrt(dw0 * dw0 + dw1 * dw1);
  const int rect_height = (int)sqrt(dh0 * dh0 + dh1 * dh1);
  if (rect_width <= 4 || rect_height <= 4) return;
  for (int m = 0; m < 4; ++m) { obox[2 * m] = f[m][0]; obox[2 * m + 1] = f[m][1]; }
  a.cand_valid[ci] = 1;
}

// ------------------------------------------------------------------ 6. ordered compaction of surviving boxes
__global__ void __launch_bounds__(64) boxes_compact_kernel(const PostArgs a, int* __restrict__ out_boxes, int cap,
                                                           int* __restrict__ out_n) {
  const int n = blockIdx.x;
  if (threadIdx.x != 0) return;
  int k = 0;
  const int nc = a.ncont[n];
  for (int c = 0; c < nc; ++c) {
    const long ci = (long)n * a.max_cand + c;
    if (!a.cand_valid[ci]) continue;
    if (k < cap)
      for (int q = 0; q < 8; ++q) out_boxes[((long)n * cap + k) * 8 + q] = a.cand_boxes[ci * 8 + q];
    ++k;
  }
  out_n[n] = k;
}

void launch_post(const PostArgs& a, int N, int* out_boxes, int cap, int* out_n, hipStream_t s) {
  launch_ccl(a.bitmap, a.labels, a.touch, N, a.H, a.W, a.img, s);
  launch_starts(a.bitmap, a.labels, a.touch, N, a.H, a.W, a.max_cand, a.ncont_all, a.ncont, a.starts, a.chunk_cnt, a.img, s);
  const dim3 gl((a.max_cand + 63) / 64, N);
  const size_t lds = ((size_t)(a.H + 2) * lds_bits_stride(a.W) + 2 * (size_t)a.max_cand) * sizeof(unsigned);  // bitmap bits + sizes + offsets
  bool in_lds = lds <= 150 * 1024;  // 960x960: 119 KB
  if (in_lds && lds > 64 * 1024) {  // more than the default dynamic LDS limit: raise it, per device
    static LdsAttrMemo attr_state;
    in_lds = raise_dynamic_lds((const void*)trace_lds_kernel, 150 * 1024, attr_state);  // refused: the global-bitmap kernels below
  }
  if (in_lds) {
    // OCR_TRACE_SLICE (tests): cap on a border's provisional slice, to drive the second-walk path with small inputs
    const int slice_limit = rt_options().trace_slice > 0 ? rt_options().trace_slice : INT_MAX;
    hipLaunchKernelGGL(trace_lds_kernel, dim3(N), dim3(kTraceThreads), lds, s, a.bitmap, a.H, a.W, a.max_cand, a.ncont, a.starts, a.npts, a.poff,
                       a.pool, a.pool_cap, a.iscratch, slice_limit, a.status, a.pool_need, a.img);
  } else {
    hipLaunchKernelGGL(trace_count_kernel, gl, dim3(64), 0, s, a.bitmap, a.H, a.W, a.max_cand, a.ncont, a.starts, a.npts, a.img);
    hipLaunchKernelGGL(trace_offsets_kernel, dim3(N), dim3(64), 0, s, a.max_cand, a.ncont, a.npts, a.poff, a.pool_cap, a.status, a.pool_need);
    hipLaunchKernelGGL(trace_store_kernel, gl, dim3(64), 0, s, a.bitmap, a.H, a.W, a.max_cand, a.ncont, a.starts, a.npts, a.poff,
                       a.pool, a.pool_cap, a.img);
  }
  hipLaunchKernelGGL((border_box_kernel<512, 512, 256, false>), dim3(a.max_cand, N), dim3(64), 0, s, a);
  hipLaunchKernelGGL(boxes_compact_kernel, dim3(N), dim3(64), 0, s, a, out_boxes, cap, out_n);
}
// second try of the per-border stage with the large LDS working set (the traced borders are still in the pool)
void launch_post_large(const PostArgs& a, int N, int* out_boxes, int cap, int* out_n, hipStream_t s) {
  hipLaunchKernelGGL((border_box_kernel<4096, 1024, 512, true>), dim3(a.max_cand, N), dim3(64), 0, s, a);
  hipLaunchKernelGGL(boxes_compact_kernel, dim3(N), dim3(64), 0, s, a, out_boxes, cap, out_n);
}

// ------------------------------------------------------------------ self-test taps (ocr_selftest_unclip*, tests only)
// The device's ClipperOffset and UnClip in front of vectors held outside the library (the reference's compiled clipper.cpp:
// tests/golden/unclip_ref.json): one workgroup of 64 per case, lane 0 works - exactly how border_box_kernel runs them.
__global__ void __launch_bounds__(64) selftest_clipper_kernel(const int* __restrict__ quads, const double* __restrict__ deltas, int n,
                                                               long long* __restrict__ out, int cap, int* __restrict__ counts,
                                                               double* __restrict__ trig) {
  __shared__ IPt s_poly[512];
  const int c = blockIdx.x;
  if (c >= n || threadIdx.x != 0) return;
  if (trig) {
    double st = 0, sn = 0, cs = 0;
    if (!(deltas[c] > -1.0e-20 && deltas[c] < 1.0e-20)) clipper_round_setup(deltas[c], st, sn, cs);
    trig[3 * c] = st; trig[3 * c + 1] = sn; trig[3 * c + 2] = cs;
  }
  IPt q[4];
  for (int i = 0; i < 4; ++i) q[i] = {(long long)quads[c * 8 + 2 * i], (long long)quads[c * 8 + 2 * i + 1]};
  const int un = clipper_offset_round(q, deltas[c], s_poly, 512);
  counts[c] = un;
  for (int i = 0; i < un && i < cap; ++i) {
    out[((long)c * cap + i) * 2] = s_poly[i].X;
    out[((long)c * cap + i) * 2 + 1] = s_poly[i].Y;
  }
}
// out14 per case: UnClip's RotatedRect (cx, cy, w, h, angle), GetMiniBoxes' ssid, its four corners (x, y) x 4; status per case:
// POST_ERR_* bit or 0, vertices of the offset polygon
__global__ void __launch_bounds__(64) selftest_unclip_box_kernel(const float* __restrict__ boxes, float unclip_ratio, int n,
                                                                  float* __restrict__ out14, int* __restrict__ status) {
  constexpr int HULL_CAP = 512, UNCLIP_CAP = 256;
  __shared__ P2f s_hull[HULL_CAP];
  __shared__ __attribute__((aligned(16))) unsigned char s_a[HULL_CAP * (sizeof(P2f) + sizeof(float))];
  __shared__ __attribute__((aligned(16))) unsigned char s_b[UNCLIP_CAP * (sizeof(IPt) + sizeof(unsigned long long))];
  __shared__ int s_ustack[UNCLIP_CAP + 2];
  __shared__ int s_uhull[2 * UNCLIP_CAP];
  const int c = blockIdx.x;
  if (c >= n || threadIdx.x != 0) return;
  P2f arr[4];
  for (int i = 0; i < 4; ++i) { arr[i].x = boxes[c * 8 + 2 * i]; arr[i].y = boxes[c * 8 + 2 * i + 1]; }
  // (border_box_kernel overlays the polygon and the calipers' scratch in one region: each is dead when the other is written;
  // here they are apart, which changes no value)
  const UnclipScratch us{(IPt*)s_b, (unsigned long long*)(s_b + UNCLIP_CAP * sizeof(IPt)), s_ustack, s_uhull, s_hull, (P2f*)s_a,
                         (float*)(s_a + HULL_CAP * sizeof(P2f))};
  RRect pts{0, 0, 0, 0, 0};
  int npoly = 0;
  const int e = unclip_min_rect<HULL_CAP, UNCLIP_CAP>(arr, unclip_ratio, us, pts, &npoly);
  status[2 * c] = e;
  status[2 * c + 1] = npoly;
  float* o = out14 + (long)c * 14;
  o[0] = pts.cx; o[1] = pts.cy; o[2] = pts.w; o[3] = pts.h; o[4] = pts.angle;
  float ssid = 0.f;
  P2f clip[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  if (!e) get_mini_boxes(pts, ssid, clip);
  o[5] = ssid;
  for (int i = 0; i < 4; ++i) { o[6 + 2 * i] = clip[i].x; o[7 + 2 * i] = clip[i].y; }
}
void launch_selftest_clipper(const int* quads, const double* deltas, int n, long long* out, int cap, int* counts, double* trig, hipStream_t s) {
  hipLaunchKernelGGL(selftest_clipper_kernel, dim3(n), dim3(64), 0, s, quads, deltas, n, out, cap, counts, trig);
}
void launch_selftest_unclip_box(const float* boxes, float unclip_ratio, int n, float* out14, int* status, hipStream_t s) {
  hipLaunchKernelGGL(selftest_unclip_box_kernel, dim3(n), dim3(64), 0, s, boxes, unclip_ratio, n, out14, status);
}

}  // namespace ocr
