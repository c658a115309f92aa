// Host-visible launch interface of kernels_post.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocr {

enum : int { POST_ERR_POOL = 1, POST_ERR_HULL = 2, POST_ERR_UNCLIP = 4 };

// Geometry of one image of a batch of MIXED sizes (the detector's ragged path): its map size, where its maps start in
// the batch's buffers (bitmap / labels share one layout, the probability maps may sit elsewhere), and the numbers
// FilterTagDetRes needs (postprocess_op.cpp:333-362).
struct PostImg {
  int h, w;
  int off;    // first pixel in bitmap / labels / touch
  int poff;   // first float in pred
  int src_h, src_w;
  float ratio_h, ratio_w;
};

struct PostArgs {
  const PostImg* img = nullptr;  // [N] device; null: N maps of H x W one after the other, one ratio / source size
  const uint8_t* bitmap;  // [N][H][W] {0,1}
  const float* pred;      // [N][H][W]
  int* labels;            // [N*H*W]
  uint8_t* touch;         // [N*H*W]
  int* chunk_cnt;         // [N][128] scratch of the ordered compaction
  int* ncont_all;         // [N] borders discovered
  int* ncont;             // [N] min(discovered, max_cand)
  int* starts;            // [N][max_cand] start pixel (image-local), reference order
  int* npts;              // [N][max_cand] CHAIN_APPROX_SIMPLE vertex count
  int* poff;              // [N][max_cand] offset into the image's key pool (-1: dropped)
  unsigned long long* pool;  // [N][pool_cap] keys
  int* iscratch;          // [N][pool_cap*4] hull stacks
  int* cand_boxes;        // [N][max_cand][8]
  int* cand_valid;        // [N][max_cand]
  int* status;            // device error bits (POST_ERR_*)
  int* pool_need;         // [N] key slots the image's first max_cand borders need (whether they fitted pool_cap or not)
  unsigned* mask_pool;    // [N][mask_pool_words] scratch of the slow (polygon) score; null in fast mode
  unsigned* mask_pool_top;  // [N] bump cursors (zeroed per call)
  unsigned mask_pool_words;
  int slow;               // det_db_score_mode == "slow"
  int fill_shifted;       // cv::fillPoly rule of the box score: 1 = OpenCV >= 4.5.2 (OCR_CV_410), 0 = 4.5.1 (OCR_CV_45)
  int pool_cap;
  int H, W, max_cand;
  float box_thresh, unclip_ratio, ratio_h, ratio_w;
  int src_h, src_w;
  int probe_stop;         // development probe (OCR_POST_STOP): border_box_kernel returns after stage n (timing only)
};

void launch_bitmap(const float* prob, uint8_t* bm, long total, int ithresh, hipStream_t s);
void launch_dilate2(const uint8_t* src, uint8_t* dst, int N, int H, int W, hipStream_t s);
// out_boxes [N][cap][8], out_n [N]
void launch_post(const PostArgs& a, int N, int* out_boxes, int cap, int* out_n, hipStream_t s);
void launch_post_large(const PostArgs& a, int N, int* out_boxes, int cap, int* out_n, hipStream_t s);

// self-test taps (ocr_selftest_unclip / ocr_selftest_unclip_box): clipper_offset_round and unclip_min_rect + get_mini_boxes as
// border_box_kernel runs them, one case per workgroup.  All pointers device memory.
void launch_selftest_clipper(const int* quads, const double* deltas, int n, long long* out, int cap, int* counts, double* trig /* [n][3] steps, m_sin, m_cos; may be null */, hipStream_t s);
void launch_selftest_unclip_box(const float* boxes, float unclip_ratio, int n, float* out14, int* status, hipStream_t s);

}  // namespace ocr
