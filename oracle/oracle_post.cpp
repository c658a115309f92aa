// TEST INFRASTRUCTURE — CPU oracle, not part of the product path.
//
// Restatement of the reference's DB post-processing and of the small host-side steps around it:
//   DBDetector::Run (map -> bitmap)        /root/reference/src/ocr_det.cpp:137-159
//   DBPostProcessor::BoxesFromBitmap       /root/reference/src/postprocess_op.cpp:255-331
//   GetMiniBoxes / BoxScoreFast / UnClip / GetContourArea   postprocess_op.cpp:134-168, 216-253, 39-72, 20-37
//   OrderPointsClockwise / FilterTagDetRes postprocess_op.cpp:87-104, 333-362
//   ClipperOffset (round join)             /root/reference/src/clipper.cpp:3628-3733, 3779-3982, 4006-4021
//   crop rectangle                         /root/reference/src/ocr_worker.cpp:245-258
//   greedy CTC decode                      /root/reference/src/ocr_rec.cpp:97-128
// The OpenCV calls on this path (threshold, dilate, findContours, minAreaRect, boxPoints, fillPoly,
// mean, boundingRect) are third-party code absent from /root/reference (OpenCV 4.x via vcpkg,
// version unpinned): their published algorithms are restated below (legacy Suzuki-Abe border
// following with CHAIN_APPROX_SIMPLE, Sklansky hull, float rotating calipers, XY_SHIFT=16 edge
// fill + Bresenham outline) — PARITY UNPINNED for those, SURVEY.md Appendix B.
// The ClipperOffset restatement (DoOffset + the union clean-up reduced to what it does on a
// convex round-join polygon) IS pinned: tests/test_oracle_unclip.py checks it point for point
// against tests/golden/unclip_ref.json, generated from the reference's own compiled clipper.cpp
// (oracle/_ref/libclipper_ref.so), and live against that library when present.
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Pt { int x, y; };
struct Pt2f { float x, y; };
struct RRect { float cx = 0, cy = 0, w = 0, h = 0, angle = 0; };

// ------------------------------------------------------------------ cv::findContours (RETR_LIST, CHAIN_APPROX_SIMPLE)
// Legacy implementation semantics: the image is padded by one zero pixel, binarised to {0,1},
// scanned in raster order; every new border is traced with icvFetchContour; contours come back
// in reverse discovery order.
void find_contours(const uint8_t* bitmap, int H, int W, std::vector<std::vector<Pt>>& out) {
  const int step = W + 2, rows = H + 2;
  std::vector<signed char> img((size_t)step * rows, 0);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) img[(size_t)(y + 1) * step + x + 1] = bitmap[(size_t)y * W + x] ? 1 : 0;
  const int deltas8[8] = {1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1};
  int deltas[16];
  for (int i = 0; i < 16; ++i) deltas[i] = deltas8[i & 7];
  static const int cdx[8] = {1, 1, 0, -1, -1, -1, 0, 1}, cdy[8] = {0, -1, -1, -1, 0, 1, 1, 1};
  std::vector<std::vector<Pt>> found;
  for (int y = 1; y < rows - 1; ++y) {
    signed char* row = img.data() + (size_t)y * step;
    int prev = 0;
    for (int x = 1; x < step - 1; ++x) {
      const int p = row[x];
      if (p == prev) continue;
      bool is_hole = false;
      bool start = false;
      if (prev == 0 && p == 1) start = true;
      else if (p == 0 && prev >= 1) { start = true; is_hole = true; }
      if (start) {
        // icvFetchContour(img + x - is_hole, step, origin, method = CHAIN_APPROX_SIMPLE)
        std::vector<Pt> c;
        signed char* i0 = row + x - (is_hole ? 1 : 0);
        Pt pt{x - (is_hole ? 1 : 0) - 1, y - 1};  // offset (-1,-1) undoes the padding
        const signed char nbd = 2;
        int s_end, s;
        s_end = s = is_hole ? 0 : 4;
        signed char* i1;
        do {
          s = (s - 1) & 7;
          i1 = i0 + deltas[s];
        } while (*i1 == 0 && s != s_end);
        if (s == s_end) {
          *i0 = (signed char)(nbd | -128);
          c.push_back(pt);
        } else {
          signed char* i3 = i0;
          signed char* i4 = nullptr;
          int prev_s = s ^ 4;
          for (;;) {
            s_end = s;
            s = std::min(s, 15);
            while (s < 15) {
              i4 = i3 + deltas[++s];
              if (*i4 != 0) break;
            }
            s &= 7;
            if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
            else if (*i3 == 1) *i3 = nbd;
            if (s != prev_s) { c.push_back(pt); prev_s = s; }
            pt.x += cdx[s];
            pt.y += cdy[s];
            if (i4 == i0 && i3 == i1) break;
            i3 = i4;
            s = (s + 4) & 7;
          }
        }
        found.push_back(std::move(c));
        // the scan resumes right after the transition with the (possibly re-marked) pixel value
        prev = row[x];
        continue;
      }
      prev = p;
    }
  }
  out.assign(found.rbegin(), found.rend());
}

// ------------------------------------------------------------------ cv::convexHull (returnPoints, clockwise=false)
template <class T> inline int sgn(T v) { return (v > 0) - (v < 0); }

template <class P, class DotT>
int sklansky(const std::vector<const P*>& a, int start, int end, int* stack, int nsign, int sign2) {
  int incr = end > start ? 1 : -1;
  int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
  int stacksize = 3;
  if (start == end || (a[start]->x == a[end]->x && a[start]->y == a[end]->y)) {
    stack[0] = start;
    return 1;
  }
  stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
  end += incr;
  while (pnext != end) {
    auto cury = a[pcur]->y, nexty = a[pnext]->y;
    auto by = nexty - cury;
    if (sgn(by) != nsign) {
      auto ax = a[pcur]->x - a[pprev]->x;
      auto bx = a[pnext]->x - a[pcur]->x;
      auto ay = cury - a[pprev]->y;
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

template <class P, class DotT>
void convex_hull(const std::vector<P>& pts, std::vector<P>& hull) {
  const int total = (int)pts.size();
  hull.clear();
  if (total == 0) return;
  const bool clockwise = false;
  std::vector<const P*> pointer(total);
  for (int i = 0; i < total; ++i) pointer[i] = &pts[i];
  std::sort(pointer.begin(), pointer.end(), [](const P* a, const P* b) {
    if (a->x != b->x) return a->x < b->x;
    if (a->y != b->y) return a->y < b->y;
    return a < b;
  });
  int miny_ind = 0, maxy_ind = 0;
  for (int i = 1; i < total; ++i) {
    auto y = pointer[i]->y;
    if (pointer[miny_ind]->y > y) miny_ind = i;
    if (pointer[maxy_ind]->y < y) maxy_ind = i;
  }
  std::vector<int> stackv(total + 2), hullbuf(total);
  int* stack = stackv.data();
  int nout = 0;
  const P* data0 = pts.data();
  if (pointer[0]->x == pointer[total - 1]->x && pointer[0]->y == pointer[total - 1]->y) {
    hullbuf[nout++] = 0;
  } else {
    int* tl_stack = stack;
    int tl_count = sklansky<P, DotT>(pointer, 0, maxy_ind, tl_stack, -1, 1);
    int* tr_stack = stack + tl_count;
    int tr_count = sklansky<P, DotT>(pointer, total - 1, maxy_ind, tr_stack, -1, -1);
    if (!clockwise) { std::swap(tl_stack, tr_stack); std::swap(tl_count, tr_count); }
    for (int i = 0; i < tl_count - 1; ++i) hullbuf[nout++] = int(pointer[tl_stack[i]] - data0);
    for (int i = tr_count - 1; i > 0; --i) hullbuf[nout++] = int(pointer[tr_stack[i]] - data0);
    int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
    int* bl_stack = stack;
    int bl_count = sklansky<P, DotT>(pointer, 0, miny_ind, bl_stack, 1, -1);
    int* br_stack = stack + bl_count;
    int br_count = sklansky<P, DotT>(pointer, total - 1, miny_ind, br_stack, 1, 1);
    if (clockwise) { std::swap(bl_stack, br_stack); std::swap(bl_count, br_count); }
    if (stop_idx >= 0) {
      int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
      if (check_idx == stop_idx || (check_idx >= 0 && pointer[check_idx]->x == pointer[stop_idx]->x &&
                                    pointer[check_idx]->y == pointer[stop_idx]->y)) {
        bl_count = std::min(bl_count, 2);
        br_count = std::min(br_count, 2);
      }
    }
    for (int i = 0; i < bl_count - 1; ++i) hullbuf[nout++] = int(pointer[bl_stack[i]] - data0);
    for (int i = br_count - 1; i > 0; --i) hullbuf[nout++] = int(pointer[br_stack[i]] - data0);
    if (nout >= 3) {
      int min_idx = 0, max_idx = 0, lt = 0, i;
      for (i = 1; i < nout; ++i) {
        int idx = hullbuf[i];
        lt += hullbuf[i - 1] < idx;
        if (lt > 1 && lt <= i - 2) break;
        if (idx < hullbuf[min_idx]) min_idx = i;
        if (idx > hullbuf[max_idx]) max_idx = i;
      }
      int mmdist = std::abs(max_idx - min_idx);
      if ((mmdist == 1 || mmdist == nout - 1) && (lt <= 1 || lt >= nout - 2)) {
        int ascending = (max_idx + 1) % nout == min_idx;
        int i0 = ascending ? min_idx : max_idx, j = i0;
        if (i0 > 0) {
          for (i = 0; i < nout; ++i) {
            int curr_idx = stack[i] = hullbuf[j];
            int next_j = j + 1 < nout ? j + 1 : 0;
            int next_idx = hullbuf[next_j];
            if (i < nout - 1 && (ascending != (curr_idx < next_idx))) break;
            j = next_j;
          }
          if (i == nout) memcpy(hullbuf.data(), stack, nout * sizeof(int));
        }
      }
    }
  }
  for (int i = 0; i < nout; ++i) hull.push_back(pts[hullbuf[i]]);
}

// ------------------------------------------------------------------ rotatingCalipers (CALIPERS_MINAREARECT), float
void rotating_calipers(const std::vector<Pt2f>& points, float out[6]) {
  const int n = (int)points.size();
  float minarea = FLT_MAX;
  std::vector<float> inv_vect_length(n);
  std::vector<Pt2f> vect(n);
  int left = 0, bottom = 0, right = 0, top = 0;
  int seq[4] = {-1, -1, -1, -1};
  float orientation = 0, base_a, base_b = 0;
  float left_x, right_x, top_y, bottom_y;
  Pt2f pt0 = points[0];
  left_x = right_x = pt0.x;
  top_y = bottom_y = pt0.y;
  for (int i = 0; i < n; ++i) {
    if (pt0.x < left_x) left_x = pt0.x, left = i;
    if (pt0.x > right_x) right_x = pt0.x, right = i;
    if (pt0.y > top_y) top_y = pt0.y, top = i;
    if (pt0.y < bottom_y) bottom_y = pt0.y, bottom = i;
    Pt2f pt = points[(i + 1) < n ? i + 1 : 0];
    double dx = pt.x - pt0.x, dy = pt.y - pt0.y;
    vect[i].x = (float)dx;
    vect[i].y = (float)dy;
    inv_vect_length[i] = (float)(1. / std::sqrt(dx * dx + dy * dy));
    pt0 = pt;
  }
  {
    double ax = vect[n - 1].x, ay = vect[n - 1].y;
    for (int i = 0; i < n; ++i) {
      double bx = vect[i].x, by = vect[i].y;
      double convexity = ax * by - ay * bx;
      if (convexity != 0) { orientation = (convexity > 0) ? 1.f : (-1.f); break; }
      ax = bx; ay = by;
    }
  }
  base_a = orientation;
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
  int best_left = 0, best_bottom = 0;
  float b_a = 0, b_w = 0, b_b = 0, b_h = 0;
  for (int k = 0; k < n; ++k) {
    float dp[4] = {
        +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y,
        -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y,
        -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y,
        +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y,
    };
    float maxcos = dp[0] * inv_vect_length[seq[0]];
    int main_element = 0;
    for (int i = 1; i < 4; ++i) {
      float cosalpha = dp[i] * inv_vect_length[seq[i]];
      if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; }
    }
    {
      int pindex = seq[main_element];
      float lead_x = vect[pindex].x * inv_vect_length[pindex];
      float lead_y = vect[pindex].y * inv_vect_length[pindex];
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
      float width = dx * base_a + dy * base_b;
      dx = points[seq[2]].x - points[seq[0]].x;
      dy = points[seq[2]].y - points[seq[0]].y;
      float height = -dx * base_b + dy * base_a;
      float area = width * height;
      if (area <= minarea) {
        minarea = area;
        best_left = seq[3];
        b_a = base_a; b_w = width; b_b = base_b; b_h = height;
        best_bottom = seq[0];
      }
    }
  }
  float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
  float C1 = A1 * points[best_left].x + points[best_left].y * B1;
  float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
  float idet = 1.f / (A1 * B2 - A2 * B1);
  float px = (C1 * B2 - C2 * B1) * idet;
  float py = (A1 * C2 - A2 * C1) * idet;
  out[0] = px; out[1] = py;
  out[2] = A1 * b_w; out[3] = B1 * b_w;
  out[4] = A2 * b_h; out[5] = B2 * b_h;
}

RRect min_area_rect_hullf(const std::vector<Pt2f>& h) {
  RRect box;
  const int n = (int)h.size();
  if (n > 2) {
    float out[6];
    rotating_calipers(h, out);
    box.cx = out[0] + (out[2] + out[4]) * 0.5f;
    box.cy = out[1] + (out[3] + out[5]) * 0.5f;
    box.w = (float)std::sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
    box.h = (float)std::sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
    box.angle = (float)atan2((double)out[3], (double)out[2]);
  } else if (n == 2) {
    box.cx = (h[0].x + h[1].x) * 0.5f;
    box.cy = (h[0].y + h[1].y) * 0.5f;
    double dx = h[1].x - h[0].x, dy = h[1].y - h[0].y;
    box.w = (float)std::sqrt(dx * dx + dy * dy);
    box.h = 0;
    box.angle = (float)atan2(dy, dx);
  } else if (n == 1) {
    box.cx = h[0].x;
    box.cy = h[0].y;
  }
  box.angle = (float)(box.angle * 180 / 3.14159265358979323846);
  return box;
}

RRect min_area_rect_i(const std::vector<Pt>& pts) {
  std::vector<Pt> hull;
  convex_hull<Pt, int>(pts, hull);
  std::vector<Pt2f> hf(hull.size());
  for (size_t i = 0; i < hull.size(); ++i) hf[i] = {(float)hull[i].x, (float)hull[i].y};
  return min_area_rect_hullf(hf);
}
RRect min_area_rect_f(const std::vector<Pt2f>& pts) {
  std::vector<Pt2f> hull;
  convex_hull<Pt2f, float>(pts, hull);
  return min_area_rect_hullf(hull);
}

// RotatedRect::points
void box_points(const RRect& r, Pt2f pt[4]) {
  double _angle = r.angle * 3.14159265358979323846 / 180.;
  float b = (float)cos(_angle) * 0.5f;
  float a = (float)sin(_angle) * 0.5f;
  pt[0].x = r.cx - a * r.h - b * r.w;
  pt[0].y = r.cy + b * r.h - a * r.w;
  pt[1].x = r.cx + a * r.h - b * r.w;
  pt[1].y = r.cy - b * r.h - a * r.w;
  pt[2].x = 2 * r.cx - pt[0].x;
  pt[2].y = 2 * r.cy - pt[0].y;
  pt[3].x = 2 * r.cx - pt[1].x;
  pt[3].y = 2 * r.cy - pt[1].y;
}

// GetMiniBoxes (postprocess_op.cpp:134-168).  std::sort with the non-strict-weak XsortFp32: for 4
// elements libstdc++ runs insertion sort; restated as a stable insertion sort on x.
void get_mini_boxes(const RRect& box, float& ssid, Pt2f out[4]) {
  ssid = std::max(box.w, box.h);
  Pt2f a[4];
  box_points(box, a);
  for (int i = 1; i < 4; ++i) {
    Pt2f v = a[i];
    int j = i - 1;
    if (v.x < a[0].x) {  // libstdc++ __insertion_sort: smaller than first -> rotate to front
      for (int k = i; k > 0; --k) a[k] = a[k - 1];
      a[0] = v;
    } else {
      while (v.x < a[j].x) { a[j + 1] = a[j]; --j; }
      a[j + 1] = v;
    }
  }
  Pt2f idx1, idx2, idx3, idx4;
  if (a[3].y <= a[2].y) { idx2 = a[3]; idx3 = a[2]; } else { idx2 = a[2]; idx3 = a[3]; }
  if (a[1].y <= a[0].y) { idx1 = a[1]; idx4 = a[0]; } else { idx1 = a[0]; idx4 = a[1]; }
  out[0] = idx1; out[1] = idx2; out[2] = idx3; out[3] = idx4;
}

// ------------------------------------------------------------------ cv::fillPoly (one polygon, color 1) into a mask
struct PolyEdge { int y0, y1; int64_t x, dx; PolyEdge* next; };

bool clip_line(int64_t w, int64_t h, int64_t& x1, int64_t& y1, int64_t& x2, int64_t& y2) {
  int64_t right = w - 1, bottom = h - 1;
  if (w <= 0 || h <= 0) return false;
  int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
  int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
  if ((c1 & c2) == 0 && (c1 | c2) != 0) {
    int64_t a;
    if (c1 & 12) { a = c1 < 8 ? 0 : bottom; x1 += (int64_t)((double)(a - y1) * (x2 - x1) / (y2 - y1)); y1 = a; c1 = (x1 < 0) + (x1 > right) * 2; }
    if (c2 & 12) { a = c2 < 8 ? 0 : bottom; x2 += (int64_t)((double)(a - y2) * (x2 - x1) / (y2 - y1)); y2 = a; c2 = (x2 < 0) + (x2 > right) * 2; }
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
      if (c1) { a = c1 == 1 ? 0 : right; y1 += (int64_t)((double)(a - x1) * (y2 - y1) / (x2 - x1)); x1 = a; c1 = 0; }
      if (c2) { a = c2 == 1 ? 0 : right; y2 += (int64_t)((double)(a - x2) * (y2 - y1) / (x2 - x1)); x2 = a; c2 = 0; }
    }
  }
  return (c1 | c2) == 0;
}

// Line(img, pt1, pt2, color, 8): LineIterator(leftToRight = true)
void draw_line(uint8_t* mask, int W, int H, Pt p1, Pt p2) {
  int64_t x1 = p1.x, y1 = p1.y, x2 = p2.x, y2 = p2.y;
  if ((uint64_t)x1 >= (uint64_t)W || (uint64_t)x2 >= (uint64_t)W || (uint64_t)y1 >= (uint64_t)H || (uint64_t)y2 >= (uint64_t)H) {
    if (!clip_line(W, H, x1, y1, x2, y2)) return;
  }
  int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
  int delta_x = 1, delta_y = 1;
  int px = (int)x1, py = (int)y1;
  if (dx < 0) { dx = -dx; dy = -dy; px = (int)x2; py = (int)y2; }
  if (dy < 0) { dy = -dy; delta_y = -1; }
  bool vert = dy > dx;
  if (vert) { std::swap(dx, dy); std::swap(delta_x, delta_y); }
  int err = dx - (dy + dy);
  int plusDelta = dx + dx, minusDelta = -(dy + dy);
  int count = dx + 1;
  // minus* applies every step (major axis), plus* only when err < 0 (minor axis)
  int majx = vert ? 0 : delta_x, majy = vert ? delta_x : 0;
  int minx = vert ? delta_y : 0, miny = vert ? 0 : delta_y;
  for (int i = 0; i < count; ++i) {
    if (px >= 0 && px < W && py >= 0 && py < H) mask[(size_t)py * W + px] = 1;
    int m = err < 0 ? -1 : 0;
    err += minusDelta + (plusDelta & m);
    px += majx + (minx & m);
    py += majy + (miny & m);
  }
}

// compat: which OpenCV the reference was built against (PARITY UNPINNED: no OpenCV here, restated from the published
// source).  CV_FILL_CLASSIC = 4.0 .. 4.5.1: edges in 16.16 at the vertices' integer x, spans [ceil(xa), floor(xb)].
// CV_FILL_SHIFTED = 4.5.2 and later (to 4.12): for non-antialiased lines CollectPolyEdges moves an edge whose outline
// segment lies inside the image by half a pixel (`pt.x += XY_ONE >> 1`) - or, when the segment had to be clipped, builds
// the edge from the CLIPPED integer end points and extrapolates it back to the edge's first row - and FillEdgeCollection
// fills [xa >> 16, xb >> 16] (`delta = 0` for line_type < LINE_AA): both span ends are the crossing rounded half up.
enum { CV_FILL_CLASSIC = 0, CV_FILL_SHIFTED = 1 };

void fill_poly(uint8_t* mask, int W, int H, const std::vector<Pt>& v, int compat) {
  const int count = (int)v.size();
  if (count == 0) return;
  const int XY_SHIFT = 16;
  const int64_t XY_ONE = 1 << XY_SHIFT;
  std::vector<PolyEdge> edges;
  edges.reserve(count + 1);
  {
    int64_t p0x = (int64_t)v[count - 1].x << XY_SHIFT, p0y = v[count - 1].y;
    for (int i = 0; i < count; ++i) {
      int64_t p1x = (int64_t)v[i].x << XY_SHIFT, p1y = v[i].y;
      Pt t0{(int)((p0x + (XY_ONE >> 1)) >> XY_SHIFT), (int)p0y}, t1{(int)((p1x + (XY_ONE >> 1)) >> XY_SHIFT), (int)p1y};
      draw_line(mask, W, H, t0, t1);
      int64_t c0x = p0x, c0y = p0y, c1x = p1x, c1y = p1y;  // pt0c, pt1c
      if (compat == CV_FILL_SHIFTED) {
        if ((unsigned)t0.x >= (unsigned)W || (unsigned)t1.x >= (unsigned)W || (unsigned)t0.y >= (unsigned)H || (unsigned)t1.y >= (unsigned)H) {
          int64_t ax = t0.x, ay = t0.y, bx = t1.x, by = t1.y;
          (void)clip_line(W, H, ax, ay, bx, by);  // moves the points as far as it gets, whatever it returns
          if (ay != by) { c0y = ay; c1y = by; c0x = ax << XY_SHIFT; c1x = bx << XY_SHIFT; }
        } else {
          c0x += XY_ONE >> 1;
          c1x += XY_ONE >> 1;
        }
      }
      if (p0y != p1y) {
        PolyEdge e;
        e.dx = (c1x - c0x) / (c1y - c0y);
        if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = c0x + (p0y - c0y) * e.dx; }
        else { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = c1x + (p1y - c1y) * e.dx; }
        e.next = nullptr;
        edges.push_back(e);
      }
      p0x = p1x; p0y = p1y;
    }
  }
  const int64_t span_delta = compat == CV_FILL_SHIFTED ? 0 : XY_ONE - 1;
  int total = (int)edges.size();
  if (total < 2) return;
  int y_max = INT_MIN, y_min = INT_MAX;
  int64_t x_max = -1, x_min = 0x7FFFFFFFFFFFFFFFLL;
  for (int i = 0; i < total; ++i) {
    PolyEdge& e1 = edges[i];
    int64_t x1 = e1.x + (int64_t)(e1.y1 - e1.y0) * e1.dx;
    y_min = std::min(y_min, e1.y0); y_max = std::max(y_max, e1.y1);
    x_min = std::min(x_min, e1.x); x_max = std::max(x_max, e1.x);
    x_min = std::min(x_min, x1); x_max = std::max(x_max, x1);
  }
  if (y_max < 0 || y_min >= H || x_max < 0 || x_min >= ((int64_t)W << XY_SHIFT)) return;
  std::sort(edges.begin(), edges.end(), [](const PolyEdge& a, const PolyEdge& b) {
    return a.y0 - b.y0 ? a.y0 < b.y0 : a.x - b.x ? a.x < b.x : a.dx < b.dx;
  });
  PolyEdge tmp;
  tmp.y0 = INT_MAX;
  tmp.next = nullptr;
  edges.push_back(tmp);
  int i = 0;
  PolyEdge* e = &edges[i];
  PolyEdge head;
  head.next = nullptr;
  y_max = std::min(y_max, H);
  for (int y = e->y0; y < y_max; ++y) {
    PolyEdge *last, *prelast, *keep_prelast;
    int draw = 0;
    int clipline = y < 0;
    prelast = &head;
    last = head.next;
    while (last || e->y0 == y) {
      if (last && last->y1 == y) {
        prelast->next = last->next;
        last = last->next;
        continue;
      }
      keep_prelast = prelast;
      if (last && (e->y0 > y || last->x < e->x)) {
        prelast = last;
        last = last->next;
      } else if (i < total) {
        prelast->next = e;
        e->next = last;
        prelast = e;
        e = &edges[++i];
      } else
        break;
      if (draw) {
        if (!clipline) {
          int x1, x2;
          if (keep_prelast->x > prelast->x) {
            x1 = (int)((prelast->x + span_delta) >> XY_SHIFT);
            x2 = (int)(keep_prelast->x >> XY_SHIFT);
          } else {
            x1 = (int)((keep_prelast->x + span_delta) >> XY_SHIFT);
            x2 = (int)(prelast->x >> XY_SHIFT);
          }
          if (x1 < W && x2 >= 0) {
            if (x1 < 0) x1 = 0;
            if (x2 >= W) x2 = W - 1;
            for (int x = x1; x <= x2; ++x) mask[(size_t)y * W + x] = 1;
          }
        }
        keep_prelast->x += keep_prelast->dx;
        prelast->x += prelast->dx;
      }
      draw ^= 1;
    }
    keep_prelast = nullptr;
    do {
      prelast = &head;
      last = head.next;
      PolyEdge* last_exchange = nullptr;
      while (last != keep_prelast && last && last->next != nullptr) {
        PolyEdge* te = last->next;
        if (last->x > te->x) {
          prelast->next = te;
          last->next = te->next;
          te->next = last;
          prelast = te;
          last_exchange = prelast;
        } else {
          prelast = last;
          last = te;
        }
      }
      if (last_exchange == nullptr) break;
      keep_prelast = last_exchange;
    } while (keep_prelast != head.next && keep_prelast != &head);
  }
}

inline int clampi(int x, int lo, int hi) { return x > hi ? hi : (x < lo ? lo : x); }

// masked mean of pred over fillPoly(pts - (xmin,ymin)) inside [xmin..xmax]x[ymin..ymax]
float masked_mean(const float* pred, int W, int xmin, int ymin, int xmax, int ymax, const std::vector<Pt>& local, int compat) {
  const int mw = xmax - xmin + 1, mh = ymax - ymin + 1;
  std::vector<uint8_t> mask((size_t)mw * mh, 0);
  fill_poly(mask.data(), mw, mh, local, compat);
  double s = 0;
  long cnt = 0;
  for (int y = 0; y < mh; ++y)
    for (int x = 0; x < mw; ++x)
      if (mask[(size_t)y * mw + x]) { s += pred[(size_t)(y + ymin) * W + x + xmin]; ++cnt; }
  return cnt ? (float)(s / (double)cnt) : 0.f;
}

float box_score_fast(const Pt2f a[4], const float* pred, int H, int W, int compat) {
  float bx[4] = {a[0].x, a[1].x, a[2].x, a[3].x}, by[4] = {a[0].y, a[1].y, a[2].y, a[3].y};
  int xmin = clampi(int(std::floor(*std::min_element(bx, bx + 4))), 0, W - 1);
  int xmax = clampi(int(std::ceil(*std::max_element(bx, bx + 4))), 0, W - 1);
  int ymin = clampi(int(std::floor(*std::min_element(by, by + 4))), 0, H - 1);
  int ymax = clampi(int(std::ceil(*std::max_element(by, by + 4))), 0, H - 1);
  std::vector<Pt> local(4);
  for (int i = 0; i < 4; ++i) local[i] = {int(a[i].x) - xmin, int(a[i].y) - ymin};
  return masked_mean(pred, W, xmin, ymin, xmax, ymax, local, compat);
}

float polygon_score_acc(const std::vector<Pt>& contour, const float* pred, int H, int W, int compat) {
  float fxmin = FLT_MAX, fxmax = -FLT_MAX, fymin = FLT_MAX, fymax = -FLT_MAX;
  for (auto& p : contour) {
    fxmin = std::min(fxmin, (float)p.x); fxmax = std::max(fxmax, (float)p.x);
    fymin = std::min(fymin, (float)p.y); fymax = std::max(fymax, (float)p.y);
  }
  int xmin = clampi(int(std::floor(fxmin)), 0, W - 1), xmax = clampi(int(std::ceil(fxmax)), 0, W - 1);
  int ymin = clampi(int(std::floor(fymin)), 0, H - 1), ymax = clampi(int(std::ceil(fymax)), 0, H - 1);
  std::vector<Pt> local(contour.size());
  for (size_t i = 0; i < contour.size(); ++i) local[i] = {contour[i].x - xmin, contour[i].y - ymin};
  return masked_mean(pred, W, xmin, ymin, xmax, ymax, local, compat);
}

// ------------------------------------------------------------------ ClipperOffset (jtRound, etClosedPolygon)
struct IPt { long long X, Y; bool operator==(const IPt& o) const { return X == o.X && Y == o.Y; } bool operator!=(const IPt& o) const { return !(*this == o); } };
struct DPt { double X, Y; };
inline long long clip_round(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }
inline DPt unit_normal(const IPt& a, const IPt& b) {
  if (a == b) return {0, 0};
  double Dx = (double)(b.X - a.X), dy = (double)(b.Y - a.Y);
  double f = 1 * 1.0 / std::sqrt(Dx * Dx + dy * dy);
  Dx *= f; dy *= f;
  return {dy, -Dx};
}
double path_area(const std::vector<IPt>& p) {
  int size = (int)p.size();
  if (size < 3) return 0;
  double a = 0;
  for (int i = 0, j = size - 1; i < size; ++i) { a += ((double)p[j].X + p[i].X) * ((double)p[j].Y - p[i].Y); j = i; }
  return -a * 0.5;
}

// returns the single solution path of ClipperOffset::Execute (empty when Execute yields nothing)
std::vector<IPt> clipper_offset_round(const std::vector<IPt>& path, double delta) {
  std::vector<IPt> dest;
  int highI = (int)path.size() - 1;
  if (highI < 0) return dest;
  while (highI > 0 && path[0] == path[highI]) highI--;
  std::vector<IPt> c;
  c.push_back(path[0]);
  for (int i = 1; i <= highI; ++i) if (c.back() != path[i]) c.push_back(path[i]);
  if ((int)c.size() < 3) return dest;
  if (!(path_area(c) >= 0)) std::reverse(c.begin(), c.end());
  const double pi = 3.141592653589793238, two_pi = pi * 2;
  if (delta > -1.0e-20 && delta < 1.0e-20) {
    dest = c;
  } else {
    double y = 0.25;
    if (0.25 > std::fabs(delta) * 0.25) y = std::fabs(delta) * 0.25;
    double steps = pi / std::acos(1 - y / std::fabs(delta));
    if (steps > std::fabs(delta) * pi) steps = std::fabs(delta) * pi;
    double m_sin = std::sin(two_pi / steps), m_cos = std::cos(two_pi / steps);
    double steps_per_rad = steps / two_pi;
    if (delta < 0.0) m_sin = -m_sin;
    const int len = (int)c.size();
    std::vector<DPt> nrm(len);
    for (int j = 0; j < len - 1; ++j) nrm[j] = unit_normal(c[j], c[j + 1]);
    nrm[len - 1] = unit_normal(c[len - 1], c[0]);
    int k = len - 1;
    for (int j = 0; j < len; ++j) {
      double sinA = nrm[k].X * nrm[j].Y - nrm[j].X * nrm[k].Y;
      if (std::fabs(sinA * delta) < 1.0) {
        double cosA = nrm[k].X * nrm[j].X + nrm[j].Y * nrm[k].Y;
        if (cosA > 0) {
          dest.push_back({clip_round(c[j].X + nrm[k].X * delta), clip_round(c[j].Y + nrm[k].Y * delta)});
          continue;  // NB: OffsetPoint returns before `k = j`
        }
      } else if (sinA > 1.0) sinA = 1.0;
      else if (sinA < -1.0) sinA = -1.0;
      if (sinA * delta < 0) {
        dest.push_back({clip_round(c[j].X + nrm[k].X * delta), clip_round(c[j].Y + nrm[k].Y * delta)});
        dest.push_back(c[j]);
        dest.push_back({clip_round(c[j].X + nrm[j].X * delta), clip_round(c[j].Y + nrm[j].Y * delta)});
      } else {
        double a = std::atan2(sinA, nrm[k].X * nrm[j].X + nrm[k].Y * nrm[j].Y);
        int st = std::max((int)clip_round(steps_per_rad * std::fabs(a)), 1);
        double X = nrm[k].X, Y = nrm[k].Y, X2;
        for (int i = 0; i < st; ++i) {
          dest.push_back({clip_round(c[j].X + X * delta), clip_round(c[j].Y + Y * delta)});
          X2 = X;
          X = X * m_cos - m_sin * Y;
          Y = X2 * m_sin + Y * m_cos;
        }
        dest.push_back({clip_round(c[j].X + nrm[j].X * delta), clip_round(c[j].Y + nrm[j].Y * delta)});
      }
      k = j;
    }
  }
  // Union clean-up (Clipper::Execute(ctUnion, pftPositive) on one outward-offset convex polygon):
  // duplicate and collinear vertices go (FixupOutPolygon), the ring is emitted starting after the
  // last top-most vertex (BuildResult order).  Verified point-for-point against the reference's
  // compiled clipper.cpp (tests/golden/unclip_ref.json).
  std::vector<IPt> p = dest;
  bool changed = true;
  while (changed && p.size() >= 3) {
    changed = false;
    size_t i = 0;
    while (i < p.size() && p.size() >= 3) {
      const IPt& a = p[(i + p.size() - 1) % p.size()];
      const IPt& b = p[i];
      const IPt& cc = p[(i + 1) % p.size()];
      if (b == cc || b == a || (b.Y - a.Y) * (cc.X - b.X) == (b.X - a.X) * (cc.Y - b.Y)) {
        p.erase(p.begin() + i);
        changed = true;
        if (i > 0) --i;
      } else
        ++i;
    }
  }
  if (p.size() < 3) return {};
  // a tiny delta on a ~1 px thin quad can fold the ring over itself: a ring with non-positive winding
  // is not part of a pftPositive union
  if (!(path_area(p) > 0)) return {};
  long long miny = p[0].Y;
  for (auto& q : p) miny = std::min(miny, q.Y);
  size_t e = 0;
  for (size_t i = 0; i < p.size(); ++i)
    if (p[i].Y == miny && p[(i + 1) % p.size()].Y != miny) { e = i; break; }
  std::vector<IPt> outp;
  for (size_t i = 0; i < p.size(); ++i) outp.push_back(p[(e + 1 + i) % p.size()]);
  return outp;
}

// DBPostProcessor::UnClip
RRect unclip(const Pt2f box[4], float unclip_ratio) {
  float area = 0.0f, dist = 0.0f;
  for (int i = 0; i < 4; ++i) {
    area += box[i].x * box[(i + 1) % 4].y - box[i].y * box[(i + 1) % 4].x;
    dist += sqrtf((box[i].x - box[(i + 1) % 4].x) * (box[i].x - box[(i + 1) % 4].x) +
                  (box[i].y - box[(i + 1) % 4].y) * (box[i].y - box[(i + 1) % 4].y));
  }
  area = fabs(float(area / 2.0));
  float distance = area * unclip_ratio / dist;
  std::vector<IPt> p(4);
  for (int i = 0; i < 4; ++i) p[i] = {(long long)int(box[i].x), (long long)int(box[i].y)};
  std::vector<IPt> soln = clipper_offset_round(p, distance);
  RRect res;
  if (soln.empty()) { res.cx = 0; res.cy = 0; res.w = 1; res.h = 1; res.angle = 0; return res; }
  std::vector<Pt2f> points(soln.size());
  for (size_t i = 0; i < soln.size(); ++i) points[i] = {(float)soln[i].X, (float)soln[i].Y};
  return min_area_rect_f(points);
}

inline float clampf(float x, float lo, float hi) { return x > hi ? hi : (x < lo ? lo : x); }

struct Box { int p[4][2]; };

void boxes_from_bitmap(const float* pred, const uint8_t* bitmap, int H, int W, float box_thresh, float unclip_ratio,
                       bool slow, std::vector<Box>& boxes, int compat) {
  const int min_size = 3, max_candidates = 1000;
  std::vector<std::vector<Pt>> contours;
  find_contours(bitmap, H, W, contours);
  int num = (int)contours.size() >= max_candidates ? max_candidates : (int)contours.size();
  for (int ci = 0; ci < num; ++ci) {
    if (contours[ci].size() <= 2) continue;
    float ssid;
    RRect box = min_area_rect_i(contours[ci]);
    Pt2f arr[4];
    get_mini_boxes(box, ssid, arr);
    if (ssid < min_size) continue;
    float score = slow ? polygon_score_acc(contours[ci], pred, H, W, compat) : box_score_fast(arr, pred, H, W, compat);
    if (score < box_thresh) continue;
    RRect pts = unclip(arr, unclip_ratio);
    if (pts.h < 1.001 && pts.w < 1.001) continue;
    Pt2f clip[4];
    get_mini_boxes(pts, ssid, clip);
    if (ssid < min_size + 2) continue;
    Box b;
    for (int k = 0; k < 4; ++k) {
      b.p[k][0] = int(clampf(roundf(clip[k].x / float(W) * float(W)), 0, float(W)));
      b.p[k][1] = int(clampf(roundf(clip[k].y / float(H) * float(H)), 0, float(H)));
    }
    boxes.push_back(b);
  }
}

void filter_tag_det_res(std::vector<Box>& boxes, float ratio_h, float ratio_w, int src_h, int src_w) {
  std::vector<Box> keep;
  for (auto& b : boxes) {
    // OrderPointsClockwise: std::sort by x (XsortInt) on 4 elements == insertion sort
    int a[4][2];
    memcpy(a, b.p, sizeof(a));
    for (int i = 1; i < 4; ++i) {
      int vx = a[i][0], vy = a[i][1];
      if (vx < a[0][0]) {
        for (int k = i; k > 0; --k) { a[k][0] = a[k - 1][0]; a[k][1] = a[k - 1][1]; }
        a[0][0] = vx; a[0][1] = vy;
      } else {
        int j = i - 1;
        while (vx < a[j][0]) { a[j + 1][0] = a[j][0]; a[j + 1][1] = a[j][1]; --j; }
        a[j + 1][0] = vx; a[j + 1][1] = vy;
      }
    }
    int l0 = 0, l1 = 1, r0 = 2, r1 = 3;
    if (a[l0][1] > a[l1][1]) std::swap(l0, l1);
    if (a[r0][1] > a[r1][1]) std::swap(r0, r1);
    int ord[4] = {l0, r0, r1, l1};
    for (int m = 0; m < 4; ++m) {
      int x = a[ord[m]][0], y = a[ord[m]][1];
      x = (int)((float)x / ratio_w);  // `int /= float`
      y = (int)((float)y / ratio_h);
      x = std::min(std::max(x, 0), src_w - 1);
      y = std::min(std::max(y, 0), src_h - 1);
      b.p[m][0] = x; b.p[m][1] = y;
    }
  }
  for (auto& b : boxes) {
    int rect_width = int(sqrt(pow(b.p[0][0] - b.p[1][0], 2) + pow(b.p[0][1] - b.p[1][1], 2)));
    int rect_height = int(sqrt(pow(b.p[0][0] - b.p[3][0], 2) + pow(b.p[0][1] - b.p[3][1], 2)));
    if (rect_width <= 4 || rect_height <= 4) continue;
    keep.push_back(b);
  }
  boxes = keep;
}


// cv_compat as include/ocr_hip.h spells it: OCR_CV_45 (45) = the OpenCV 4.5.1 rules, OCR_CV_410 (410, and 0 = default) =
// the rules of 4.5.2 and later; the reference's README points at a 2025 vcpkg build (README.md:105-118)
int fill_rule(int cv_compat) { return cv_compat == 45 ? CV_FILL_CLASSIC : CV_FILL_SHIFTED; }

}  // namespace

extern "C" {

// pred [H][W] f32 -> bitmap [H][W] {0,1}:  cbuf=(uchar)(p*255); threshold(thr=det_db_thresh*255) ; optional 2x2 dilate
void oracle_bitmap(const float* pred, int H, int W, double det_db_thresh, int use_dilation, uint8_t* bitmap) {
  const double threshold = det_db_thresh * 255;
  const int ithresh = (int)std::floor(threshold);
  std::vector<uint8_t> bm((size_t)H * W);
  for (size_t i = 0; i < (size_t)H * W; ++i) {
    unsigned char cb = (unsigned char)(pred[i] * 255);
    bm[i] = cb > ithresh ? 1 : 0;
  }
  if (use_dilation) {
    // 2x2 rect, anchor (1,1): dst(y,x) = max src(y-1..y, x-1..x)
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        uint8_t m = 0;
        for (int dy = -1; dy <= 0; ++dy)
          for (int dx = -1; dx <= 0; ++dx) {
            int yy = y + dy, xx = x + dx;
            if (yy >= 0 && xx >= 0) m = std::max(m, bm[(size_t)yy * W + xx]);
          }
        bitmap[(size_t)y * W + x] = m;
      }
  } else {
    memcpy(bitmap, bm.data(), bm.size());
  }
}

// full det post-processing on a probability map. boxes: cap x 8 ints. returns count.
int oracle_det_post(const float* pred, int H, int W, double det_db_thresh, double box_thresh, double unclip_ratio,
                    int use_dilation, int slow, int src_h, int src_w, int* boxes, int cap, int cv_compat) {
  std::vector<uint8_t> bitmap((size_t)H * W);
  oracle_bitmap(pred, H, W, det_db_thresh, use_dilation, bitmap.data());
  std::vector<Box> bx;
  boxes_from_bitmap(pred, bitmap.data(), H, W, (float)box_thresh, (float)unclip_ratio, slow != 0, bx, fill_rule(cv_compat));
  float ratio_h = float(H) / float(src_h), ratio_w = float(W) / float(src_w);
  filter_tag_det_res(bx, ratio_h, ratio_w, src_h, src_w);
  int n = std::min((int)bx.size(), cap);
  for (int i = 0; i < n; ++i) memcpy(boxes + i * 8, bx[i].p, 8 * sizeof(int));
  return (int)bx.size();
}

// --- pieces, exposed for unit tests ---
int oracle_find_contours(const uint8_t* bitmap, int H, int W, int* pts_xy, int cap_pts, int* sizes, int cap_contours) {
  std::vector<std::vector<Pt>> c;
  find_contours(bitmap, H, W, c);
  int k = 0;
  for (size_t i = 0; i < c.size(); ++i) {
    if ((int)i < cap_contours) sizes[i] = (int)c[i].size();
    for (auto& p : c[i]) {
      if (k < cap_pts) { pts_xy[2 * k] = p.x; pts_xy[2 * k + 1] = p.y; }
      ++k;
    }
  }
  return (int)c.size();
}

void oracle_min_area_rect_i(const int* pts_xy, int n, float* out5) {
  std::vector<Pt> p(n);
  for (int i = 0; i < n; ++i) p[i] = {pts_xy[2 * i], pts_xy[2 * i + 1]};
  RRect r = min_area_rect_i(p);
  out5[0] = r.cx; out5[1] = r.cy; out5[2] = r.w; out5[3] = r.h; out5[4] = r.angle;
}

int oracle_clipper_offset(const long long* xy, int n, double delta, long long* out_xy, int cap) {
  std::vector<IPt> p(n);
  for (int i = 0; i < n; ++i) p[i] = {xy[2 * i], xy[2 * i + 1]};
  auto s = clipper_offset_round(p, delta);
  int k = std::min((int)s.size(), cap);
  for (int i = 0; i < k; ++i) { out_xy[2 * i] = s[i].X; out_xy[2 * i + 1] = s[i].Y; }
  return (int)s.size();
}

// DBPostProcessor::UnClip -> GetMiniBoxes of one box (postprocess_op.cpp:39-72, 134-168): out14 = RotatedRect (cx, cy, w, h,
// angle), ssid, the four corners; returns the vertex count of the offset polygon
int oracle_unclip_box(const float* box8, float unclip_ratio, float* out14) {
  Pt2f b[4];
  for (int i = 0; i < 4; ++i) b[i] = {box8[2 * i], box8[2 * i + 1]};
  // (the polygon's vertex count, for the tests' bookkeeping: the same call unclip() makes)
  std::vector<IPt> p(4);
  float area = 0.0f, dist = 0.0f;
  for (int i = 0; i < 4; ++i) {
    area += b[i].x * b[(i + 1) % 4].y - b[i].y * b[(i + 1) % 4].x;
    dist += sqrtf((b[i].x - b[(i + 1) % 4].x) * (b[i].x - b[(i + 1) % 4].x) + (b[i].y - b[(i + 1) % 4].y) * (b[i].y - b[(i + 1) % 4].y));
  }
  area = fabs(float(area / 2.0));
  const float distance = area * unclip_ratio / dist;
  for (int i = 0; i < 4; ++i) p[i] = {(long long)int(b[i].x), (long long)int(b[i].y)};
  const int npoly = (int)clipper_offset_round(p, distance).size();
  RRect r = unclip(b, unclip_ratio);
  out14[0] = r.cx; out14[1] = r.cy; out14[2] = r.w; out14[3] = r.h; out14[4] = r.angle;
  float ssid;
  Pt2f c[4];
  get_mini_boxes(r, ssid, c);
  out14[5] = ssid;
  for (int i = 0; i < 4; ++i) { out14[6 + 2 * i] = c[i].x; out14[7 + 2 * i] = c[i].y; }
  return npoly;
}

float oracle_box_score_fast(const float* box8, const float* pred, int H, int W, int cv_compat) {
  Pt2f a[4];
  for (int i = 0; i < 4; ++i) a[i] = {box8[2 * i], box8[2 * i + 1]};
  return box_score_fast(a, pred, H, W, fill_rule(cv_compat));
}

void oracle_fill_poly(uint8_t* mask, int W, int H, const int* pts_xy, int n, int cv_compat) {
  std::vector<Pt> p(n);
  for (int i = 0; i < n; ++i) p[i] = {pts_xy[2 * i], pts_xy[2 * i + 1]};
  fill_poly(mask, W, H, p, fill_rule(cv_compat));
}

// crop rectangle of ocr_worker.cpp:245-258: boundingRect(Point2f of int coords) & image rect.  returns 0 if empty
int oracle_crop_rect(const int* box8, int rows, int cols, int* x, int* y, int* w, int* h) {
  int xs[4] = {box8[0], box8[2], box8[4], box8[6]}, ys[4] = {box8[1], box8[3], box8[5], box8[7]};
  int x0 = *std::min_element(xs, xs + 4), x1 = *std::max_element(xs, xs + 4);
  int y0 = *std::min_element(ys, ys + 4), y1 = *std::max_element(ys, ys + 4);
  int rx = x0, ry = y0, rw = x1 - x0 + 1, rh = y1 - y0 + 1;
  // Rect &= Rect(0,0,cols,rows)
  int ix0 = std::max(rx, 0), iy0 = std::max(ry, 0);
  int ix1 = std::min(rx + rw, cols), iy1 = std::min(ry + rh, rows);
  int iw = ix1 - ix0, ih = iy1 - iy0;
  if (iw <= 0 || ih <= 0) { *x = *y = *w = *h = 0; return 0; }
  *x = ix0; *y = iy0; *w = iw; *h = ih;
  return 1;
}

// greedy CTC decode of one line from per-step (argmax, maxprob).  returns kept length or -1 when the
// reference `continue`s (NaN score: nothing kept).
int oracle_ctc_decode(const int* amax, const float* pmax, int T, int* ids, float* score) {
  int last_index = 0, count = 0;
  float s = 0.f;
  for (int n = 0; n < T; ++n) {
    int idx = amax[n];
    if (idx > 0 && (!(n > 0 && idx == last_index))) {
      s += pmax[n];
      ids[count] = idx;
      count += 1;
    }
    last_index = idx;
  }
  s /= count;
  if (std::isnan(s)) return -1;
  *score = s;
  return count;
}

// Utility::argsort (std::sort on indices by ratio).  libstdc++'s introsort is restated as-is by
// calling it: the oracle and the reference share the host C++ library semantics here.
void oracle_argsort(const float* v, int n, int* idx) {
  std::vector<size_t> a(n);
  for (int i = 0; i < n; ++i) a[i] = i;
  std::sort(a.begin(), a.end(), [v](size_t p1, size_t p2) { return v[p1] < v[p2]; });
  for (int i = 0; i < n; ++i) idx[i] = (int)a[i];
}
// The same with ties kept in input order: what std::sort does in the reference's own toolchain (MSVC STL: insertion
// sort for ranges of at most _ISORT_MAX = 32 elements) for up to 32 crops.  PARITY UNPINNED beyond 32 elements:
// MSVC's median-guess quicksort is not restated.
void oracle_argsort_stable(const float* v, int n, int* idx) {
  std::vector<size_t> a(n);
  for (int i = 0; i < n; ++i) a[i] = i;
  std::stable_sort(a.begin(), a.end(), [v](size_t p1, size_t p2) { return v[p1] < v[p2]; });
  for (int i = 0; i < n; ++i) idx[i] = (int)a[i];
}

}  // extern "C"
