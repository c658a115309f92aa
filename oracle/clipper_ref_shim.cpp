// TEST INFRASTRUCTURE.  C-ABI shim (this build's own code) around the REFERENCE's ClipperLib,
// compiled together with /root/reference/src/clipper.cpp into oracle/_ref/libclipper_ref.so by
// oracle/Makefile.  It makes exactly the calls DBPostProcessor::UnClip makes
// (/root/reference/src/postprocess_op.cpp:46-55): AddPath(jtRound, etClosedPolygon), Execute(soln, delta).
#include <paddle_ocr/clipper.h>

extern "C" int clipper_ref_offset(const long long* xy, int n, double delta, long long* out_xy, int cap_pts,
                                  int* path_sizes, int max_paths, int* npaths) {
  ClipperLib::ClipperOffset offset;
  ClipperLib::Path p;
  for (int i = 0; i < n; ++i) p.emplace_back((ClipperLib::cInt)xy[2 * i], (ClipperLib::cInt)xy[2 * i + 1]);
  offset.AddPath(p, ClipperLib::jtRound, ClipperLib::etClosedPolygon);
  ClipperLib::Paths soln;
  if (!offset.Execute(soln, delta)) return -1;
  *npaths = (int)soln.size();
  int k = 0;
  for (size_t j = 0; j < soln.size(); ++j) {
    if ((int)j < max_paths) path_sizes[j] = (int)soln[j].size();
    for (auto& pt : soln[j]) {
      if (k >= cap_pts) return -2;
      out_xy[2 * k] = pt.X;
      out_xy[2 * k + 1] = pt.Y;
      ++k;
    }
  }
  return k;
}
