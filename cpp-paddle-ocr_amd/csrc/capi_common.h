// Shared helpers of the C-ABI translation units.
#pragma once
#include "rt_options.h"
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <memory>
#include <string>

#include "../../include/ocr_hip.h"
#include "net.h"

namespace ocr {

void set_last_error(const std::string& msg);
void priority_anchor(int device_id, int want = 1);  // up to `want` (<= 2) idle high-priority streams per device, created once per process (capi_net.hip)
const char* embedded_plan(const char* kind);  // "det" | "cls" | "rec" -> plan text or nullptr

// Resolves the weights file of a model directory the way the reference resolves its model file
// (ocr_det.cpp:28-40 probes four names): inference.pdiparams, model.pdiparams, then the build's
// synthetic.pdiparams.  Loads names from the .pdmodel next to it.
// kind ("det" | "cls" | "rec"): the graph in model_dir must carry the signature stored in that plan
bool load_model_dir(const std::string& model_dir, const char* weights_override, const char* kind, WeightMap& w, std::string& err);

// BASELINE configs[4] (server networks; hand-written plans, NOT reference artifacts): a model directory WITHOUT a .pdmodel whose
// arch.txt names a server plan ("srv_det" | "srv_rec"); parameter names / shapes come from the plan's parameter table.
// server_arch: the tag of such a directory, "" for an ordinary (graph-carrying) model directory
std::string server_arch(const std::string& model_dir);
bool load_server_model_dir(const std::string& model_dir, const char* kind, WeightMap& w, std::string& err);

int fail(int code, const std::string& msg);

// ocr_det_cfg.cv_compat -> OCR_CV_45 | OCR_CV_410 (0: OCR_CV_COMPAT from the environment, else OCR_CV_410); any other
// value is passed through and refused by DetStage::create
inline int resolve_cv_compat(int v) {
  if (v != OCR_CV_DEFAULT) return v;
  if (const char* e = getenv("OCR_CV_COMPAT")) return atoi(e);
  return OCR_CV_410;
}

#define CAPI_HIP(expr)                                                             \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return ocr::fail(OCR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace ocr
