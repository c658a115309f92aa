// Shared helpers of the C-ABI translation units.
#pragma once
#include "rt_options.h"
#include <hip/hip_runtime.h>

#include <memory>
#include <string>

#include "../../include/ocr_hip.h"
#include "net.h"

namespace ocr {

void set_last_error(const std::string& msg);
void priority_anchor(int device_id, bool again = false);             // one idle high-priority stream per device (capi_net.hip)
const char* embedded_plan(const char* kind);  // "det" | "cls" | "rec" -> plan text or nullptr

// Resolves the weights file of a model directory the way the reference resolves its model file
// (ocr_det.cpp:28-40 probes four names): inference.pdiparams, model.pdiparams, then the build's
// synthetic.pdiparams.  Loads names from the .pdmodel next to it.
// kind ("det" | "cls" | "rec"): the graph in model_dir must carry the signature stored in that plan
bool load_model_dir(const std::string& model_dir, const char* weights_override, const char* kind, WeightMap& w, std::string& err);

int fail(int code, const std::string& msg);

#define CAPI_HIP(expr)                                                             \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      return ocr::fail(OCR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

}  // namespace ocr
