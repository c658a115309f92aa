// Runtime switches of the library, read from the environment ONCE per process (the first call of rt_options(), which
// ocr_rt_init makes before any handle exists) - INTEGRATION.md documents every field.  They select other kernel shapes /
// launch lists for A/B measurements and for the tests that drive production-only paths with small inputs; results are
// identical in every setting.  Three switches are per HANDLE instead (read when the handle is created): OCR_GRAPH,
// OCR_PIPE_PHASES, OCR_DET_LANES.
#pragma once
#include <string>

namespace ocr {

struct RtOptions {
  bool fuse_gate = true;       // OCR_FUSE_GATE=0
  bool fuse_dwpw = true;       // OCR_FUSE_DWPW=0
  bool fuse_gap = true;        // OCR_FUSE_GAP=0
  long fuse_gap_min = 65536;   // OCR_FUSE_GAP_MIN=n
  bool fuse_dbhead = true;     // OCR_FUSE_DBHEAD=0
  bool dbhead_mfma = true;     // OCR_DBHEAD_MFMA=0: the fused DB head's first stage on the VALU (db_head_kernel)
  bool fuse_rse = true;        // OCR_FUSE_RSE=0
  bool fuse_concat = true;     // OCR_FUSE_CONCAT=0: the DB neck's concat is materialised (A/B; results are identical)
  bool fuse_mb = false;        // OCR_FUSE_MB=1: the classifier's SE bottlenecks as one launch each (kernels_mb.hip: correct, measured slower, off)
  int conv_impl = 0;           // OCR_CONV_IMPL=direct (1) | lds (2); 0 = per shape
  bool conv_small_nt = true;   // OCR_CONV_SMALL_NT=0
  int conv_nt_max = 4;         // OCR_CONV_NT_MAX=n
  bool conv_mt2 = true;        // OCR_CONV_MT2=0
  bool conv_mt2_force = false; // OCR_CONV_MT2=force (tests): two pixel tiles per wave whatever the launch's size and K
  bool conv_c24 = true;        // OCR_CONV_C24=0
  bool conv_tile = true;       // OCR_CONV_TILE=0
  int dw_patch_to = 0, dw_patch_r = 0;  // OCR_DW_PATCH=TOxR; 0 = per shape
  bool attn_line = true;       // OCR_ATTN_LINE=0: attention as a wave per (line, head) also for lines of <= 64 tokens (A/B)
  bool mfma_x16 = true;        // OCR_MFMA_X16=0: precision "fp16" keeps v_mfma_f32_32x32x8_f16 in the big 1x1 convs (A/B)
  bool dw_lds = true;          // OCR_DW_LDS=0: the low-map 5x5 depthwise layers keep dw_conv_kernel (A/B; results are identical)
  bool dwpw2 = true;           // OCR_DWPW2=0: the fused depthwise blocks keep their first form (registers -> LDS) where the LDS-DMA form exists (A/B)
  int dwpw_items = 32;         // OCR_DWPW_ITEMS=n
  int dwpw_force_upw = 0;      // OCR_DWPW_FORCE_UPW=n (tests)
  bool dwpw_t4_thin = false;   // OCR_DWPW_T4=thin
  int trace_slice = 0;         // OCR_TRACE_SLICE=n (tests); 0 = unlimited
  bool prio_anchor = true;     // OCR_PRIO_ANCHOR=0
};
const RtOptions& rt_options();

// Fault injection for the tests (ocr_selftest_refuse_launch): a network launch whose name contains this string is
// refused as if its launcher had rejected the shape; empty = off.
std::string rt_refuse_launch();  // a copy (taken under the setter's lock)
void rt_set_refuse_launch(const char* substr);

}  // namespace ocr
