// Runtime switches of the library, read from the environment ONCE per process (the first call of rt_options(), which
// ocr_rt_init makes before any handle exists) - INTEGRATION.md documents every field.  They select other kernel shapes /
// launch lists for A/B measurements and for the tests that drive production-only paths with small inputs; results are
// identical in every setting.  Every switch here is listed in INTEGRATION.md and driven by tests/test_gpu_parity.py
// (test_ab_switches_do_not_change_results); the A/B switches of settled questions were retired in round 5.  Three switches are per HANDLE instead (read when the handle is created): OCR_GRAPH,
// OCR_PIPE_PHASES, OCR_DET_LANES.
#pragma once
#include <string>

namespace ocr {

struct RtOptions {
  bool fuse = true;            // OCR_FUSE=0: the launch list without its bind-time fusions (folded SE gates, fused depthwise blocks, row
                               // sums out of the depthwise convs, fused DB head, two-pass RSE blocks, folded concats): one launch per plan op
  long fuse_gap_min = 65536;   // OCR_FUSE_GAP_MIN=n (tests: small shapes take the row-sum path too)
  bool conv_small_nt = true;   // OCR_CONV_SMALL_NT=0
  bool conv_mt2 = true;        // OCR_CONV_MT2=0
  bool conv_mt2_force = false; // OCR_CONV_MT2=force (tests): two pixel tiles per wave whatever the launch's size and K
  bool conv_c24 = true;        // OCR_CONV_C24=0: the 3x3 96 -> 24 convs on the 32-column tile kernel (the 4x4x1 kernel's fallback)
  bool attn_line = true;       // OCR_ATTN_LINE=0: attention as a wave per (line, head) also for lines of <= 64 tokens (A/B)
  bool mfma_x16 = true;        // OCR_MFMA_X16=0: precision "fp16" keeps v_mfma_f32_32x32x8_f16 in the big 1x1 convs (A/B)
  bool dw_lds = true;          // OCR_DW_LDS=0: the low-map 5x5 depthwise layers keep dw_conv_kernel (A/B; results are identical)
  bool xdw = true;             // OCR_XDW=0: the classifier's expand 1x1 -> depthwise 5x5 pairs as two launches (A/B; kernels_xdw.hip)
  bool dwpw2 = true;           // OCR_DWPW2=0: the fused depthwise blocks keep their first form (registers -> LDS) where the LDS-DMA form exists (A/B)
  int dwpw_items = 32;         // OCR_DWPW_ITEMS=n
  int dwpw_force_upw = 0;      // OCR_DWPW_FORCE_UPW=n (tests)
  bool dwpw_t4_thin = false;   // OCR_DWPW_T4=thin
  int trace_slice = 0;         // OCR_TRACE_SLICE=n (tests); 0 = unlimited
};
const RtOptions& rt_options();

// Fault injection for the tests (ocr_selftest_refuse_launch): a network launch whose name contains this string is
// refused as if its launcher had rejected the shape; empty = off.
std::string rt_refuse_launch();  // a copy (taken under the setter's lock)
void rt_set_refuse_launch(const char* substr);

}  // namespace ocr
