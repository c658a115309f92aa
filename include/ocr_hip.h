/* libocr_hip — C-ABI of the MI355X-native OCR hot path (det -> cls -> rec).
 *
 * Drop-in boundary: these entry points are what a maintainer of sssxyd/cpp-paddle-ocr binds in
 * place of the Paddle Inference predictor + OpenCV pre/post code inside
 *   DBDetector::Run        /root/reference/src/ocr_det.cpp:93-176   (include/paddle_ocr/ocr_det.h:95-97)
 *   Classifier::Run        /root/reference/src/ocr_cls.cpp:23-106   (include/paddle_ocr/ocr_cls.h:81-82)
 *   CRNNRecognizer::Run    /root/reference/src/ocr_rec.cpp:24-135   (include/paddle_ocr/ocr_rec.h:92-95)
 *   OCRWorker::processRequest  /root/reference/src/ocr_worker.cpp:213-311
 * Plain pointers and sizes only; every function returns 0 on success and a negative code on
 * failure (never exit(): compare ocr_det.cpp:41-45).  ocr_last_error() gives the message of the
 * calling thread's last failure.  One handle = one HIP stream + its device buffers; use one
 * handle per host thread (the reference's stage objects are not re-entrant either).
 * INTEGRATION.md shows the C++ shim classes that keep the reference signatures on top of this.
 */
#ifndef OCR_HIP_H_
#define OCR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCR_OK 0
#define OCR_ERR_ARG (-1)      /* bad argument */
#define OCR_ERR_MODEL (-2)    /* model files missing / do not match the supported graph */
#define OCR_ERR_DEVICE (-3)   /* HIP runtime failure (message has the HIP error string) */
#define OCR_ERR_CAPACITY (-4) /* caller buffer too small */

const char* ocr_last_error(void);
/* Selects the HIP device for the calling thread; fails when no gfx950 device is visible. */
int ocr_rt_init(int device_id);
int ocr_rt_device_count(void);
/* How the library's host threads wait for their streams (process-wide; round 6): 1 = block on an event created with
 * hipEventBlockingSync (the thread sleeps until the interrupt; default), 0 = hipStreamSynchronize's spin (what rounds 1-5 did:
 * three spinning host threads per pipeline handle, 24 busy cores for the eight ranks of a node).  OCR_WAIT_MODE=spin|block in
 * the environment sets the initial mode.  The reference has no counterpart: its workers wait inside Paddle's predictor. */
int ocr_rt_set_wait_mode(int mode);
int ocr_rt_get_wait_mode(void);

/* An image view: CV_8UC3 BGR rows, possibly a non-contiguous ROI (row_stride in bytes) —
 * the `const cv::Mat&` of the reference signatures. */
typedef struct ocr_img {
  const uint8_t* data;
  int rows, cols;
  size_t row_stride;
} ocr_img;

/* ---------------------------------------------------------------- detector */
/* Field-for-field the constructor arguments of DBDetector (ocr_det.h:60-75) that still mean
 * something without Paddle; defaults of ocr_det_cfg_default() are the literals OCRWorker passes
 * (ocr_worker.cpp:21-35: "max", 512, 0.2, 0.4, 1.8, "fast", no dilation). */
typedef struct ocr_det_cfg {
  const char* model_dir; /* directory holding inference.pdmodel + inference.pdiparams */
  int device_id;
  const char* limit_type; /* "max" | "min" */
  int limit_side_len;
  double det_db_thresh;
  double det_db_box_thresh;
  double det_db_unclip_ratio;
  const char* det_db_score_mode; /* "fast" | "slow" */
  int use_dilation;
  /* "fp32": the bit-exact arithmetic contract (default).  "fp16": the reference's TensorRT precision switch
   * (ocr_det.cpp:50-56) - activation tensors stored as f16, matrix-core products as f16 instructions with f32 accumulation,
   * every other chain and reduction in f32: results within a stated tolerance of fp32, not identical.
   * "int8" is rejected. */
  const char* precision;
  int max_batch;         /* images per ocr_det_run_batch call the handle is sized for (>=1) */
  /* Which OpenCV the reference binary was built against, where two 4.x releases differ on this path (DESIGN.md
   * section 5).  Today one rule depends on it: cv::fillPoly's scan fill behind BoxScoreFast / PolygonScoreAcc
   * (postprocess_op.cpp:205-251) - OCR_CV_45: the rule of OpenCV 4.0 - 4.5.1, spans [ceil(xa), floor(xb)] of unshifted
   * edges; OCR_CV_410: the rule of 4.5.2 and later (edges moved by half a pixel, spans [floor, floor]).  0 = default:
   * the environment variable OCR_CV_COMPAT ("45" | "410") if set, else OCR_CV_410 - the reference's README builds with
   * a 2025 MSVC toolchain against vcpkg's current opencv port (README.md:105-118). */
  int cv_compat;
} ocr_det_cfg;
#define OCR_CV_DEFAULT 0
#define OCR_CV_45 45
#define OCR_CV_410 410
void ocr_det_cfg_default(ocr_det_cfg* cfg);

typedef struct ocr_det ocr_det;
int ocr_det_create(const ocr_det_cfg* cfg, ocr_det** out);
void ocr_det_destroy(ocr_det* h);
/* DBDetector::Run for one image.  boxes: cap x 8 int32 (4 points x,y clockwise from top-left, source
 * image coordinates), *n = boxes found; times[3] = pre / infer / post in ms (ocr_det.cpp:168-175). */
int ocr_det_run(ocr_det* h, const ocr_img* img, int32_t* boxes, int cap, int* n, double times[3]);
/* Same for `count` images of identical size in one device pass (batch on the GEMM row axis).
 * boxes: count x cap x 8; n: count entries. */
int ocr_det_run_batch(ocr_det* h, const ocr_img* imgs, int count, int32_t* boxes, int cap, int* n, double times[3]);
/* Parity taps: sizes of the network input of the last run, its probability map (rows*cols f32)
 * and bitmap. */
int ocr_det_last_shape(ocr_det* h, int* count, int* rows, int* cols);
int ocr_det_prob_map(ocr_det* h, int index, float* out, size_t cap_floats);
int ocr_det_bitmap(ocr_det* h, int index, uint8_t* out, size_t cap_bytes);
int ocr_det_resized(ocr_det* h, int index, uint8_t* out, size_t cap_bytes); /* u8 BGR after ResizeImgType0 */
/* Post-processing alone on a caller-supplied probability map (rows x cols f32, values in [0,1]):
 * threshold (+dilate) -> contours -> boxes -> FilterTagDetRes against a src_rows x src_cols image. */
int ocr_det_post(ocr_det* h, const float* prob, int rows, int cols, int src_rows, int src_cols, int32_t* boxes, int cap,
                 int* n);

/* ---------------------------------------------------------------- classifier */
typedef struct ocr_cls_cfg {
  const char* model_dir;
  int device_id;
  double cls_thresh; /* stored, never consulted — as in the reference (ocr_cls.cpp never reads it) */
  int cls_batch_num;
  const char* precision;
} ocr_cls_cfg;
void ocr_cls_cfg_default(ocr_cls_cfg* cfg);
typedef struct ocr_cls ocr_cls;
int ocr_cls_create(const ocr_cls_cfg* cfg, ocr_cls** out);
void ocr_cls_destroy(ocr_cls* h);
/* Classifier::Run: labels/scores are caller-sized to n (ocr_worker.cpp:271-272). */
int ocr_cls_run(ocr_cls* h, const ocr_img* imgs, int n, int* labels, float* scores, double times[3]);
/* tap: softmax [n][2] of the last run */
int ocr_cls_probs(ocr_cls* h, float* out, size_t cap_floats);

/* ---------------------------------------------------------------- recognizer */
typedef struct ocr_rec_cfg {
  const char* model_dir;
  int device_id;
  const char* label_path; /* ppocr_keys_v1.txt; "#"/" " are added like ocr_rec.h:82-84 */
  int rec_batch_num;
  int rec_img_h, rec_img_w;
  const char* precision;
  /* how lines with EQUAL w/h ratio are ordered before the batches of rec_batch_num are cut (Utility::argsort is
   * std::sort, utility.cpp:192-203, whose order of ties is the host library's): OCR_SORT_STD = this build's
   * std::sort (libstdc++ introsort, what the oracle runs); OCR_SORT_STABLE = ties keep their input order, which is
   * what MSVC's std::sort (the reference's toolchain) does for up to 32 crops (insertion sort below _ISORT_MAX).  Beyond 32
   * crops of one image MSVC's quicksort order of ties is NOT restated: OCR_SORT_STABLE keeps input order there too (a
   * documented choice, not MSVC's order); OCR_SORT_MSVC_STRICT is OCR_SORT_STABLE that REFUSES such an image - more than 32
   * crops of which two have equal ratios - with OCR_ERR_ARG and a message naming the image, instead of answering in an
   * order the reference's build may not produce. */
  int sort_mode;
} ocr_rec_cfg;
enum { OCR_SORT_STD = 0, OCR_SORT_STABLE = 1, OCR_SORT_MSVC_STRICT = 2 };
void ocr_rec_cfg_default(ocr_rec_cfg* cfg);
typedef struct ocr_rec ocr_rec;
int ocr_rec_create(const ocr_rec_cfg* cfg, ocr_rec** out);
void ocr_rec_destroy(ocr_rec* h);
/* CRNNRecognizer::Run.  For line i: ids[i*max_len .. +lens[i]) are the kept CTC class ids
 * (dictionary indices, blank and repeats removed), scores[i] the mean max-probability.  Lines the
 * reference leaves untouched (NaN score: no kept step) get lens[i] = 0, scores[i] = 0. */
int ocr_rec_run(ocr_rec* h, const ocr_img* imgs, int n, int32_t* ids, int max_len, int* lens, float* scores,
                double times[3]);
/* UTF-8 label of a class id (valid until the handle is destroyed); NULL when out of range. */
const char* ocr_rec_label(ocr_rec* h, int id);
int ocr_rec_num_classes(ocr_rec* h);
/* tap: per-step arg max / max prob of line `index` of the last run (T entries), and T */
int ocr_rec_steps(ocr_rec* h, int index, int32_t* amax, float* pmax, int cap, int* T);


/* ---------------------------------------------------------------- pipeline */
/* OCRWorker::processRequest (/root/reference/src/ocr_worker.cpp:213-311) for a batch of images in one
 * device pass: det -> crops (crop_mode) -> [cls -> in-place 180 degree rotation] -> rec -> CTC.
 * The three stage handles live on one device; images are uploaded once and the crops are taken
 * from the device copy (the reference's ROI views of the request's cv::Mat clone). */
typedef struct ocr_pipe_cfg {
  ocr_det_cfg det;
  ocr_cls_cfg cls;
  ocr_rec_cfg rec;
  int enable_cls; /* OCRWorker(..., enable_cls = false) */
  int crop_mode;  /* OCR_CROP_BOUNDING_RECT: ROI views, what the worker does (ocr_worker.cpp:244-259);
                   * OCR_CROP_ROTATE: Utility::GetRotateCropImage per box (utility.cpp:137-190) */
  int phases;     /* chains a batch is run on, 1..4 (0 = the default, 2): the batch is cut into that many parts that run side by
                   * side, each with its own stage objects, streams and host thread (the latency-bound phases of one run
                   * under the dense kernels of another; results are per image and do not change); 1 = one chain, one
                   * kernel at a time owns the device.  OCR_PIPE_PHASES in the environment overrides. */
} ocr_pipe_cfg;
enum { OCR_CROP_BOUNDING_RECT = 0, OCR_CROP_ROTATE = 1 };
void ocr_pipe_cfg_default(ocr_pipe_cfg* cfg);
/* WordResult (ocr_worker.h:34-38): text as class ids ids[ids_off .. ids_off+ids_len) */
typedef struct ocr_word {
  int32_t box[8];
  int32_t ids_off, ids_len;
  float confidence;
} ocr_word;
typedef struct ocr_pipe ocr_pipe;
int ocr_pipe_create(const ocr_pipe_cfg* cfg, ocr_pipe** out);
void ocr_pipe_destroy(ocr_pipe* h);
/* imgs: `count` images (any mix of sizes).  Images of equal size share one det pass; the text lines of ALL images
 * then go through one cls pass and one rec pass (each image's lines are still sorted and cut into batches of
 * rec_batch_num on their own, ocr_rec.cpp:34-57: results do not depend on what else is in the call).
 * words: cap_words entries, image i owns words[word_off[i] .. word_off[i]+nwords[i]); ids: cap_ids class ids.
 * times[3] = det / cls / rec wall ms.  = ocr_pipe_stage(slot 0) + ocr_pipe_run_staged(slot 0). */
int ocr_pipe_run(ocr_pipe* h, const ocr_img* imgs, int count, ocr_word* words, int cap_words, int* word_off, int* nwords,
                 int32_t* ids, int cap_ids, double times[3]);
/* The same with inputs already resident in HBM: `count` packed BGR images of rows x cols (one every
 * rows*cols*3 bytes, device pointer), and optionally `dev_prob` = count probability maps of the
 * detector's input resolution that replace the network's map for thresholding/scoring (benchmark
 * protocol for synthetic weights; the network still runs and is timed). */
int ocr_pipe_run_device(ocr_pipe* h, const void* dev_bgr, int rows, int cols, int count, const float* dev_prob, ocr_word* words,
                        int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3]);
/* Double-buffered input: ocr_pipe_stage copies the host images into pinned memory (a few
 * host threads) and starts their upload on a copy stream, then returns; ocr_pipe_run_staged waits for that upload on
 * the device and runs the batch.  Two slots (0, 1): one host thread stages batch k+1 into the other slot while
 * another is inside ocr_pipe_run_staged for batch k; a slot may be run any number of times (its images then are
 * "already resident in HBM").
 * ocr_pipe_slot_probs (benchmark protocol): per staged image a pointer (host or device memory) to a probability map
 * of the detector's input resolution (ocr_pipe_det_shape) that replaces the network's map, as dev_prob of
 * ocr_pipe_run_device; the maps stay attached while the slot is re-staged with the same sizes in the same order. */
int ocr_pipe_stage(ocr_pipe* h, int slot, const ocr_img* imgs, int count);
int ocr_pipe_slot_probs(ocr_pipe* h, int slot, const float* const* probs, int count);
int ocr_pipe_run_staged(ocr_pipe* h, int slot, ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids,
                        int cap_ids, double times[3]);
/* Two batches in flight on one handle (round 5).  The reference's throughput shape is a pool of workers on one GPU
 * (/root/reference/src/gpu_worker_pool.cpp:12-16 pins every worker to GPU 0, ocr_worker.cpp:213-311 runs one request at a
 * time per worker): consecutive requests overlap, the halves of one request do not.  The `_on` forms run the WHOLE batch on
 * ONE of the handle's chains (0 <= chain < ocr_pipe_cfg.phases, default 2; own stage objects, streams, arenas and clone
 * buffer per chain) instead of cutting it over all of them.  Calls that name different chains - and, for the staged form,
 * different slots - may run concurrently from different host threads: chain 0 works on batch k while chain 1 works on
 * batch k+1, each call returning its own batch's results.  Results are those of the plain calls (per image, independent
 * of the batch composition).  Two calls on the SAME chain (or slot) at once are the caller's error. */
int ocr_pipe_run_device_on(ocr_pipe* h, int chain, const void* dev_bgr, int rows, int cols, int count, const float* dev_prob,
                           ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids, int cap_ids, double times[3]);
int ocr_pipe_run_staged_on(ocr_pipe* h, int chain, int slot, ocr_word* words, int cap_words, int* word_off, int* nwords, int32_t* ids,
                           int cap_ids, double times[3]);
/* JPEG inputs with the pixel half of the decoder on the device: the caller runs the
 * bit-serial entropy decoding (host/jpeg_decode.h, Decoder::decode_coefficients) and hands over quantised DCT
 * coefficients; dequantisation, IDCT, chroma upsampling and colour conversion run on the copy stream straight into
 * the slot - otherwise exactly ocr_pipe_stage.  Results equal decoding on the host first, bit for bit. */
typedef struct ocr_jpeg_comp {
  const int16_t* coef; /* host: bw*bh blocks x 64 quantised coefficients, blocks row-major, natural order inside */
  uint16_t quant[64];  /* quantisation table, natural order */
  int bw, bh;          /* blocks per row / column (padded to whole MCUs) */
  int dw, dh;          /* component size in samples: ceil(cols*h/hmax), ceil(rows*v/vmax) */
} ocr_jpeg_comp;
typedef struct ocr_jpeg_img {
  int rows, cols, ncomp; /* ncomp 1 (grey) or 3 (YCbCr) */
  int hmax, vmax;        /* luma sampling factors, chroma 1x1: 1x1 (4:4:4), 2x1 (4:2:2), 2x2 (4:2:0) */
  ocr_jpeg_comp comp[3];
} ocr_jpeg_img;
int ocr_pipe_stage_jpeg(ocr_pipe* h, int slot, const ocr_jpeg_img* imgs, int count);
/* the same device decode of one image with the pixels copied back to the host (tests, tools) */
int ocr_jpeg_decode(const ocr_jpeg_img* img, int device_id, uint8_t* bgr, size_t cap_bytes);
const char* ocr_pipe_label(ocr_pipe* h, int id);
/* network input size the detector uses for a rows x cols image (ResizeImgType0) */
int ocr_pipe_det_shape(ocr_pipe* h, int rows, int cols, int* net_rows, int* net_cols);
/* binding-cache statistics summed over the pipeline's networks: out = {network runs, runs that had to bind a new
 * shape (plan the arena, build the launch list), runs replayed from a recorded hipGraph} */
int ocr_pipe_stats(ocr_pipe* h, long long out[3]);
/* HIP-event timing of every kernel launch of the three networks during subsequent runs */
int ocr_pipe_timing(ocr_pipe* h, int enable);
/* restrict the events to launches whose name contains substr (NULL or "": all launches) */
int ocr_pipe_timing_filter(ocr_pipe* h, const char* substr);
int ocr_pipe_timing_report(ocr_pipe* h, char* buf, size_t cap);

/* Utility::GetRotateCropImage (/root/reference/src/utility.cpp:137-190) for n boxes (n x 8 ints,
 * x0,y0..x3,y3) of one host image: bounding-box crop, cv::getPerspectiveTransform onto
 * int(|p0p1|) x int(|p0p3|), cv::warpPerspective (bilinear, constant-0 border: the reference passes
 * BORDER_REPLICATE in the flags slot), and a 90-degree turn when rows >= 1.5 cols.
 * Crop k is out[out_off[k] .. out_off[k+1]) as packed BGR of out_rows[k] x out_cols[k];
 * out_off has n+1 entries.  A box whose bounding box is empty or leaves the image is an OCR_ERR_ARG
 * (the reference would throw from cv::Mat::operator() inside a noexcept function). */
int ocr_rotate_crop(const uint8_t* bgr, int rows, int cols, size_t row_stride, const int32_t* boxes, int n, uint8_t* out,
                    size_t out_cap, size_t* out_off, int* out_rows, int* out_cols);
/* size of that crop without computing it (host arithmetic only) */
int ocr_rotate_crop_shape(int rows, int cols, const int32_t* box, int* out_rows, int* out_cols);

/* The classifier's in-place rotations (/root/reference/src/ocr_worker.cpp:255-262: `cv::rotate(img_list[i], img_list[i], 1)`
 * on crops that are ROI views of ONE image, so overlapping crops see each other's result): rotates the n rectangles
 * (n x 4 ints: x, y, w, h, inside the image) of a host image by 180 degrees in list order, on the device.  Rectangles
 * that are connected through intersections form a group that keeps list order; groups run concurrently. */
int ocr_rotate180_rois(uint8_t* bgr, int rows, int cols, size_t row_stride, const int32_t* rects, int n);

/* device memory helpers for callers without a HIP binding of their own (bench, tests) */
int ocr_dev_alloc(void** p, size_t bytes);
int ocr_dev_free(void* p);
int ocr_dev_upload(void* dst, const void* src, size_t bytes);
int ocr_dev_download(void* dst, const void* src, size_t bytes);
int ocr_dev_sync(void);

/* ---------------------------------------------------------------- raw network taps (parity tests) */
typedef struct ocr_net ocr_net;
/* kind: "det" | "cls" | "rec".  weights: path of a .pdiparams file (NULL: <model_dir>/inference.pdiparams,
 * falling back to <model_dir>/synthetic.pdiparams). */
int ocr_net_create(const char* kind, const char* model_dir, const char* weights, int device_id, ocr_net** out);
/* the same with the stages' precision parameter ("fp32" | "fp16") */
int ocr_net_create_precision(const char* kind, const char* model_dir, const char* weights, int device_id, const char* precision,
                             ocr_net** out);
void ocr_net_destroy(ocr_net* h);
/* x: host f32 [N,H,W,3] (already normalised, BGR order).  keep_all: 0 = production launch list and arena reuse;
 * 1 = every plan tensor materialised in its own slot (no fusion); 2 = the production launch list (SE gates folded,
 * depthwise->pointwise pairs fused) with every tensor it writes kept in its own slot. */
int ocr_net_forward(ocr_net* h, const float* x, int N, int H, int W, int keep_all);
/* Ragged batch (rec only): N text lines of height H, line n of width widths[n]; x = the lines' [H][widths[n]][3]
 * blocks one after the other.  One launch list for all widths (the recognizer's production path: every batch of
 * rec_batch_num lines has its own tensor width, src/ocr_rec.cpp:47-72); each line's results are those of
 * ocr_net_forward on that line alone.  ocr_net_fetch then returns a tensor's lines one after the other with
 * dims = {1, 1, total pixels, C} (per-line vectors such as the SE gates: {N, 1, 1, C}). */
int ocr_net_forward_ragged(ocr_net* h, const float* x, int N, int H, const int* widths, int keep_all);
/* Ragged batch of IMAGES (det only): N images of their own sizes heights[n] x widths[n] (multiples of 32, what
 * ResizeImgType0 produces, src/preprocess_op.cpp:84-88) in one launch list - the detector's path for batches of mixed
 * sizes; x = the images' [h][w][3] blocks one after the other.  keep_all 0 or 2.  ocr_net_fetch returns a tensor's
 * images one after the other, dims = {1, 1, total pixels at that tensor's resolution, C}. */
int ocr_net_forward_ragged_images(ocr_net* h, const float* x, int N, const int* heights, const int* widths, int keep_all);
int ocr_net_num_tensors(ocr_net* h);
/* 1 if the last forward wrote tensor `tid` to device memory (fused-away tensors never exist), else 0 */
int ocr_net_tensor_exists(ocr_net* h, int tid);
/* tid < 0: network output.  out receives logical NHWC; dims = {N,H,W,C}. */
int ocr_net_fetch(ocr_net* h, int tid, float* out, size_t cap_floats, int dims[4]);
/* HIP-event timing of the launches of subsequent ocr_net_forward calls */
int ocr_net_timing(ocr_net* h, int enable);
/* writes "name ms count flops bytes\n" lines */
int ocr_net_timing_report(ocr_net* h, char* buf, size_t cap);

/* ---------------------------------------------------------------- server networks (BASELINE configs[4]) - raw taps
 * "PP-OCRv4_server_det (ResNet50 backbone) + SVTR-large rec, fp16": the reference ships no such graph or weights (SURVEY.md
 * section 8d, cfg5) - the two plans (cpp-paddle-ocr_amd/plans/srv_det.plan, srv_rec.plan) are hand-written from the public
 * PaddleOCR model definitions, NOT reference artifacts; parameters are seeded (<model_dir>/synthetic.pdiparams, names and shapes
 * from the plan's parameter table).  What is matched is the reference's `precision` constructor argument
 * (/root/reference/src/ocr_det.cpp:50-57, /root/reference/include/paddle_ocr/ocr_det.h:60-89): "fp16" = f16 tensors, f16 matrix
 * instructions, f32 accumulation; "fp32" = the PARITY TWIN of the same launch list, bit-identical to the oracle's run of the
 * plan.  kind: "det" | "rec".  x: host f32 [N,H,W,3] (normalised; det: H, W multiples of 32; rec: 48 x 320).  Output tensor
 * (tid < 0): det = the probability map [N,H,W,1]; rec = the CTC logits [N,1,W/4,6625]. */
typedef struct ocr_srv_net ocr_srv_net;
int ocr_srv_net_create(const char* kind, const char* model_dir, int device_id, const char* precision, ocr_srv_net** out);
void ocr_srv_net_destroy(ocr_srv_net* h);
/* keep_all != 0: every tensor keeps its own arena slot (parity taps of intermediate tensors) */
int ocr_srv_net_forward(ocr_srv_net* h, const float* x, int N, int H, int W, int keep_all);
/* `iters` more runs on the input the last forward uploaded (timing loops) */
int ocr_srv_net_rerun(ocr_srv_net* h, int N, int H, int W, int iters);
int ocr_srv_net_num_tensors(ocr_srv_net* h);
int ocr_srv_net_fetch(ocr_srv_net* h, int tid, float* out, size_t cap_floats, int dims[4]);
int ocr_srv_net_timing(ocr_srv_net* h, int enable);
int ocr_srv_net_timing_report(ocr_srv_net* h, char* buf, size_t cap);

/* self-tests / fault injection (tests): ocr_selftest_refuse_launch - a network launch whose name contains `substr` is
 * refused as if its launcher had rejected the shape (NULL or "" switches it off): the run must fail with OCR_ERR_DEVICE
 * and a message, never abort.  ocr_selftest_lds_memo - the per-device dynamic-LDS attribute memo of the kernel
 * launchers exercised on faked device indices (no HIP call: runs without a GPU). */
int ocr_selftest_refuse_launch(const char* substr);
int ocr_selftest_lds_memo(void);

/* The device's ClipperOffset / UnClip alone, for vectors held OUTSIDE the library (tests/golden/unclip_ref*: outputs of the
 * reference's own compiled src/clipper.cpp) - the one place where the HIP code can face the reference itself.
 * ocr_selftest_unclip: ClipperOffset(jtRound, etClosedPolygon).AddPath(quad).Execute(delta) as DBPostProcessor::UnClip
 * calls it (/root/reference/src/postprocess_op.cpp:46-55, /root/reference/src/clipper.cpp:3779-4021) for n quads
 * (n x 8 int32: x0,y0..x3,y3) and their deltas; out: n x cap x 2 int64 (X, Y of the solution path), counts[n] = its vertex
 * count (0: Execute returned no path; -1: more than the kernel's 512-vertex scratch); trig (may be NULL): n x 3 doubles, the
 * round join's steps, m_sin, m_cos as the device's math library computes them (clipper.cpp:3800-3818), before any rounding.
 * ocr_selftest_unclip_box: the whole UnClip -> cv::minAreaRect -> GetMiniBoxes of one candidate
 * (/root/reference/src/postprocess_op.cpp:39-72,134-168) for n boxes (n x 8 float, GetMiniBoxes order) at one unclip ratio;
 * out14 per box: RotatedRect (cx, cy, w, h, angle), ssid, four corners (x, y); status per box: {POST_ERR bit or 0,
 * vertices of the offset polygon}. */
int ocr_selftest_unclip(const int32_t* quads, const double* deltas, int n, int64_t* out, int cap, int* counts, double* trig);
int ocr_selftest_unclip_box(const float* boxes, float unclip_ratio, int n, float* out14, int* status);

/* numerics probe (tests): out[8*n] = a/b, sqrt|a|, ocr_expf(a), fma(a,b,a), a*b+a, rint(a*log2e), hswish(a), hswish(b) */
int ocr_probe(const float* a, const float* b, float* out, int n);

#ifdef __cplusplus
}
#endif
#endif /* OCR_HIP_H_ */
