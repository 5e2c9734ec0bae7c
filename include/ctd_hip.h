/*
 * ctd_hip.h -- C ABI of the MI355X (gfx950) comic-text-detector hot path.
 *
 * This library is the third backend behind the reference's backend seam:
 *
 *     blks, mask, lines_map = self.net(img_in)          (reference inference.py:146)
 *
 * next to `TextDetBase.forward` (reference basemodel.py:240-244, torch) and
 * `TextDetBaseDNN.__call__` (reference basemodel.py:252-256, OpenCV-DNN/ONNX,
 * tensor names `images` -> `blk, seg, det`, reference utils/export.py:43-44).
 *
 * Plain pointers and sizes only; no torch types.  All pointers named *_dev are
 * device (HBM) pointers on the engine's device; everything else is host memory.
 * Every entry point returns CTD_OK (0) or a negative error code, never throws;
 * `ctd_last_error()` returns a thread-local human readable message.
 * Calls are asynchronous on the `stream` argument (a hipStream_t passed as
 * void*; NULL = the null stream) unless stated otherwise.
 */
#ifndef CTD_HIP_H
#define CTD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTD_ABI_VERSION 6

/* ---- error codes ------------------------------------------------------ */
#define CTD_OK 0
#define CTD_ERR_INVALID (-1)     /* bad argument / malformed program            */
#define CTD_ERR_HIP (-2)         /* a HIP runtime call failed                   */
#define CTD_ERR_UNSUPPORTED (-3) /* op/shape outside what the engine implements */
#define CTD_ERR_NOMEM (-4)

/* ---- arithmetic modes -------------------------------------------------- */
#define CTD_PREC_F32 0 /* fp32 activations, exact fmaf chains (parity / config 2)   */
#define CTD_PREC_F16 1 /* fp16 activations+weights, fp32 accumulate on MFMA (config 3) */
#define CTD_PREC_F32S 2 /* fp32 activations and weights, every product computed on the fp16 MFMA from
                           split operands (x = hi + lo: 3 MFMAs per product, ~22 mantissa bits,
                           fp32 accumulate): the fp32 engine's results at several times its rate */

/* ---- activations (reference models/yolov5/common.py:36-44, basemodel.py) -- */
#define CTD_ACT_NONE 0
#define CTD_ACT_SILU 1
#define CTD_ACT_LEAKY 2 /* LeakyReLU(0.1) */
#define CTD_ACT_RELU 3
#define CTD_ACT_SIGMOID 4

/* ---- network input formats -------------------------------------------- */
#define CTD_IN_NCHW_F32 0 /* (B,3,H,W) float in [0,1]: what preprocess_img hands the net
                             (reference inference.py:72-83)                           */
#define CTD_IN_NHWC_U8 1  /* (B,H,W,3) uint8 letterboxed page, channel order as the net
                             sees it (BGR, reference SURVEY App.C-1); /255 is fused    */

/* ---- op kinds of the lowered program ----------------------------------- */
#define CTD_OP_INPUT 1     /* network input -> 3-channel NHWC activation tensor          */
#define CTD_OP_CONV 2      /* conv2d k x k / stride / pad, <=2 concatenated sources with
                              optional nearest x2 upsample each, bias, act, residual     */
#define CTD_OP_CONVT 3     /* ConvTranspose2d k x k / stride / pad, bias, act            */
#define CTD_OP_MAXPOOL 4   /* k x k, stride 1, pad k/2 (SPPF, reference common.py:188)   */
#define CTD_OP_AVGPOOL2 5  /* 2x2 stride 2 (reference basemodel.py:38)                   */
#define CTD_OP_DETECT 6    /* YOLO Detect decode of one level (reference yolo.py:23-44)  */
#define CTD_OP_EXPORT 7    /* 1-channel activation -> plane of an f32 NCHW output,
                              optional u8 side output                                    */
#define CTD_OP_STEM 8      /* fused CTD_OP_INPUT + 6x6/s2/p2 conv (3 -> cout)            */
#define CTD_OP_SEG_FINAL 9 /* fused ConvT 4x4/s2/p1 (cin -> 1) + sigmoid + exports       */
#define CTD_OP_DB_UP 10    /* fused DB tail: per branch ConvT2x2(q->q)+ReLU, ConvT2x2(q->1),
                              sigmoid, exports (reference basemodel.py:99-102,138-142)   */

/* ---- external outputs (selected by ctd_op.aux[0] of EXPORT-like ops) ---- */
#define CTD_OUT_MASK 0    /* mask      (B,1,H,W) f32 + mask_u8 (B,H,W) = (uint8)(p*255)
                             (reference inference.py:85-99 postprocess_mask)             */
#define CTD_OUT_LINES 1   /* lines_map (B,2,H,W) f32 + bitmap (B,H,W) = plane0 > thresh
                             (reference utils/db_utils.py:71-72 binarize)                */

/* A tensor of the program: NHWC activation with `channels` channels at
 * spatial size (H >> log2_down, W >> log2_down). */
typedef struct ctd_tensor {
  int32_t channels;
  int32_t log2_down;
  int32_t dtype; /* 0 = the engine's activation type (f32 / f16), 1 = always f32
                    (used for the raw Detect logits so the box decode stays fp32) */
} ctd_tensor;

/* One op.  Sources are (tensor id, channel offset, channel count, upsample
 * flag); the op reads the channel-concatenation [src0 | src1]. */
typedef struct ctd_op {
  int32_t kind;
  int32_t src0, src0_coff, src0_c, src0_up;
  int32_t src1, src1_coff, src1_c, src1_up; /* src1 = -1: unused */
  int32_t res, res_coff;                    /* residual added AFTER act (reference common.py:104); -1: none */
  int32_t dst, dst_coff;
  int32_t cout;
  int32_t k, stride, pad;
  int32_t act;
  int64_t w_off; /* element offset of the weights in the f32 parameter blob:
                    CONV : (cout, cin, k, k)   [torch Conv2d layout, BN folded]
                    CONVT: (cin, cout, k, k)   [torch ConvTranspose2d layout, BN folded] */
  int64_t b_off; /* element offset of the bias (cout floats), or -1 */
  int32_t aux[8];
  /* DETECT   : aux[0]=level stride, aux[1]=row offset into blks PER 64x64 INPUT UNIT
                (scaled by (H/64)*(W/64) at run time), aux[2]=na, aux[3]=no
     EXPORT   : aux[0]=CTD_OUT_*, aux[1]=plane index, aux[2]=planes of that output (0 = the default: 2 for
                CTD_OUT_LINES) -- a program lowered without the DB threshold branch has ONE plane
     SEG_FINAL: aux[0]=CTD_OUT_MASK
     DB_UP    : aux[0]=CTD_OUT_LINES, aux[1]=q (branch channels), aux[2]=branches lowered (0 or 2: binarize and
                thresh; 1: binarize only = the shrink map `SegDetectorRepresenter` reads); parameter
                layout at w_off: for branch in (binarize, thresh)[:aux[2]]:
                  W1 (q,q,2,2), b1 (q), W2 (q,1,2,2), b2 (1)                          */
  float faux[8];
  /* DETECT   : faux[0..2*na) = anchors in pixels (anchor * stride)
     EXPORT / DB_UP : faux[0] = binarisation threshold for the u8 bitmap               */
} ctd_op;

typedef struct ctd_engine ctd_engine;

/* ---- engine ------------------------------------------------------------ */

/* Builds an engine from a lowered program.  `params` is a host blob of
 * `n_params` floats holding every (BN-folded) weight and bias; the engine
 * repacks (and for CTD_PREC_F16 converts) them into its own device layouts, so
 * the caller may free the blob afterwards.  Synchronous. */
int ctd_engine_create(ctd_engine** out, const ctd_tensor* tensors, int32_t n_tensors,
                      const ctd_op* ops, int32_t n_ops, const float* params, int64_t n_params,
                      int32_t precision, int32_t device);

void ctd_engine_destroy(ctd_engine* e);

/* Row count of `blks` for an H x W input (sum over Detect levels of na*ny*nx),
 * and the per-row width `no` (5 + nc). */
int ctd_engine_blks_shape(const ctd_engine* e, int32_t H, int32_t W, int32_t* rows, int32_t* no);

/* The fused forward: replaces `TextDetBase.forward` (reference basemodel.py:240-244).
 *   input_dev : B pages in `input_fmt`
 *   blks_dev  : (B, rows, no) f32           (`blk`, reference yolo.py:44)
 *   mask_dev  : (B, 1, H, W) f32            (`seg`, reference basemodel.py:74); NULL: not written (the u8
 *               side output below is what the detector consumes)
 *   lines_dev : (B, P, H, W) f32            (`det`, reference basemodel.py:125); P = 2, or 1 for a program
 *               lowered without the threshold branch
 *   mask_u8_dev / bitmap_dev : (B, H, W) u8 fused post-processing side outputs
 *               (reference inference.py:96-99 / utils/db_utils.py:71-72); may be NULL.
 * H and W must be multiples of 64 (reference SURVEY section 5).  Workspace is
 * (re)planned internally when (B,H,W) changes (synchronous in that case). */
int ctd_engine_forward(ctd_engine* e, const void* input_dev, int32_t input_fmt, int32_t B, int32_t H,
                       int32_t W, float* blks_dev, float* mask_dev, float* lines_dev,
                       uint8_t* mask_u8_dev, uint8_t* bitmap_dev, void* stream);

/* Introspection for tests / bench: per-op algorithmic work for the last
 * planned (B,H,W).  Arrays have ctd_engine_n_ops() entries. */
int32_t ctd_engine_n_ops(const ctd_engine* e);
int ctd_engine_op_work(const ctd_engine* e, double* flops, double* bytes, int32_t* kernel_class);
/* kernel_class: 0 = pointwise/pool/export, 1 = MFMA implicit-GEMM conv, 2 = MFMA convT,
 *               3 = direct (VALU) conv, 4 = fused stem / seg-final / db-up */

/* ABI v5.  Name of the kernel op `op_index` launches under the CURRENT plan and tuning ("conv_halo3_kernel",
 * "conv_igemm_kernel", "c3b_kernel", ...; "(fused)" for an op whose work another op's launch does): lets a test assert
 * WHICH dispatch a parity comparison ran through (grid thresholds pick different kernels at B = 1 and B = 32).  `name` gets
 * at most cap - 1 characters and a terminator.  Valid after a forward / profile of the shape in question. */
int ctd_engine_op_kernel(const ctd_engine* e, int32_t op_index, char* name, int32_t cap);

/* Runs one forward with a hipEvent pair around every op on `stream` and
 * returns the per-op milliseconds (synchronous). */
int ctd_engine_profile(ctd_engine* e, const void* input_dev, int32_t input_fmt, int32_t B, int32_t H,
                       int32_t W, float* blks_dev, float* mask_dev, float* lines_dev,
                       uint8_t* mask_u8_dev, uint8_t* bitmap_dev, void* stream, float* op_ms);

/* Copies activation tensor `tensor_id` of the last forward to host as f32
 * NHWC (debug / per-layer parity tests).  Synchronous. */
int ctd_engine_read_tensor(ctd_engine* e, int32_t tensor_id, float* host_out, int64_t n_floats);

/* Bytes of HBM held by the activation arena for the current plan. */
int64_t ctd_engine_workspace_bytes(const ctd_engine* e);
/* Counter that changes whenever the arena is reallocated (a forward with a (B,H,W) that needs more than the
 * arena holds).  A hipGraph captured from ctd_engine_forward bakes the arena's addresses in: it may only be
 * replayed while this value equals the one read at capture time.  Re-planning for a shape that fits does NOT
 * change it (tensor offsets differ per shape, the allocation stays), so graphs of several shapes can coexist
 * when the largest shape was run first.  A forward that would have to grow the arena inside a stream capture
 * fails with CTD_ERR_INVALID. */
int32_t ctd_engine_arena_generation(const ctd_engine* e);

/* Kernel-dispatch knobs (process-wide; no reference counterpart).  Keys: "halo_min_patches" (default 1024:
 * maps with fewer 16x16 patches take the implicit-GEMM kernel), "halo" (0: never the halo kernel), "halo_pair",
 * "halo_1x1"; "fuse" = bit mask of the fp16 engine's multi-layer kernels (1: C3 block with 32 hidden channels,
 * 2: SPPF's three pools, 4: stem + layer 1; default 7; 0 = one launch per layer; results are bit-identical either
 * way), "c3_min_patches" (default 1024: smaller grids take the per-layer kernels); "db_up_mfma" / "seg_final_mfma"
 * (default 1: the DB tail / the seg-final layer with their channel reductions on the MFMA, 0: the VALU kernels; same
 * results within 2e-4 / 1e-6).  The environment variables CTD_HALO_* / CTD_FUSE / CTD_DBUP_MFMA / CTD_SEGFINAL_MFMA give
 * the initial values.  Engines re-plan on their next forward.  For tests and A/B
 * measurements. */
int ctd_tuning_set(const char* key, int64_t value);

/* ---- post-processing kernels ------------------------------------------- */

/* Class-aware greedy NMS on the decoded Detect rows; replaces
 * `non_max_suppression` (reference utils/yolov5_utils.py:124-218, incl. the
 * torchvision.ops.nms call at :202) for multi_label=False, agnostic=False.
 *   blks_dev   : (B, rows, no) f32 [cx,cy,w,h,obj,cls...]
 *   dets_dev   : (B, max_det, 6) f32 [x1,y1,x2,y2,conf,cls], score-descending
 *   counts_dev : (B) i32 number of valid rows in dets
 * `ws_dev` scratch of at least ctd_nms_workspace_bytes(B, rows) bytes. */
size_t ctd_nms_workspace_bytes(int32_t B, int32_t rows);
int ctd_nms(const float* blks_dev, int32_t B, int32_t rows, int32_t no, float conf_thres,
            float iou_thres, int32_t max_det, int32_t max_nms, float max_wh, float* dets_dev,
            int32_t* counts_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Connected-component labelling with statistics of (img > thresh) on a batch
 * of u8 images; replaces `cv2.connectedComponentsWithStats` (reference
 * utils/textmask.py:93,113,138).  connectivity 4 or 8.
 *   labels_dev : (B, H, W) i32; 0 = background, components numbered 1..n in
 *                raster order of their first pixel
 *   n_dev      : (B) i32 number of components (excluding background)
 *   stats_dev  : (B, max_labels, 5) i32 [x, y, w, h, area] for labels 1..n at
 *                row label-1 (rows beyond max_labels are dropped, n still counts them;
 *                rows of labels that do not exist are not written) */
size_t ctd_ccl_workspace_bytes(int32_t B, int32_t H, int32_t W);
int ctd_ccl(const uint8_t* img_dev, int32_t B, int32_t H, int32_t W, int32_t thresh,
            int32_t connectivity, int32_t* labels_dev, int32_t* n_dev, int32_t* stats_dev,
            int32_t max_labels, void* ws_dev, size_t ws_bytes, void* stream);

/* Both labellings `SegDetectorRepresenter.boxes_from_bitmap` needs (reference utils/db_utils.py:134-136:
 * `cv2.findContours(bitmap, RETR_LIST)` = one contour per 8-connected foreground component and per enclosed
 * 4-connected background region) in ONE union-find over the image:
 *   labels_dev : (B, H, W) i32 signed: +id = foreground component (img > thresh, 8-connected),
 *                -id = background region (4-connected); ids per class in raster order of the first pixel
 *   n_f / n_b  : (B) counts; stats_f / stats_b : (B, max_labels, 5) as ctd_ccl; first_f / first_b :
 *                (B, max_labels) linear index of each component's first pixel
 * Label for label what ctd_ccl(img, connectivity 8) and ctd_ccl(complement of img, connectivity 4) give. */
int ctd_ccl_dual(const uint8_t* img_dev, int32_t B, int32_t H, int32_t W, int32_t thresh, int32_t* labels_dev,
                 int32_t* n_f_dev, int32_t* n_b_dev, int32_t* stats_f_dev, int32_t* stats_b_dev, int32_t* first_f_dev,
                 int32_t* first_b_dev, int32_t max_labels, void* ws_dev, size_t ws_bytes, void* stream);

/* `DBHead.step_function` (reference basemodel.py:159-160), the differentiable binarisation
 * 1 / (1 + exp(-k (P - T))) of the shrink map P and the threshold map T = the two planes of `lines_map`
 * (B,2,H,W); out (B,1,H,W) f32 is what `DBHead.forward(step_eval=True)` returns (basemodel.py:121-122);
 * bitmap (B,H,W) u8 = out > thresh, or NULL.  k = 50 in the reference (basemodel.py:84). */
int ctd_db_step(const float* lines_dev, int32_t B, int32_t H, int32_t W, float k, float* out_dev, uint8_t* bitmap_dev,
                float thresh, void* stream);

/* ---- pre / post resampling ---------------------------------------------- */

/* cv2.resize(src, (dW,dH), INTER_LINEAR) for uint8 images with C = 1 or 3 interleaved
 * channels, OpenCV's fixed-point arithmetic, written into the top-left corner of a
 * (canvasH, canvasW, C) buffer whose remaining bottom/right area is zero filled.
 * With canvas > d this is the reference's `letterbox` (utils/imgproc_utils.py:86-117:
 * resize :113 + copyMakeBorder :116); with canvas == d it is the mask resize of
 * inference.py:165. */
int ctd_resize_linear_u8(const uint8_t* src_dev, int32_t sH, int32_t sW, int32_t C, uint8_t* dst_dev,
                         int32_t dH, int32_t dW, int32_t canvasH, int32_t canvasW, void* stream);

/* ---- the detector tail ------------------------------------------------------ */

/* Everything `TextDetector.__call__` does after `self.net(img_in)` (reference inference.py:148-178) for a
 * whole batch of pages, driven natively: NMS + `postprocess_yolo` (inference.py:101-114), the DB
 * text-line stage (`SegDetectorRepresenter`, utils/db_utils.py:32-211: two labelling passes and contour
 * tables on the GPU, hull / min-area rectangle / unclip on the host), the mask crop + resize
 * (inference.py:164-165), `group_output` (utils/textblock.py:421-508), `refine_mask` and
 * `refine_undetected_mask` (utils/textmask.py:135-169: histograms, xor distances, candidate masks,
 * labelling, merge rounds, dilation, hole filling and the final OR on the GPU; colour / threshold picks
 * on the host).  A `ctd_tail` owns a HIP stream and its buffers; different objects may run concurrently
 * from different host threads.  All calls are synchronous: results are complete on return. */
typedef struct ctd_tail ctd_tail;

typedef struct ctd_tail_page {
  const uint8_t* img_dev; /* the page as the caller passed it: BGR u8 (im_h, im_w, 3) on the device      */
  int32_t im_h, im_w;     /* page size                                                                    */
  int32_t dw, dh;         /* right / bottom letterbox padding of the network input (inference.py:143)     */
} ctd_tail_page;

typedef struct ctd_tail_params {
  float conf_thresh, nms_thresh; /* reference inference.py:121 defaults 0.4 / 0.35                        */
  float box_thresh;              /* DB line score threshold, 0.6 (inference.py:159)                       */
  int32_t max_candidates;        /* 1000 (utils/db_utils.py:33)                                           */
  double unclip_ratio;           /* 1.5                                                                   */
  int32_t refine;                /* 0: stop after group_output                                            */
  int32_t refine_mode;           /* REFINEMASK_INPAINT 0 / REFINEMASK_ANNOTATION 1 (utils/textmask.py:13) */
  int32_t keep_undetected_mask;  /* run refine_undetected_mask (inference.py:175-176)                     */
  int32_t pad_;
} ctd_tail_params;

int ctd_tail_create(ctd_tail** out, int32_t device);
void ctd_tail_destroy(ctd_tail* t);
void* ctd_tail_stream(ctd_tail* t); /* the hipStream_t the tail's kernels run on */

/* Network outputs of a batch (all on the device): blks (B,rows,no) f32, mask_u8 (B,Hn,Wn) u8 =
 * `postprocess_mask` (fused in the engine), prob = plane 0 of lines_map (page b at prob_dev +
 * b * prob_stride floats), bitmap (B,Hn,Wn) u8 = prob > 0.3.  mask_out[b] / refined_out[b]: host arrays of
 * im_h * im_w bytes (the `mask` and `mask_refined` the reference returns; entries or the arrays may be NULL).
 * ready_event: a hipEvent_t recorded after the network on its stream (waited for on the tail's stream), or
 * NULL if the outputs are already complete.  Blocks: ctd_tail_page_counts / ctd_tail_page_fetch. */
int ctd_tail_run(ctd_tail* t, int32_t B, int32_t Hn, int32_t Wn, const float* blks_dev, int32_t rows, int32_t no,
                 const uint8_t* mask_u8_dev, const float* prob_dev, int64_t prob_stride, const uint8_t* bitmap_dev,
                 const ctd_tail_page* pages, const ctd_tail_params* prm, uint8_t* const* mask_out,
                 uint8_t* const* refined_out, void* ready_event);

/* Host wall clock of the stages of the last ctd_tail_run in ms: [0] enqueue of NMS / labelling / contour tables /
 * page masks, [1] wait for them, [2] table download + contour geometry, [3] yolo unpack + group_output,
 * [4] refine: wait for the histograms, [5] refine: wait for the xor sums, [6] refine: host decisions + enqueue,
 * [7] refine_undetected_mask, [8] final wait + copies, [9] total, [10] the part of [2] spent waiting for the
 * table download, [11]-[13] parts of [0]: NMS + buffers, labelling + contour tables, page-mask copies, [14] the part
 * of [8] spent waiting for the refine stage's kernels, [15] refine: wait for the window-local merge kernel's overflow
 * flags.  ms must hold 16 doubles. */
int ctd_tail_timings(const ctd_tail* t, double* ms);

/* Which path the refine windows of the last ctd_tail_run / ctd_tail_refine took (reference utils/textmask.py:73-132,
 * merge_mask_list per window): counts3 = [windows merged by the window-local kernel (one block per window on bit planes in
 * LDS), windows merged through the packed canvases (too large for the LDS, or re-done after an overflow), run-table
 * overflows of the window-local kernel].  Same results on every path; ABI v6. */
int ctd_tail_refine_paths(const ctd_tail* t, int32_t* counts3);

/* The DB text-line stage alone (`SegDetectorRepresenter.__call__`, reference utils/db_utils.py:40-69): boxes and
 * scores of every contour of every page, read back with ctd_tail_page_counts / ctd_tail_page_fetch. */
int ctd_tail_db_boxes(ctd_tail* t, int32_t B, int32_t Hn, int32_t Wn, const float* prob_dev, int64_t prob_stride,
                      const uint8_t* bitmap_dev, int32_t max_candidates, double unclip_ratio);

/* `refine_mask` (+ `refine_undetected_mask`) alone for given pages, page-size masks (HOST) and block boxes
 * (blk_xyxy: all pages' (x1,y1,x2,y2) concatenated, blk_counts[b] per page).  mask_out receives the masks
 * as `refine_undetected_mask` edits them in place (may be NULL). */
int ctd_tail_refine(ctd_tail* t, int32_t n_pages, const ctd_tail_page* pages, const uint8_t* const* masks_host,
                    const int32_t* blk_xyxy, const int32_t* blk_counts, int32_t refine_mode, int32_t keep_undetected_mask,
                    uint8_t* const* mask_out, uint8_t* const* refined_out);

/* Results of the last ctd_tail_run for one page: grouped blocks (ctd_blk, below) with their line and distance
 * pools, every DB contour box (n,4,2) i16 + score as `SegDetectorRepresenter.__call__` returns them, and the
 * NMS blocks (xyxy i32, class, confidence) as `postprocess_yolo` returns them.  Any output may be NULL. */
int ctd_tail_page_counts(const ctd_tail* t, int32_t page, int32_t* n_blocks, int32_t* n_lines, int32_t* n_dist,
                         int32_t* n_db_boxes, int32_t* n_yolo);
struct ctd_blk;
int ctd_tail_page_fetch(const ctd_tail* t, int32_t page, struct ctd_blk* blocks, int32_t* lines, double* dist,
                        int16_t* db_boxes, float* db_scores, int32_t* yolo_xyxy, int32_t* yolo_cls, float* yolo_conf);

/* The same grouped blocks for the WHOLE batch of the last ctd_tail_run in two calls (the per-page pair above costs the
 * Python host side two foreign calls and three allocations per page, under the interpreter lock its tail workers share):
 * counts (B,5) i32 = per page n_blocks, n_lines, n_dist, n_db_boxes, n_yolo; then the pages' ctd_blk records, line
 * quads (n,8) i32 and distance triples (m,3) f64 back to back in page order (a block's line_off / dist_off stay
 * relative to ITS page's first line / triple).  Any output may be NULL.  Pure host code. */
int ctd_tail_batch_counts(const ctd_tail* t, int32_t* counts);
int ctd_tail_batch_fetch(const ctd_tail* t, struct ctd_blk* blocks, int32_t* lines, double* dist);

/* The grouped blocks of EVERY page of the last ctd_tail_run as fixed-capacity f64 records, the unit of the multi-GPU
 * record gather (comic-text-detector_amd/dist.py): per page
 *   [n_blocks, n_lines (true counts), cap_blk, cap_line,
 *    cap_blk x 12: x1, y1, x2, y2, language, vertical, angle, font_size, n_lines, norm, vec_x, vec_y,
 *    cap_line x 8: the line quads in block order]
 * truncated (counts stay true) when a page has more blocks / lines than the capacities.  `out` holds
 * B * (4 + 12 cap_blk + 8 cap_line) doubles, zero filled where unused.  Pure host code. */
int ctd_tail_pack_records(const ctd_tail* t, int32_t cap_blk, int32_t cap_line, double* out);

/* Host threads the per-page / per-window host loops of this tail object may use (default 8; >= 1).  A node running one
 * process per GPU divides its cores between the ranks' tail workers. */
int ctd_tail_set_threads(ctd_tail* t, int32_t n);

/* ---- host-side input staging ---------------------------------------------------------------- */

/* Copies `n` host buffers (the caller's page images, reference inference.py:141 `img`) back to back into
 * `dst` (page-locked memory the caller then uploads with ONE asynchronous copy), on up to `threads` host
 * threads.  Pure host code; called through an FFI it runs without the caller's interpreter lock. */
int ctd_host_gather(void* dst, const void* const* srcs, const size_t* sizes, int32_t n, int32_t threads);

/* ---- host-side contour geometry of the DB text-line stage -------------- */

/* `SegDetectorRepresenter.boxes_from_bitmap` (reference utils/db_utils.py:134-211) downstream of
 * the two labelling passes: all pointers are HOST memory, no device work (scalar O(#contours)
 * double arithmetic).  prob (H,W) f32 = shrink map; lab_f / st_f / n_f = labels,
 * [x,y,w,h,area] stats and count of `ctd_ccl(bitmap, connectivity 8)`; lab_b / st_b / n_b =
 * those of `ctd_ccl(1 - bitmap, connectivity 4)`.  The contour set of
 * cv2.findContours(RETR_LIST) (db_utils.py:142) = one outer border per foreground component + one
 * hole border per background component that does not touch the frame, newest first.  Per
 * contour: get_mini_boxes (:177-194), the min-side test (:146-147), box_score_fast (:196-211),
 * unclip (:168-174), second get_mini_boxes (:154) and the rescale/clip of :158-163 with dest
 * size == bitmap size.  Outputs: boxes (n,4,2) i16 and scores (n) f32 with n = min(#contours,
 * max_candidates) in *n_out; rejected contours keep all-zero rows, like the reference's
 * preallocated arrays (:143-144).  boxes / scores must hold max_candidates entries. */
int ctd_db_boxes(const float* prob, const int32_t* lab_f, const int32_t* st_f, int32_t n_f, const int32_t* lab_b,
                 const int32_t* st_b, int32_t n_b, int32_t W, int32_t H, int32_t max_candidates,
                 double unclip_ratio, int16_t* boxes, float* scores, int32_t* n_out);

/* The same stage (`boxes_from_bitmap`, reference utils/db_utils.py:123-211) from tables compacted on the
 * DEVICE, so that no label image or probability map is downloaded (csrc/kernels_tail.hip `launch_dbc`, driven
 * by `ctd_tail_run`).  HOST memory only.  Per polarity (f = 8-connected foreground components, b =
 * 4-connected background components; labels 1..n in first-pixel order, row l-1 of every table):
 *   st_*    (n,5) [x,y,w,h,area]            first_* (n) linear index of the component's first pixel
 *   par_f   (n_f) background label left of the first pixel (0 at the page edge)
 *   par_b   (n_b) foreground label left of the first pixel if the component is a HOLE (does not touch the
 *                 page frame), else 0
 *   off_f   (n_f) first entry of the component's h rows in row_lo / row_hi (leftmost / rightmost x per row)
 *   off_b   (n_b) first entry of the h + 2 rows of a hole's border ring (the ringing component's pixels that
 *                 4-touch the hole), rows y - 1 .. y + h
 *   sum_f / sum_b  sum of the probability map over the component / the hole
 *   ring_sum / ring_cnt  sum of the probability map over the border ring / its pixel count
 * Outputs as `ctd_db_boxes`. */
int ctd_db_boxes_compact(int32_t W, int32_t H, int32_t n_f, const int32_t* st_f, const int32_t* first_f,
                         const int32_t* par_f, const int32_t* off_f, const double* sum_f, int32_t n_b,
                         const int32_t* st_b, const int32_t* first_b, const int32_t* par_b, const int32_t* off_b,
                         const double* sum_b, const double* ring_sum, const int32_t* ring_cnt, const int32_t* row_lo,
                         const int32_t* row_hi, int32_t max_candidates, double unclip_ratio, int16_t* boxes,
                         float* scores, int32_t* n_out);

/* ---- host-side block / line grouping ------------------------------------- */

/* One grouped text block: the detection fields of the reference's `TextBlock` record (reference
 * utils/textblock.py:45-56).  Lines and distances live in pools: lines (n,4,2) i32 at line_off;
 * `TextBlock.distance` at dist_off as triples (distance, c, d) with distance = |sin(arccos(c)) * d|
 * (utils/textblock.py:327-328; c and d let the caller re-evaluate that expression with its own math
 * library).  n_dist is NOT always n_lines: `split_textblk` hands every part the whole array of the
 * block it came from (:395-396). */
typedef struct ctd_blk {
  int32_t xyxy[4];
  int32_t language;      /* 0 eng, 1 ja, 2 unknown (reference utils/textblock.py:9) */
  int32_t vertical;
  int32_t angle;
  int32_t font_is_float; /* Python type of font_size in the reference: int (0), float after a merge (1) */
  double font_size;
  double vec[2];
  double norm;
  double weight;
  int32_t merged;
  int32_t line_off, n_lines;
  int32_t dist_off, n_dist;
  int32_t pad_;
} ctd_blk;

/* `group_output` (reference utils/textblock.py:421-508, sort_blklist=True) for one page.  HOST memory
 * only, no device work.  blines (n_blk,4) i32 / cls (n_blk) i32 = `postprocess_yolo`'s blocks (reference
 * inference.py:101-114); lines (n_lines,4,2) i32 = the rescaled DB boxes (inference.py:166-172);
 * mask (im_h rows of mask_pitch bytes) = the page-size u8 mask, or NULL (mask=None).
 * Outputs: blocks in reading order; capacities blk_cap >= n_blk + n_lines, line_cap >= n_blk + n_lines,
 * dist_cap >= (n_blk + n_lines) * max(1, n_lines) are always enough (CTD_ERR_NOMEM otherwise). */
int ctd_group_output(const int32_t* blines, const int32_t* cls, int32_t n_blk, const int32_t* lines, int32_t n_lines,
                     int32_t im_w, int32_t im_h, const uint8_t* mask, int32_t mask_pitch, ctd_blk* blks_out,
                     int32_t blk_cap, int32_t* lines_out, int32_t line_cap, double* dist_out, int32_t dist_cap,
                     int32_t* n_blk_out, int32_t* n_lines_out, int32_t* n_dist_out);

/* ---- host decisions of the mask refinement (exposed for tests; HOST memory, no device work) ---- */

/* np.histogram(px, bins=255) + get_topk_color(k=3, color_var=10, bin_tol=0.001) (reference
 * utils/textmask.py:16-27,61-62) from the 256-bin histogram of the selected grey pixels; writes up to 3
 * colours (left bin edges, float64) and returns how many. */
int ctd_topk_colors(const int64_t* hist256, double* colors3);
/* Threshold picked by cv2.threshold(.., THRESH_OTSU) (reference utils/textmask.py:47) from a 256-bin histogram. */
int ctd_otsu_from_hist(const int64_t* hist256);
/* Integer bounds cv2.inRange(u8, lo, hi) derives from double scalars (reference utils/textmask.py:68);
 * lb > ub on return = the empty range. */
void ctd_inrange_bounds(double lo, double hi, int32_t* lb, int32_t* ub);

/* ---- misc -------------------------------------------------------------- */
const char* ctd_last_error(void);
int32_t ctd_abi_version(void);
/* Fills name (<=255 chars) and returns the gfx arch number (950 on MI355X), or <0. */
int ctd_device_info(int32_t device, char* name, int32_t* cu_count, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CTD_HIP_H */
