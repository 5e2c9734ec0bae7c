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

#define CTD_ABI_VERSION 1

/* ---- error codes ------------------------------------------------------ */
#define CTD_OK 0
#define CTD_ERR_INVALID (-1)     /* bad argument / malformed program            */
#define CTD_ERR_HIP (-2)         /* a HIP runtime call failed                   */
#define CTD_ERR_UNSUPPORTED (-3) /* op/shape outside what the engine implements */
#define CTD_ERR_NOMEM (-4)

/* ---- arithmetic modes -------------------------------------------------- */
#define CTD_PREC_F32 0 /* fp32 activations, exact fmaf chains (parity / config 2)   */
#define CTD_PREC_F16 1 /* fp16 activations+weights, fp32 accumulate on MFMA (config 3) */

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
     EXPORT   : aux[0]=CTD_OUT_*, aux[1]=plane index
     SEG_FINAL: aux[0]=CTD_OUT_MASK
     DB_UP    : aux[0]=CTD_OUT_LINES, aux[1]=q (branch channels); parameter
                layout at w_off: for branch in (binarize, thresh):
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
 *   mask_dev  : (B, 1, H, W) f32            (`seg`, reference basemodel.py:74)
 *   lines_dev : (B, 2, H, W) f32            (`det`, reference basemodel.py:125)
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
 *                row label-1 (rows beyond max_labels are dropped, n still counts them) */
size_t ctd_ccl_workspace_bytes(int32_t B, int32_t H, int32_t W);
int ctd_ccl(const uint8_t* img_dev, int32_t B, int32_t H, int32_t W, int32_t thresh,
            int32_t connectivity, int32_t* labels_dev, int32_t* n_dev, int32_t* stats_dev,
            int32_t max_labels, void* ws_dev, size_t ws_bytes, void* stream);

/* ---- pre / post resampling ---------------------------------------------- */

/* cv2.resize(src, (dW,dH), INTER_LINEAR) for uint8 images with C = 1 or 3 interleaved
 * channels, OpenCV's fixed-point arithmetic, written into the top-left corner of a
 * (canvasH, canvasW, C) buffer whose remaining bottom/right area is zero filled.
 * With canvas > d this is the reference's `letterbox` (utils/imgproc_utils.py:86-117:
 * resize :113 + copyMakeBorder :116); with canvas == d it is the mask resize of
 * inference.py:165. */
int ctd_resize_linear_u8(const uint8_t* src_dev, int32_t sH, int32_t sW, int32_t C, uint8_t* dst_dev,
                         int32_t dH, int32_t dW, int32_t canvasH, int32_t canvasW, void* stream);

/* ---- per-window kernels of the mask refinement (reference utils/textmask.py:29-71) ---- */

/* One text-block window of a page: both images live on the device. */
typedef struct ctd_window {
  const uint8_t* img;  /* page, BGR u8 interleaved, row = img_w * 3 bytes          */
  const uint8_t* mask; /* predicted mask of the page, u8, row = mask_w bytes        */
  int32_t img_w, mask_w;
  int32_t x1, y1, w, h; /* window [x1, x1+w) x [y1, y1+h) (reference textmask.py:162-164) */
} ctd_window;

/* A candidate-mask rule: kind -1 unused; 0 = cv2.inRange(grey, lo, hi);
 * 1/2/3 = threshold(channel B/G/R, lo, 255, THRESH_BINARY).  `invert` selects the negative
 * (255 - mask), `aux` = window index (render only). */
typedef struct ctd_rule {
  int32_t kind;
  float lo, hi;
  int32_t invert;
  int32_t aux;
} ctd_rule;

/* hist_dev (n,4,256) u32: [0] grey (BGR2GRAY) of the pixels whose 3x3-eroded mask > 127
 * (textmask.py:58-61), [1..3] B, G, R of the whole window (Otsu input, textmask.py:44-47).
 * `wins` is a HOST array (copied internally). */
int ctd_win_hist(const ctd_window* wins, int32_t n, uint32_t* hist_dev, void* stream);

/* sums_dev (n,nrules) u64: sum over the window of (rule(pixel) ? 255 - m : m), the xor distance
 * of `minxor_thresh` (textmask.py:36-37); nrules <= 6; rules is a HOST array (n*nrules). */
int ctd_win_xor(const ctd_window* wins, int32_t n, const ctd_rule* rules, int32_t nrules, uint64_t* sums_dev,
                void* stream);

/* Renders nbands candidate masks {0,255} into a (rows, canvas_w) u8 canvas: band k uses window
 * bands[k].aux with rule bands[k] and starts at row tops[k], left aligned.  The canvas feeds
 * ctd_ccl (reference textmask.py:93).  bands / tops are HOST arrays. */
int ctd_win_render(const ctd_window* wins, int32_t n, const ctd_rule* bands, const int32_t* tops, int32_t nbands,
                   uint8_t* canvas_dev, int32_t canvas_w, void* stream);

/* ---- merge stage of the mask refinement (reference utils/textmask.py:74-131) ---- */

/* A band of a labelled canvas: window `win`, first row `top` in the label canvas, first row `mtop`
 * in the merged-mask canvas (one band per window there). */
typedef struct ctd_band {
  int32_t win, top, mtop;
} ctd_band;

/* One accept round of `merge_mask_list` (textmask.py:93-107, and the hole-filling pass :113-131) for
 * the given bands: a labelled component is OR-ed into the merged mask iff it is allowed and, among
 * its pixels not merged yet, more lie on pred_bin = 255 than on pred_bin = 0 (pred_bin = 3x3 cross
 * erosion of the window's mask > 60, :85-89) -- the reference's `xor_merged < xor_origin` test.
 * Allowed: allowed_dev[label-1] != 0 when given, else bbox w*h >= min_box from stats_dev (:98-99).
 * labels_dev (rows, canvas_w) i32 and stats_dev (nlab,5) i32 are `ctd_ccl` outputs; counters_dev
 * = 2*(nlab+1) u32, zeroed by the caller before the first round of a canvas (labels are unique per
 * band, so rounds do not collide).  wins / bands are HOST arrays. */
int ctd_win_accept(const ctd_window* wins, int32_t n, const ctd_band* bands, int32_t nbands,
                   const int32_t* labels_dev, int32_t canvas_w, const int32_t* stats_dev,
                   const uint8_t* allowed_dev, int32_t min_box, uint8_t* merged_dev, int32_t merged_w,
                   uint32_t* counters_dev, void* stream);

/* merged_out = 3x3 rect dilation of merged_in inside each window (`dilate` != 0, REFINEMASK_INPAINT,
 * textmask.py:110-111) or a copy; comp_dev = 255 - merged_out (the canvas of the hole-filling
 * labelling, :113); count255_dev[win] += #pixels == 255 (the caller zeroes it).  mtops (n) HOST. */
int ctd_win_dilate(const ctd_window* wins, int32_t n, const int32_t* mtops, const uint8_t* merged_in_dev,
                   uint8_t* merged_out_dev, uint8_t* comp_dev, int32_t merged_w, uint32_t* count255_dev,
                   int32_t dilate, void* stream);

/* page[y1:y1+h, x1:x1+w] |= merged band of every window (textmask.py:167); page_dev must be 4-byte
 * aligned and its allocation a multiple of 4 bytes (word-wide atomic OR: windows may overlap; the
 * word holding the last pixels may reach up to 3 bytes past H*W). */
int ctd_win_commit(const ctd_window* wins, int32_t n, const int32_t* mtops, const uint8_t* merged_dev,
                   int32_t merged_w, uint8_t* page_dev, int32_t page_w, void* stream);

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

/* ---- misc -------------------------------------------------------------- */
const char* ctd_last_error(void);
int32_t ctd_abi_version(void);
/* Fills name (<=255 chars) and returns the gfx arch number (950 on MI355X), or <0. */
int ctd_device_info(int32_t device, char* name, int32_t* cu_count, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* CTD_HIP_H */
