// Device tables and launchers of the detector tail (csrc/tail.hip drives them, csrc/kernels_tail.hip
// implements them): everything after the network of reference inference.py:148-178 that is O(pixels).
#pragma once
#include "ctd_common.h"

extern int g_tail_max_blocks;   // kernels_post.hip: grid cap of the tail's big-grid kernels ("tail_max_blocks")

// ---- DB text-line stage: per-contour tables compacted on the device ---------------------------------
// Inputs: the dual labelling of a page batch (8-connected foreground of the bitmap, 4-connected
// background, one signed label image) with per-component stats and first pixels.  Outputs per page, fixed capacities `cap`
// (components per polarity) and `rcap` (row-table entries):
struct DbcTables {
  int B, H, W, cap, rcap;
  const float* prob;          // (B) planes of H*W f32, plane b at prob + b * prob_stride
  long long prob_stride;
  const int* lab;             // (B,H,W) signed labels of launch_ccl_dual: +id foreground component, -id background region
  const int* n_f;             // (B)
  const int* n_b;             // (B)
  const int* st_f;            // (B,cap,5) [x,y,w,h,area]
  const int* st_b;
  const int* first_f;         // (B,cap) linear index of the component's first pixel
  const int* first_b;
  int* par_f;                 // (B,cap) background label left of the first pixel (0: page edge)
  int* par_b;                 // (B,cap) foreground label left of the first pixel of a HOLE (0: not a hole)
  int* off_f;                 // (B,cap) first row-table entry of the component (h entries)
  int* off_b;                 // (B,cap) first row-table entry of the hole's border ring (h + 2 entries)
  int* hdr;                   // (B,4) [min(n_f,cap), min(n_b,cap), rows used, overflow flag]
  int* row_lo;                // (B,rcap) leftmost x of the component / ring in that row
  int* row_hi;                // (B,rcap) rightmost x
  double* sum_f;              // (B,cap) sum of prob over the component
  double* sum_b;              // (B,cap) sum of prob over the hole
  double* ring_sum;           // (B,cap) sum of prob over the hole's border ring
  int* ring_cnt;              // (B,cap) pixels of that ring
};
void launch_dbc(const DbcTables& t, hipStream_t st);

// ---- mask refinement: windows, candidate bands, packed canvases ---------------------------------------
struct TWin {
  const uint8_t* img;   // page, BGR u8 interleaved
  const uint8_t* mask;  // page-size predicted mask u8
  uint8_t* out;         // page-size refined mask (commit target)
  int img_w, mask_w, out_w;   // row pitches in pixels
  int x1, y1, w, h;     // window in the page
  int mx, my;           // top-left of the window's band in the merged canvas
  int band0, nband;     // this window's candidate bands: bands[band0 .. band0 + nband), in merge order (round = index)
};
struct TRule {          // candidate rule of a window
  int kind;             // -1 unused; 0 grey range [lo, hi]; 1..3 channel B/G/R > lo
  int lo, hi;
};
struct TBand {          // one rendered candidate in the label canvas
  int win;
  int cx, cy;           // top-left in the label canvas
  int kind, lo, hi, invert;
  int round;            // merge order inside the window (0 = closest candidate first)
};

void launch_tw_hist(const TWin* wins, int n, int max_pix, unsigned* hist, hipStream_t st);
void launch_tw_xor(const TWin* wins, const TRule* rules, int n, int max_pix, unsigned long long* sums, hipStream_t st);
void launch_tw_render(const TWin* wins, const TBand* bands, int nbands, int max_pix, uint8_t* canvas, int canvas_w,
                      hipStream_t st);
// one merge round: bands with .round == round (round < 0: every band), allowed = bbox w*h >= min_box
void launch_tw_accept(const TWin* wins, const TBand* bands, int nbands, int max_pix, int round, const int* labels,
                      int canvas_w, const int* stats, int max_labels, int min_box, uint8_t* merged, int merged_w,
                      unsigned* counters, hipStream_t st);
// ALL merge rounds of every window in one launch: a block owns a window and walks its bands in merge order (count,
// decide, apply, next band) -- the rounds of a window depend on each other, windows do not
// zeroes the per-label counter pairs of the labels a labelling launch produced (their count is on the device)
void launch_label_counters_zero(unsigned* counters, const int* n_labels, int cap, hipStream_t st);
void launch_tw_accept_all(const TWin* wins, const TBand* bands, int n, const int* labels, int canvas_w, const int* stats,
                          int max_labels, int min_box, uint8_t* merged, int merged_w, unsigned* counters, hipStream_t st);
void launch_tw_dilate(const TWin* wins, int n, int max_pix, const uint8_t* in, uint8_t* out, uint8_t* comp, int merged_w,
                      unsigned* count255, int dilate, hipStream_t st);
// hole filling (reference utils/textmask.py:113-131) on the labelled complement canvas: per-window
// area threshold = second largest of {set-pixel count, component areas} (top2: (n,3) [max, #max, runner-up]),
// then the accept round restricted to components with area < threshold
void launch_tw_holes(const TWin* wins, int n, int max_pix, const int* labels2, const int* stats2, const int* first2,
                     int max_labels, const unsigned* count255, int* top2, uint8_t* merged, int merged_w,
                     unsigned* counters2, hipStream_t st);
// the four passes of the hole filling as one launch, a block per window
void launch_tw_holes_all(const TWin* wins, int n, const int* labels2, const int* stats2, const int* first2, int max_labels,
                         const unsigned* count255, uint8_t* merged, int merged_w, unsigned* counters2, hipStream_t st);
void launch_tw_commit(const TWin* wins, int n, int max_pix, const uint8_t* merged, int merged_w, hipStream_t st);
// ---- the merge stage of a window (render -> merge rounds -> dilation -> hole filling -> commit) as ONE block on bit planes
// in LDS (kernels_twlds.hip): windows order[0 .. n) of `wins`, planes of at most `max_words` 32-pixel words, `rcap` runs per
// labelling; ovf[window] = 1 where a labelling had more runs (nothing committed for that window)
size_t tw_lds_bytes(int max_words, int rcap);
int tw_lds_rcap(int max_words);
// (false: the launch could not get its LDS -- nothing was enqueued)
bool launch_tw_lds(const TWin* wins, const TBand* bands, const int* order, int n, int max_words, int rcap, int dilate, int* ovf,
                   hipStream_t st);
// mask[p] = 0 where refined[p] > thr (reference utils/textmask.py:136)
void launch_mask_clear_where(uint8_t* mask, const uint8_t* refined, long long n, int thr, hipStream_t st);
// dst (rows x cols, pitch dpitch) = src (pitch spitch): the crop of inference.py:164
void launch_copy2d_u8(const uint8_t* src, int spitch, uint8_t* dst, int dpitch, int rows, int cols, hipStream_t st);

// ---- many copies / fills in one launch (kernels_tail.hip) ------------------------------------------------------------
struct MSeg {
  void* dst;
  const void* src;                 // null: fill with the byte `fill`
  unsigned long long row_bytes;    // bytes per row (the whole segment when rows == 1)
  long long dpitch, spitch;        // bytes between rows
  int rows;
  int fill;
  int vec;                         // set by the launcher: 16 / 4 / 1 bytes per element
  int pad_;
};
constexpr int kMSegMax = 24;
struct MSegs {
  int n = 0;
  int blk_off[kMSegMax + 1];
  MSeg s[kMSegMax];
};
// launches one kernel for the segments of `m` (no-op when empty) and empties it; dst / src: device memory or
// device-accessible (hipHostMalloc / hipHostRegister) host memory
void launch_multi_copy(MSegs& m, hipStream_t st);
