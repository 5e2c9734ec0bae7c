// Host-side launchers of every device kernel (definitions in the .hip files).
#pragma once
#include <vector>

#include "ctd_common.h"

// ---- kernels_basic.hip : direct (VALU) kernels, T = float | half_t -------
// weights for the direct conv: f32 [KH*KW][cin_total][N]
void launch_conv_direct(const ConvArgs& a, bool f16, hipStream_t st);
// generic transposed conv; weights f32 [k*k][cin][N]; a.stride = s, a.dy0 = pad, a.KH = k
void launch_convt_direct(const ConvArgs& a, bool f16, hipStream_t st);
// dst: NHWC with `pitch` >= 3 channels per pixel; channels 3.. are zero filled
void launch_input_nchw(const float* in, void* dst, int pitch, int B, int H, int W, bool f16, hipStream_t st);
void launch_input_u8(const uint8_t* in, void* dst, int pitch, int B, int H, int W, bool f16, hipStream_t st);
void launch_maxpool(const void* src, int pitchS, void* dst, int pitchD, int C, int B, int H, int W, int k,
                    bool f16, hipStream_t st);
// SPPF: the three chained stride-1 max pools of one cat tensor in one launch (esize 2: fp16 tensors, 4: fp32)
bool sppf_pool3_supported(int pitch, int slot, int C, int H, int W, int k, const void* cat, int esize = 2);
void launch_sppf_pool3(void* cat, int pitch, int slot, int C, int B, int H, int W, int k, hipStream_t st, int esize = 2);
void launch_avgpool2(const void* src, int pitchS, void* dst, int pitchD, int C, int B, int Ho, int Wo,
                     bool f16, hipStream_t st);
// raw: (B,ny,nx,pitch) with na*no used channels -> blks rows [row_off, row_off+na*ny*nx)
void launch_detect_decode(const void* raw, int pitch, bool raw_f16, float* blks, int rows_total, int row_off,
                          int B, int ny, int nx, int na, int no, float stride, const float* anchors_px,
                          hipStream_t st);
// 1-channel activation -> plane `plane` of an (B,nplanes,H,W) f32 tensor (+ optional u8)
// u8_mode: 0 none, 1 = (uint8)(v*255) (truncate), 2 = v > thresh
void launch_export_plane(const void* src, int pitch, bool f16, float* out, int nplanes, int plane, uint8_t* u8,
                         int u8_mode, float thresh, int B, int H, int W, hipStream_t st);

// DBHead.step_function on the two planes of lines_map (B,2,H,W) -> out (B,1,H,W) f32 (+ bitmap u8 = out > thresh)
void launch_db_step(const float* lines, float k, float* out, uint8_t* bitmap, float thresh, int B, int H, int W,
                    hipStream_t st);

// ---- kernels_igemm.hip : MFMA implicit-GEMM conv (fp16 in, fp32 acc) ------
// weights: half [nphase][Npad][K], K index = (ty*KW+tx)*(c0+c1) + c
extern int g_tail_max_blocks;   // kernels_post.hip: grid cap of the tail's big-grid kernels ("tail_max_blocks")
extern int g_igemm_occ_lo;
extern int g_igemm_force_bk;  // tuning knob: 0 = heuristic, 32 / 64 = forced K step
int igemm_pick_bk(int c0, int c1, int K, int N, int log2_down);
void igemm_pack_weights(const float* logical, int nphase, int N, int K, int bn, int bk, bool tiled,
                        std::vector<half_t>& out);
bool igemm_supported(const ConvArgs& a);
int igemm_ntile(int N);  // N tile the dispatcher will use (weights must be padded to it)
void launch_conv_igemm(const ConvArgs& a, bool dst_f32, hipStream_t st);   // dispatches to the halo kernel when it applies

// ---- kernels_f32.hip : f32-operand MFMA implicit-GEMM conv (the exact-fp32 engine) ----
// weights: f32 [nphase][Npad][K], K index = (ty*KW+tx)*(c0+c1) + c; bias f32 padded to Npad
int f32_mfma_ntile(int N);
bool conv_f32_mfma_supported(const ConvArgs& a);
void launch_conv_f32_mfma(const ConvArgs& a, hipStream_t st);

// ---- kernels_split.hip : split-operand (fp16 hi + lo, 3 MFMAs per product) conv on f32 tensors: the "fp32s" engine ----
#ifdef CTD_AB_VARIANTS
extern int g_split_wdma;   // selftest build: 0 = weight tiles through registers instead of LDS-DMA (ctd_tuning_set("split_wdma"))
extern int g_split_bm256;  // selftest build: 1 = 256-pixel blocks for 64-channel N tiles (ctd_tuning_set("split_bm256"))
#endif
bool conv_split_supported(const ConvArgs& a);
void launch_conv_split(const ConvArgs& a, hipStream_t st);   // dispatches to the halo kernel below when it applies
// ---- kernels_split_stem.hip : the fp32s engine's first layer straight from the network input (no INPUT launch) ----
bool stem_split_supported(const ConvArgs& a);
void launch_stem_split(const ConvArgs& a, const void* input, int in_fmt, hipStream_t st);
// ---- kernels_split_halo.hip : the same arithmetic on a 256-pixel haloed patch staged once per channel chunk (3x3 / ConvT) ----
extern int g_split_halo;                    // 0 disables ("split_halo")
#ifdef CTD_AB_VARIANTS
extern int g_split_halo_small;              // selftest build: 64-channel layers on 16x8 patches ("split_halo_small")
#endif
extern long long g_split_halo_min_patches;  // "split_halo_min_patches"
bool conv_split_halo_supported(const ConvArgs& a);
void launch_conv_split_halo(const ConvArgs& a, hipStream_t st);
// logical f32 [nphase][N][K] -> hi plane + lo plane (halves, [nphase][npad/32][K/32][32][32] each) + oscale[npad]
void split_pack_weights(const float* logical, int nphase, int N, int K, int npad, std::vector<half_t>& out,
                        std::vector<float>& oscale);

// ---- kernels_halo.hip : halo-tile MFMA conv (stride-1 3x3, ConvTranspose phases) ----
extern int g_conv_halo;   // 0 disables (selftest A/B)
bool conv_halo_supported(const ConvArgs& a, bool dst_f32);
int conv_tuning_set(const char* key, long long value);   // dispatch knobs of the MFMA conv kernels (kernels_halo.hip)
void launch_conv_halo(const ConvArgs& a, hipStream_t st);

// ---- kernels_halo2.hip : 256-pixel x 256-column (phase, channel) tiles for the ConvTranspose layers, one block per CU ----
extern long long g_halo2;              // 0 disables ("halo2")
extern long long g_halo2_min_blocks;   // "halo2_min_blocks"
bool conv_halo2_supported(const ConvArgs& a, bool dst_f32);
void launch_conv_halo2(const ConvArgs& a, hipStream_t st);
int halo2_tuning_set(const char* key, long long value);

// ---- kernels_halo3.hip : the same K loop on 256-pixel x 128-column tiles, four waves per block, two blocks per CU ----
extern long long g_halo3;              // 0 disables ("halo3")
extern long long g_halo3_min_blocks;   // "halo3_min_blocks"
bool conv_halo3_supported(const ConvArgs& a, bool dst_f32);
bool conv_halo3_post_supported(const ConvArgs& a);
bool conv_halo3_segp_supported(const ConvArgs& a);   // `a` carries post_w = seg-final taps, post_dst = P, post_n = -16   // `a` carries post_*: the layer + its single 1x1 consumer in one launch
void launch_conv_halo3(const ConvArgs& a, hipStream_t st);
int halo3_tuning_set(const char* key, long long value);

// ---- kernels_c3.hip : one-kernel C3 block (32 hidden channels, one bottleneck) -------------
// Weights / biases are the packed arrays of the four unfused ops (tile-major, 32-channel K step):
// w12 [Cin/32][64][32] (cv1 rows 0-31, cv2 rows 32-63), wm1 [32][32], wm2 [9 taps][32][32], wc3 [2][64][32].
struct C3Args {
  SrcView s0, s1;          // x = cat(s0, s1); s1.c == 0 when unused
  int B, H, W;
  const half_t *w12, *wm1, *wm2, *wc3;
  const float *b12, *bm1, *bm2, *bc3;
  void* dst;               // (B,H,W,pitchD) fp16, 64 channels written, already offset by the channel offset
  int pitchD;
  int act;
  const void* zeros;       // >= 16 B of zeros in HBM (source of out-of-image rows)
  int prio;                // as ConvArgs::prio
  half_t* dbg;             // selftest only: three (B,H,W,32) planes receiving y2, t, b of every patch pixel; null in the product
};
extern int g_fuse;         // fusion bit mask (CTD_FUSE / ctd_tuning_set("fuse")): 1 C3 block, 2 SPPF pools, 4 stem + model.1, 8 C3 bottleneck + cv3 (kernels_c3b.hip)
extern long long g_c3_min_patches;
bool c3_fused_supported(const C3Args& a);
void launch_c3_fused(const C3Args& a, hipStream_t st);

// ---- kernels_c3b.hip : bottleneck (m.cv1 1x1 + m.cv2 3x3 + shortcut) [+ cv3] of a C3 block with 64 / 128 hidden channels ----
// Weights / biases are the packed arrays of the unfused ops (tile-major, 32-channel K step, N tile = igemm_ntile):
// wm1 [ch/32][ch][32], wm2 [9 ch/32][ch][32] (K step = tap * ch/32 + chunk), wc3 [2ch/128][2ch/32][128][32].
struct C3bArgs {
  SrcView y1;              // the bottleneck's input (ch channels), read with a one-pixel halo
  SrcView y2;              // cv3 only: the C3's second branch (ch channels), cv3's K rows ch .. 2ch-1
  int B, H, W;
  int ch;                  // hidden channels: 64 or 128
  const half_t *wm1, *wm2, *wc3;
  const float *bm1, *bm2, *bc3;
  void* dst;               // (B,H,W,pitchD) fp16: 2 ch channels (cv3) or ch channels (the bottleneck's output), already offset
  int pitchD;
  int act;
  int add;                 // the bottleneck has a shortcut
  int cv3;                 // 1: cv3 follows in the same launch
  int tap_major;           // K walk of the 3x3: 1 = K-linear (kernels_igemm.hip), 0 = channel chunk outer (kernels_halo.hip)
  const void* zeros;       // >= 16 B of zeros in HBM (source of out-of-image rows)
  int prio;                // as ConvArgs::prio
};
extern long long g_c3b_min_patches;   // "c3b_min_patches"
extern int g_c3b_max_ch;              // "c3b_max_ch"
extern int g_c3b_cfg64, g_c3b_cfg128; // tiling variants per hidden width ("c3b_cfg64" 0 / 1 / 2, "c3b_cfg128" 0 / 1)
bool c3b_supported(const C3bArgs& a);
void launch_c3b(const C3bArgs& a, hipStream_t st);

// ---- kernels_stem2.hip : stem (6x6/s2, 3 -> 32) + layer 1 (3x3/s2, 32 -> 64) in one kernel -----------
struct Stem2Args {
  const void* in; int in_fmt;          // network input (CTD_IN_NCHW_F32 / CTD_IN_NHWC_U8), filled per launch
  int B, H, W;
  const half_t* wfrag; const float* bias0; int act0;    // stem: stem_pack_weights fragments
  const half_t* w1; const float* bias1; int act1;       // layer 1: implicit-GEMM packing [9 taps][64][32]
  half_t* dst; int pitchD;             // (B, H/4, W/4, pitchD), 64 channels written
  int prio;                            // as ConvArgs::prio
};
bool stem_conv2_supported(const Stem2Args& a);
void launch_stem_conv2(const Stem2Args& a, hipStream_t st);

// ---- kernels_fused.hip ----------------------------------------------------
// stem: 6x6 s2 p2, 3 -> N (N <= 32... multiple of 8), reads the network input directly.
// weights: MFMA A fragments from stem_pack_weights (two variants: float / uint8 input), bias f32 [N]
void launch_stem(const void* in, int in_fmt, half_t* dst, int pitchD, int B, int H, int W, int N,
                 const half_t* wfrag, const float* bias, int act, hipStream_t st);
void stem_pack_weights(const float* W /* (32, 3, 6, 6) */, std::vector<half_t>& out);
// seg final: ConvT 4x4 s2 p1 (C -> 1) + sigmoid; src (B,H,W,C) half; weights f32 [16][C] (ky*4+kx)
void launch_seg_final(const half_t* src, int pitch, int C, int B, int H, int W, const float* w, float bias,
                      float* mask, uint8_t* mask_u8, hipStream_t st);
// db tail: src (B,H,W, 2q) half [binarize q | thresh q] (already conv3x3+BN+ReLU)
// params f32 per branch: W1[q][q][2][2] (cin,cout,ky,kx) b1[q] W2[q][1][2][2] b2[1]
void launch_seg_final_f32(const float* src, int pitch, int C, int B, int H, int W, const float* w, float* mask,
                          uint8_t* mask_u8, hipStream_t st);
// the same layer from the per-tap products P (B,H,W,16) f32 its producer left (kernels_halo3.hip SEGP): col2im + sigmoid + u8
void launch_seg_final_gather(const float* P, int B, int H, int W, float bias, float* mask, uint8_t* mask_u8, hipStream_t st);
extern int g_seg_final_mfma;   // fp16 engine: seg-final's channel reduction on the MFMA (CTD_SEGFINAL_MFMA / "seg_final_mfma")
extern int g_db_up_mfma;   // fp16 engine: DB tail's first stage on the MFMA (CTD_DBUP_MFMA / ctd_tuning_set("db_up_mfma"))
void launch_db_up(const void* src, bool f32in, int pitch, int q, int nbr, int B, int H, int W, const float* params, float* lines,
                  uint8_t* bitmap, float thresh, hipStream_t st);

// ---- kernels_post.hip -------------------------------------------------------
size_t nms_workspace_bytes(int B, int rows);
void launch_nms(const float* blks, int B, int rows, int no, float conf, float iou, int max_det, int max_nms,
                float max_wh, float* dets, int* counts, void* ws, hipStream_t st);
size_t ccl_workspace_bytes(int B, int H, int W);
// invert: foreground = !(img > thresh); first (B,max_labels): linear index of every component's first pixel
// in raster order (= its union-find root), or null; bg_negative: background pixels of `labels` keep the -1 the union-find
// left there instead of being rewritten to 0 (callers that only test `label > 0`: 4 B less per background pixel)
void launch_ccl(const uint8_t* img, int B, int H, int W, int thresh, int conn, int* labels, int* n_out,
                int* stats, int max_labels, void* ws, hipStream_t st, int invert = 0, int* first = nullptr, int bg_negative = 0);
// Foreground (img > thresh, 8-connected) and background (4-connected) components in one union-find:
// labels (B,H,W) signed (+id / -id, per-class raster order), n / stats / first per class, same workspace.
void launch_ccl_dual(const uint8_t* img, int B, int H, int W, int thresh, int* labels, int* n_f, int* n_b, int* st_f,
                     int* st_b, int* first_f, int* first_b, int max_labels, void* ws, hipStream_t st);

// ---- kernels_pre.hip ----------------------------------------------------------
// cv2.resize(INTER_LINEAR) u8 (C = 1 or 3) into the top-left (dH,dW) of a zero-padded canvas
void launch_resize_linear_u8(const uint8_t* src, int sH, int sW, int C, uint8_t* dst, int dH, int dW, int canvasH,
                             int canvasW, hipStream_t st);

// ---- mfma layout probe (selftest) -------------------------------------------
void launch_mfma_probe(const half_t* a, const half_t* b, float* out, hipStream_t st);
