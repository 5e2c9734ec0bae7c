// Fused C3 block for gfx950: the whole `C3.forward` of a 32-hidden-channel, one-bottleneck C3
// (reference models/yolov5/common.py:125-138 with Bottleneck :94-106) in ONE kernel:
//
//     y1 = act(cv1 x)   y2 = act(cv2 x)          1x1, Cin -> 32 each (merged weight matrix, graph.py)
//     t  = act(m.cv1 y1)                         1x1, 32 -> 32
//     b  = y1 + act(m.cv2 t)                     3x3, 32 -> 32, zero padding, shortcut
//     out = act(cv3 [b ; y2])                    1x1, 64 -> 64
//
// Why: as four launches the block moves 14 channel rows per pixel through HBM (x in, y out / in three times, t out / in,
// residual in, cat in, out) for ~340 flop per byte; `model.2` (256x256x64 maps at 1024x1024, the only C3 of this network
// with 32 hidden channels and its most byte-bound chain) ran at 45 % of what its bytes allow: 0.437 ms per 32 pages.
// Here a block owns a 16x8 pixel patch, reads the 18x10 haloed patch of x once, keeps y1 / y2 / t / b in LDS and writes
// `out` once: 2 (+ halo overlap, served by L2) channel rows per pixel; 0.321 ms (latency-bound now).
//
// Arithmetic is the unfused path's, step for step (fp16 operands, fp32 MFMA accumulation over the same K order,
// every intermediate rounded to fp16 where the unfused kernels store it, shortcut added to the ROUNDED conv
// output), so the result is bit-identical to the four launches -- which is how the selftest and the GPU tests check it.
//
// Structure: 256 threads = 4 waves, 2 blocks per CU (65 KB of LDS each).  MFMA convention of kernels_igemm.hip:
// weights = A operand (rows = output channels), pixels = B operand (columns), v_mfma_f32_32x32x16_f16; a lane
// ends with 4 consecutive channels of one pixel.  LDS rows are 32 channels (64 B) with the 16-B chunks XOR
// swizzled by (row >> 2) & 3; everything arrives by LDS-DMA with the swizzle on the source address.
//
//   stage  pixels (fragments of 32)                         K            weights (packed as the unfused ops use them)
//   S1     y1: 180 haloed rows, padded to 192 (6)           Cin          w12 [Cin/32][64][32]
//          y2: the 128 patch pixels (4)
//   S2     t : 192 haloed rows (6), zero outside the image  32           wm1 [32][32]
//   S3     b : 128 patch pixels (4), 9 taps                 9 x 32       wm2 [9][32][32]
//   S4     out: 128 patch pixels (4) x 2 N fragments        64           wc3 [2][64][32]
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace {

constexpr int C3_TW = 16, C3_TH = 8;                 // pixel patch
constexpr int C3_HW = C3_TW + 2, C3_HH = C3_TH + 2;  // haloed patch 18 x 10
constexpr int C3_HROWS = C3_HW * C3_HH;              // 180
constexpr int C3_NROW = 192;                         // padded to 6 MFMA pixel fragments
constexpr int C3_PX = C3_TW * C3_TH;                 // 128
// LDS map (halves)
constexpr int C3_XR = 0;                             // S1: x chunks [2][192][32]; then wm2 [9][32][32] + wm1 [32][32]; then out [128][72]
constexpr int C3_XBUF = C3_NROW * 32;                // 6144
constexpr int C3_WR = 2 * C3_XBUF;                   // 12288: S1: w12 ring [2][64][32]; then wc3 [2][64][32]
constexpr int C3_WBUF = 64 * 32;                     // 2048
constexpr int C3_Y1 = C3_WR + 2 * C3_WBUF;           // 16384: y1 [192][32]; its patch rows become b
constexpr int C3_T = C3_Y1 + C3_NROW * 32;           // 22528: t [192][32]
constexpr int C3_Y2 = C3_T + C3_NROW * 32;           // 28672: y2 [128][32]
constexpr int C3_LDS = C3_Y2 + C3_PX * 32;           // 32768 halves = 64 KB
constexpr int C3_OP = 72;                            // pitch of the staged output tile (64 channels + 16 B)
constexpr int C3_WM1 = 9 * 32 * 32;                  // offset of wm1 behind the nine tap tiles (in C3_XR)
static_assert(C3_PX * C3_OP <= 2 * C3_XBUF, "output tile fits the x buffers");
static_assert(C3_WM1 + 32 * 32 <= 2 * C3_XBUF, "tap tiles fit the x buffers");

// DBG: selftest-only instantiation that also dumps the block's intermediates (y2, t, b on the patch pixels) to a.dbg
template <int ACT, bool DBG = false>
__global__ __launch_bounds__(256, 2) void c3_fused_kernel(C3Args a) {
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  __shared__ __attribute__((aligned(16))) half_t lds[C3_LDS + 2 * 192];
  float* bias_s = (float*)(lds + C3_LDS);   // [0,64) cv1|cv2, [64,96) m.cv1, [96,128) m.cv2, [128,192) cv3

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- block -> (page, patch); XCD-aware: each XCD gets a contiguous run of patches (shared halos share an L2)
  const int tilesX = (a.W + C3_TW - 1) / C3_TW, tilesY = (a.H + C3_TH - 1) / C3_TH;
  const int nblk = tilesX * tilesY * a.B;
  int v = blockIdx.x;
  {
    const int xcd = v & 7, within = v >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tpx = v % tilesX;
  v /= tilesX;
  const int tpy = v % tilesY;
  const int b = v / tilesY;
  const int y0 = tpy * C3_TH, x0 = tpx * C3_TW;

  if (t < 192) bias_s[t] = t < 64 ? a.b12[t] : t < 96 ? a.bm1[t - 64] : t < 128 ? a.bm2[t - 96] : a.bc3[t - 128];

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  auto swz = [](int row) { return (row >> 2) & 3; };
  // LDS-DMA: the 64 lanes of a wave write 64 consecutive 16-B chunks from the wave-uniform base `dst`
  auto dma = [&](const void* g, half_t* dst) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)dst, 16, 0, 0);
  };
  const int pos = t & 3;   // chunk position this thread's DMA lands on (row = q >> 2 of the pass)

  // ---- the haloed patch of x: 192 rows x 4 chunks = three passes of the block -------------------------------
  auto inside = [&](int r) {   // haloed row r (18 per patch row) lies inside the image
    const int hy = r / C3_HW, hx = r - hy * C3_HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    return r < C3_HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
  };
  int aoff0[3], aoff1[3];
  bool aok[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = (i * 256 + t) >> 2;
    const int hy = r / C3_HW, hx = r - hy * C3_HW;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    aok[i] = inside(r);
    const int gs = (pos ^ swz(r)) * 8;
    // tensors are < 2 GiB (engine plan guard): byte offsets fit 32 bits
    aoff0[i] = aok[i] ? (((b * a.s0.H + iy) * a.s0.W + ix) * a.s0.pitch + gs) * 2 : 0;
    aoff1[i] = aok[i] && a.s1.c ? (((b * a.s1.H + iy) * a.s1.W + ix) * a.s1.pitch + gs) * 2 : 0;
  }
  auto dma_x = [&](int chunk, int buf) {
    const int cc = chunk * 32;
    const bool first = cc < a.s0.c;
    const char* base = first ? (const char*)a.s0.ptr + (size_t)cc * 2 : (const char*)a.s1.ptr + (size_t)(cc - a.s0.c) * 2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const void* g = aok[i] ? (const void*)(base + (first ? aoff0[i] : aoff1[i])) : a.zeros;
      dma(g, lds + C3_XR + buf * C3_XBUF + (i * 256 + w * 64) * 8);
    }
  };
  const int wrow = t >> 2;                                           // row of a 64-row weight tile this thread fetches
  const int woff = wrow * 32 + ((pos ^ swz(wrow)) * 8);              // halves, swizzle on the source
  auto dma_w12 = [&](int chunk, int buf) {
    dma(a.w12 + (size_t)chunk * C3_WBUF + woff, lds + C3_WR + buf * C3_WBUF + w * 64 * 8);
  };

  // ---- fragment addressing ------------------------------------------------------------------------------------
  // patch fragment of this wave = patch rows 2w, 2w+1.  The second row's lanes are rotated by HW - 16 columns so
  // the 16-lane ds_read_b128 groups meet 16 distinct bank slots (kernels_halo.hip).
  const int prow = 2 * w + (l31 >> 4);
  const int pcol = (l31 < 16) ? l31 : ((l31 - (C3_HW - 16)) & 15);
  const int pl = prow * C3_TW + pcol;                       // pixel index in the patch (row of y2 / out tile)
  const int rowIn = (prow + 1) * C3_HW + pcol + 1;          // its row in haloed coordinates (centre tap)
  const int rowA = 32 * w + l31;                            // haloed fragment w (linear rows)
  const int rowC = 32 * (4 + w) + l31;                      // haloed fragment 4 + w (waves 0, 1)
  const bool twoH = w < 2;                                  // wave-uniform
  auto ld = [&](const half_t* base, int row, int kc) {
    return *(const half8_t*)(base + row * 32 + ((kc ^ swz(row)) * 8));
  };
  auto zero16 = [](float16_t& x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
  };
  // bias + activation of one 32-channel accumulator tile -> fp16 row `row` of an LDS image (swizzled chunks);
  // keep = false writes zeros (t outside the image: the 3x3's zero padding applies to t, not to x)
  // (`keep` as a MASK on the packed halves: `keep ? act(..) : 0` per value compiles to an exec-mask branch around every
  // activation, kernels_c3b.hip / DESIGN 4.11)
  auto store_frag = [&](half_t* base, int row, const float16_t& acc, const float* bias, bool keep) {
    const unsigned km = keep ? 0xffffffffu : 0u;
    float4_t bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *(const float4_t*)(bias + 8 * g + 4 * khalf);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)ctd_act_fast<ACT>(acc[4 * g + e] + bv[g][e]);
      uint2 u = __builtin_bit_cast(uint2, o);
      u.x &= km;
      u.y &= km;
      *(uint2*)(base + row * 32 + ((g ^ swz(row)) * 8) + 4 * khalf) = u;
    }
  };

  // ================= S1: y1 (haloed) and y2 (patch) = act(W12 x) ==============================================
  const int nch = (a.s0.c + a.s1.c) / 32;
  float16_t accA, accB, accC;
  zero16(accA); zero16(accB); zero16(accC);
  // Both buffers are filled up front (one HBM round trip for a 64-channel x instead of two in a row: a block's
  // time is latency, not bytes); a third / fourth chunk follows into the buffer the step before it has just released.
  dma_x(0, 0);
  dma_w12(0, 0);
  if (nch > 1) {
    dma_x(1, 1);
    dma_w12(1, 1);
  }
  __syncthreads();   // the compiler's barrier sequence waits for the LDS-DMAs (vmcnt 0) first
  for (int c = 0; c < nch; ++c) {
    if (c >= 1 && c + 1 < nch) {       // buffer (c + 1) & 1 was read by step c - 1, which every wave has left
      dma_x(c + 1, (c + 1) & 1);
      dma_w12(c + 1, (c + 1) & 1);
    }
    const half_t* Xb = lds + C3_XR + (c & 1) * C3_XBUF;
    const half_t* Wb = lds + C3_WR + (c & 1) * C3_WBUF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kc = kk * 2 + khalf;
      const half8_t fw1 = ld(Wb, l31, kc);            // cv1 rows 0..31
      const half8_t fw2 = ld(Wb, 32 + l31, kc);       // cv2 rows 32..63
      const half8_t fa = ld(Xb, rowA, kc);
      const half8_t fb = ld(Xb, rowIn, kc);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fa, accA, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw2, fb, accB, 0, 0, 0);
      if (twoH) {
        const half8_t fc = ld(Xb, rowC, kc);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fc, accC, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // the x buffers and the w12 ring are free now: fetch the weights of S2-S4 under the S1 epilogue
#pragma unroll
  for (int i = 0; i < 5; ++i) {       // wm2 (288 rows) + wm1 (32 rows) = 320 rows x 4 chunks = 5 passes
    const int row = (i * 256 + t) >> 2;
    const half_t* src = row < 288 ? a.wm2 + row * 32 : a.wm1 + (row - 288) * 32;
    dma(src + ((pos ^ swz(row)) * 8), lds + C3_XR + (i * 256 + w * 64) * 8);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {       // wc3: 2 K chunks x 64 rows
    const int row = (i * 256 + t) >> 2;
    dma(a.wc3 + row * 32 + ((pos ^ swz(row)) * 8), lds + C3_WR + (i * 256 + w * 64) * 8);
  }
  store_frag(lds + C3_Y1, rowA, accA, bias_s, true);
  store_frag(lds + C3_Y2, pl, accB, bias_s + 32, true);
  if (twoH) store_frag(lds + C3_Y1, rowC, accC, bias_s, true);
  __syncthreads();

  // ================= S2: t = act(Wm1 y1) on the haloed patch, zero outside the image ===========================
  zero16(accA); zero16(accC);
  {
    const half_t* Wb = lds + C3_XR + C3_WM1;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kc = kk * 2 + khalf;
      const half8_t fw = ld(Wb, l31, kc);
      const half8_t fa = ld(lds + C3_Y1, rowA, kc);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fa, accA, 0, 0, 0);
      if (twoH) {
        const half8_t fc = ld(lds + C3_Y1, rowC, kc);
        accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fc, accC, 0, 0, 0);
      }
    }
  }
  store_frag(lds + C3_T, rowA, accA, bias_s + 64, inside(rowA));
  if (twoH) store_frag(lds + C3_T, rowC, accC, bias_s + 64, inside(rowC));
  __syncthreads();

  // ================= S3: b = y1 + act(Wm2 * t)  (3x3 over the haloed t) ========================================
  zero16(accB);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ty = tap / 3, tx = tap - 3 * ty;
    const int row = (prow + ty) * C3_HW + pcol + tx;
    const half_t* Wb = lds + C3_XR + tap * 1024;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kc = kk * 2 + khalf;
      const half8_t fw = ld(Wb, l31, kc);
      const half8_t fx = ld(lds + C3_T, row, kc);
      accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fx, accB, 0, 0, 0);
    }
  }
  // shortcut: conv output rounded to fp16, then added to y1 and rounded (the reference's half-precision
  // `x + cv2(cv1(x))`, as kernels_halo.hip does); b overwrites y1's patch rows, each element by its own lane
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4_t bv = *(const float4_t*)(bias_s + 96 + 8 * g + 4 * khalf);
    half4_t* p = (half4_t*)(lds + C3_Y1 + rowIn * 32 + ((g ^ swz(rowIn)) * 8) + 4 * khalf);
    const half4_t y = *p;
    half4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const half_t u = (half_t)ctd_act_fast<ACT>(accB[4 * g + e] + bv[e]);
      o[e] = (half_t)((float)u + (float)y[e]);
    }
    *p = o;
  }
  __syncthreads();

  if (DBG) {   // three (B,H,W,32) planes: y2, t, b
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = t + 256 * j, p = idx >> 2, ch = idx & 3;
      const int pr = p >> 4, pc = p & 15, ri = (pr + 1) * C3_HW + pc + 1;
      const int oy = y0 + pr, ox = x0 + pc;
      if (oy < a.H && ox < a.W) {
        const size_t plane = (size_t)a.B * a.H * a.W * 32;
        half_t* d = a.dbg + (((size_t)b * a.H + oy) * a.W + ox) * 32 + ch * 8;
        *(half8_t*)d = *(const half8_t*)(lds + C3_Y2 + p * 32 + ((ch ^ swz(p)) * 8));
        *(half8_t*)(d + plane) = *(const half8_t*)(lds + C3_T + ri * 32 + ((ch ^ swz(ri)) * 8));
        *(half8_t*)(d + 2 * plane) = *(const half8_t*)(lds + C3_Y1 + ri * 32 + ((ch ^ swz(ri)) * 8));
      }
    }
  }

  // ================= S4: out = act(Wc3 [b ; y2]) ================================================================
  zero16(accA); zero16(accB);   // N fragments 0 / 1
#pragma unroll
  for (int kchunk = 0; kchunk < 2; ++kchunk) {
    const half_t* Wb = lds + C3_WR + kchunk * C3_WBUF;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kc = kk * 2 + khalf;
      const half8_t fw0 = ld(Wb, l31, kc);
      const half8_t fw1 = ld(Wb, 32 + l31, kc);
      const half8_t fx = kchunk == 0 ? ld(lds + C3_Y1, rowIn, kc) : ld(lds + C3_Y2, pl, kc);
      accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw0, fx, accA, 0, 0, 0);
      accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw1, fx, accB, 0, 0, 0);
    }
  }
  half_t* Os = lds + C3_XR;   // [128][72]; the tap tiles it overwrites were last read before the previous barrier
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float16_t& acc = i ? accB : accA;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4_t bv = *(const float4_t*)(bias_s + 128 + 32 * i + 8 * g + 4 * khalf);
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)ctd_act_fast<ACT>(acc[4 * g + e] + bv[e]);
      *(half4_t*)(Os + pl * C3_OP + 32 * i + 8 * g + 4 * khalf) = o;
    }
  }
  __syncthreads();
  // 16-B channel-row stores: 8 lanes per pixel, 32 pixels per pass
  const int cch = t & 7;
#pragma unroll
  for (int it = 0; it < C3_PX / 32; ++it) {
    const int p = it * 32 + (t >> 3);
    const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
    if (oy < a.H && ox < a.W)
      *(half8_t*)((half_t*)a.dst + (((size_t)b * a.H + oy) * a.W + ox) * a.pitchD + cch * 8) =
          *(const half8_t*)(Os + p * C3_OP + cch * 8);
  }
}

}  // namespace

// Multi-layer fusions of the fp16 engine, one bit each: 1 = C3 block (this file), 2 = SPPF's three pools
// (kernels_basic.hip), 4 = stem + model.1 (kernels_fused.hip), 8 = bottleneck + cv3 of the wider C3 blocks (kernels_c3b.hip), 16 = a 128-channel ConvTranspose + its single 1x1 consumer, 32 = the last 64-channel ConvTranspose + the tap products of the 64 -> 1 one behind it (kernels_halo3.hip).  ctd_tuning_set("fuse", 0) runs the layer-per-launch
// program (the bit-identity tests and A/B runs use it).
int g_fuse = 63;

long long g_c3_min_patches = 1024;   // fewer 128-pixel patches: the per-layer kernels (ctd_tuning_set("c3_min_patches"))

bool c3_fused_supported(const C3Args& a) {
  if (!(g_fuse & 1)) return false;
  if (a.s0.c < 32 || a.s0.c % 32 || a.s1.c % 32 || a.s0.up || (a.s1.c && a.s1.up)) return false;
  if (a.s0.pitch % 8 || (a.s1.c && a.s1.pitch % 8) || a.pitchD % 8) return false;
  if (a.s0.H != a.H || a.s0.W != a.W || (a.s1.c && (a.s1.H != a.H || a.s1.W != a.W))) return false;
  if (a.act != CTD_ACT_SILU && a.act != CTD_ACT_LEAKY && a.act != CTD_ACT_RELU) return false;
  // small maps: too few patches to fill 256 CUs twice over -> the per-layer kernels do better
  const long long patches = (long long)a.B * ((a.H + C3_TH - 1) / C3_TH) * ((a.W + C3_TW - 1) / C3_TW);
  return patches >= g_c3_min_patches;
}

void launch_c3_fused(const C3Args& a, hipStream_t st) {
  const int tilesX = (a.W + C3_TW - 1) / C3_TW, tilesY = (a.H + C3_TH - 1) / C3_TH;
  const dim3 grid((unsigned)(tilesX * tilesY * a.B), 1, 1);
  if (a.dbg) {   // selftest: intermediates dumped
    switch (a.act) {
      case CTD_ACT_SILU: hipLaunchKernelGGL((c3_fused_kernel<CTD_ACT_SILU, true>), grid, dim3(256), 0, st, a); break;
      case CTD_ACT_LEAKY: hipLaunchKernelGGL((c3_fused_kernel<CTD_ACT_LEAKY, true>), grid, dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL((c3_fused_kernel<CTD_ACT_RELU, true>), grid, dim3(256), 0, st, a); break;
    }
    return;
  }
  switch (a.act) {
    case CTD_ACT_SILU: hipLaunchKernelGGL((c3_fused_kernel<CTD_ACT_SILU>), grid, dim3(256), 0, st, a); break;
    case CTD_ACT_LEAKY: hipLaunchKernelGGL((c3_fused_kernel<CTD_ACT_LEAKY>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((c3_fused_kernel<CTD_ACT_RELU>), grid, dim3(256), 0, st, a); break;
  }
}
