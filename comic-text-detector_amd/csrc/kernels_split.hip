// Split-operand convolution for gfx950: fp32 activations and fp32 weights, each written as the sum of two fp16
// numbers (x = xh + xl, |xl| <= ulp(xh) / 2), multiplied on the fp16 matrix cores with three MFMAs per product
//
//     w * x  ~=  wh*xh + wh*xl + wl*xh                 (the dropped term wl*xl is <= 2^-22 |w x|)
//
// and accumulated in fp32.  Every fp16 x fp16 product is exact in the fp32 accumulator (11 x 11 mantissa bits), so the
// result carries the operands to ~22 bits -- below the rounding noise of an fp32 accumulation chain of the same
// length (scripts/experiments/split_accuracy.py: K = 1152, representation error 1.0e-7 rms against 4.8e-7 rms of an
// fp32 matmul).  `v_mfma_f32_32x32x16_f16` runs at 16x the rate of the f32-operand MFMA the exact-fp32 engine uses
// (kernels_f32.hip), so three of them are a ~5x higher ceiling for the SAME activations in HBM (fp32 NHWC) and the
// same op program.  This is the engine of `precision="fp32s"`: the reference runs fp32 (reference inference.py:129,
// basemodel.py:222-244), and north_star's "identical boxes / mask bit-exact after threshold" is an fp32-level
// statement.
//
//   D[n][m] = sum_k W[n][k] * X[m][k]       n: output channel, m: output pixel, k = tap * Ctot + c
//
// Covers what kernels_f32.hip covers for channel counts that are multiples of 32: Conv k x k stride 1 / 2 (+ folded
// BN, activation, residual), two concatenated sources, nearest x2 upsampled sources, ConvTranspose 4x4/s2/p1 as four
// 2x2-tap phase GEMMs, and the first layer over the 4-channel image (K = 8 taps per step).  Two siblings share its weight
// packing and its epilogue (split_epilogue.h): kernels_split_halo.hip (3x3 / ConvT on 256-pixel haloed patches of
// split-plane tensors -- launch_conv_split dispatches to it) and kernels_split_stem.hip (the first layer from the page).
//
// Weights: split once on the host.  Each output channel is first scaled by a power of two so that its largest
// weight lies in [512, 1024) -- exact, undone by `oscale` in the epilogue -- which keeps the low halves of all but
// negligible weights in fp16's normal range.  Layout: [phase][Npad / 32][K / 32][32 rows][32 halves] (one 2-KB
// block per 32 output channels and K step, so any N tile that is a multiple of 32 reads contiguous blocks), hi
// plane followed by the lo plane.
//
// fp32 activations (the image, tensors the heads or the pools touch): split in the kernel between the global load and
// the LDS store (2 x v_cvt_f16_f32 + v_sub per element and K step; ~1/5 of the MFMA cycles of a step).  No scaling: a
// value beyond fp16's range (|x| > 65504) becomes inf - inf = NaN in the output, loudly.
//
// Split-plane activations (`x_sp` / `d_sp` / `r_sp`): a tensor that only this kernel writes and reads is kept SPLIT in
// HBM -- per pixel and 32-channel group the 32 hi halves followed by the 32 lo halves, in the 128 B its 32 floats would
// occupy (same pitch, same offsets for channel slices at multiples of 32).  The producer's epilogue splits each value
// once; every consumer (each tap, each N tile) then moves 16-B chunks straight to LDS by LDS-DMA like the weights, with
// no conversion and no register staging in the K loop.  hi + lo is what the matrix cores saw before, so a layer's
// result is unchanged; only a residual input is hi + lo (22 bits) instead of the fp32 value.  The engine decides per
// tensor (engine.hip: every writer and reader must be this kernel).
//
// Tiling: 256 threads = 4 waves, BN x 128-pixel tile, K step 32 channels.  LDS rows of 32 halves (64 B, XOR-swizzled
// 16-B chunks), hi and lo planes for pixels and weights, double buffered: 64 KB at BN = 128 (2 blocks per CU).  Per K
// step and wave at BN = 128: 16 ds_read_b128 feed 24 MFMAs.  Epilogue: accumulators x oscale + bias -> an fp32 tile in LDS
// (the K loop's buffers) -> split_store_tile: activation, residual, coalesced 16-B stores of fp32 or split-plane rows.
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>

#include "kernels.h"
#include "split_epilogue.h"

namespace {

constexpr int SBK = 32;    // channels per K step

__device__ __forceinline__ void split8(const float4_t& a, const float4_t& b, half8_t& hi, half8_t& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const half_t h0 = (half_t)a[e], h1 = (half_t)b[e];
    hi[e] = h0;
    hi[4 + e] = h1;
    lo[e] = (half_t)(a[e] - (float)h0);
    lo[4 + e] = (half_t)(b[e] - (float)h1);
  }
}

// WDMA: weight tiles by LDS-DMA (global_load_lds, swizzle on the source chunk) instead of through registers
// SBM = pixels per block: 128, or 256 for the 64-channel outputs (the weight tile of a K step then feeds twice the MFMAs
// and every wave holds a 2x2-fragment tile like the 128 x 128 configuration)
// XSP: the sources are stored SPLIT (see "Split-plane activations" above): their 16-B chunks go to LDS by LDS-DMA like the
// weights, and nothing is converted in the K loop
template <int BN, int SBM, int WGN, int WGM, bool WDMA, bool XSP>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(ConvArgs a) {
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  constexpr int TN = BN / (32 * WGN);
  constexpr int TM = SBM / (32 * WGM);
  constexpr int AR = SBM / 64;                  // pixel rows each thread stages
  static_assert(WGN * WGM == 4, "4 waves");
  constexpr int XT = SBM * SBK;                 // halves of one pixel plane
  constexpr int WT = BN * SBK;                  // halves of one weight plane
  constexpr int WCH = BN * 4;                   // 16-B chunks of one weight plane
  constexpr int WROWS = (WCH + 255) / 256;
  // one LDS object (see kernels_igemm.hip: a second array costs a vmcnt(0) in front of every K step's first ds_read)
  // pixels: [buf][hi, lo][SBM][32]; weights: [buf][hi, lo][BN][32]
  __shared__ __attribute__((aligned(16))) half_t lds[4 * XT + 4 * WT];
  half_t* Xs = lds;
  half_t* Ws = lds + 4 * XT;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave % WGN, wm = wave / WGN;
  const int l31 = lane & 31, khalf = lane >> 5;

  // XCD-aware tile order: consecutive block ids land on different XCDs, so give each XCD a contiguous run of tiles
  // (the N tiles / phases of one pixel tile and neighbouring pixel tiles share an L2)
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + SBM - 1) / SBM;
  const int nblk = ntn * ntm * a.nphase;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, within = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tile_n = bid % ntn;
  bid /= ntn;
  int phase = 0;
  if (a.nphase == 4) {
    phase = bid & 3;
    bid >>= 2;
  }
  const int tile_m = bid;
  const int n0 = tile_n * BN, m0 = tile_m * SBM;

  int dy0 = a.dy0, dx0 = a.dx0, ooy = a.ooy, oox = a.oox;
  const half_t* __restrict__ wh = (const half_t*)a.w;
  const half_t* __restrict__ wl = (const half_t*)a.w2;
  if (a.nphase == 4) {   // ConvTranspose 4x4 s2 p1: sub-pixel phase (py, px), 2x2 taps
    const int py = phase >> 1, px = phase & 1;
    dy0 = py ? 0 : -1;
    dx0 = px ? 0 : -1;
    ooy = py;
    oox = px;
    wh += (size_t)phase * a.w_phase_stride;
    wl += (size_t)phase * a.w_phase_stride;
  }
  const int Ct = a.s0.c + a.s1.c;
  const int nk = a.K / SBK;
  auto swz = [](int row) { return (row >> 2) & 3; };

  // this thread's pixel rows (rows t/4 + 64 i; 8 channels = two 16-B loads at chunk t%4)
  const int seg = t & 3;
  int pb[AR], poy[AR], pox[AR];
  bool pv[AR];
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + (t >> 2) + 64 * i;
    pv[i] = m < a.M;
    const int mm = pv[i] ? m : 0;
    pox[i] = mm % a.Mw;
    const int q = mm / a.Mw;
    poy[i] = q % a.Mh;
    pb[i] = q / a.Mh;
  }
  // this thread's weight chunks: chunk q = t + 256 i of the tile = row q/4, LDS chunk position q%4
  int woff[WROWS], wlds[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int q = t + 256 * i;
    const int r = (q >> 2) % BN, pos = q & 3;
    const int srcc = WDMA ? (pos ^ swz(r)) : pos;                    // DMA writes lane-linear: swizzle the source
    woff[i] = (((n0 + r) >> 5) * nk * 32 + (r & 31)) * SBK + srcc * 8;   // + ks * 1024 per K step
    wlds[i] = r * SBK + (WDMA ? pos : (pos ^ swz(r))) * 8;
  }

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  auto dma = [&](const void* g, half_t* plane, int i) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(plane + (i * 256 + wave * 64) * 8), 16, 0, 0);
  };

  float4_t ra[AR][2];
  half8_t rwh[WROWS], rwl[WROWS];
  int nx_cc = 0, nx_ty = 0, nx_tx = 0;   // (channel offset, tap) of the NEXT tile to load
  auto load_tile = [&](int ks, int dbuf) {
    if (XSP) {
      const int cc = nx_cc, ty = nx_ty, tx = nx_tx;
      nx_cc += SBK;
      if (nx_cc == Ct) {
        nx_cc = 0;
        if (++nx_tx == a.KW) nx_tx = 0, ++nx_ty;
      }
      const bool first = cc < a.s0.c;
      const SrcView& s = first ? a.s0 : a.s1;
      const int ch = first ? cc : cc - a.s0.c;                       // a multiple of 32: one 128-B group per pixel
      const int srcb = (seg ^ ((t >> 4) & 3)) * 16;                  // swz(row): rows t/4 + 64 i share it
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        const int iy = poy[i] * a.stride + dy0 + ty, ix = pox[i] * a.stride + dx0 + tx;
        const bool ok = pv[i] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
        const int sy = s.up ? iy >> 1 : iy, sx = s.up ? ix >> 1 : ix;
        const char* p = (const char*)((const float*)s.ptr + ((size_t)((size_t)pb[i] * s.H + sy) * s.W + sx) * s.pitch + ch) + srcb;
        dma(ok ? (const void*)p : a.zeros, Xs + (dbuf * 2 + 0) * XT, i);
        dma(ok ? (const void*)(p + 64) : a.zeros, Xs + (dbuf * 2 + 1) * XT, i);
      }
    } else if (Ct == 4) {
      // the stem: the image is stored with a zero 4th channel, so one tap = one 16-B chunk and a K step = 8 taps
      // (this thread's two chunks are taps 8 ks + 2 seg, + 1); taps beyond KH * KW pad K to a multiple of 32 (zero weights)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tap = ks * 8 + seg * 2 + h;
        const int ty = tap / a.KW, tx = tap - ty * a.KW;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
          const int iy = poy[i] * a.stride + dy0 + ty, ix = pox[i] * a.stride + dx0 + tx;
          const bool ok = pv[i] && ty < a.KH && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
          float4_t v = {0.f, 0.f, 0.f, 0.f};
          if (ok) v = *(const float4_t*)((const float*)a.s0.ptr + ((size_t)((size_t)pb[i] * a.s0.H + iy) * a.s0.W + ix) * a.s0.pitch);
          ra[i][h] = v;
        }
      }
    } else {
    const int cc = nx_cc, ty = nx_ty, tx = nx_tx;
    nx_cc += SBK;
    if (nx_cc == Ct) {
      nx_cc = 0;
      if (++nx_tx == a.KW) nx_tx = 0, ++nx_ty;
    }
    const bool first = cc < a.s0.c;
    const SrcView& s = first ? a.s0 : a.s1;
    const int ch = (first ? cc : cc - a.s0.c) + seg * 8;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const int iy = poy[i] * a.stride + dy0 + ty, ix = pox[i] * a.stride + dx0 + tx;
      const bool ok = pv[i] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int sy = s.up ? iy >> 1 : iy, sx = s.up ? ix >> 1 : ix;
      float4_t v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const float* p = (const float*)s.ptr + ((size_t)((size_t)pb[i] * s.H + sy) * s.W + sx) * s.pitch + ch;
        v0 = *(const float4_t*)p;
        v1 = *(const float4_t*)(p + 4);
      }
      ra[i][0] = v0;
      ra[i][1] = v1;
    }
    }
    const int kofs = ks * (32 * SBK);
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      if (WCH >= 256 * (i + 1) || t + 256 * i < WCH) {
        if (WDMA) {
          dma(wh + woff[i] + kofs, Ws + (dbuf * 2 + 0) * WT, i);
          dma(wl + woff[i] + kofs, Ws + (dbuf * 2 + 1) * WT, i);
        } else {
          rwh[i] = *(const half8_t*)(wh + woff[i] + kofs);
          rwl[i] = *(const half8_t*)(wl + woff[i] + kofs);
        }
      }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < (XSP ? 0 : AR); ++i) {
      const int r = (t >> 2) + 64 * i;
      half8_t hi, lo;
      split8(ra[i][0], ra[i][1], hi, lo);
      const int o = r * SBK + ((seg ^ swz(r)) * 8);
      *(half8_t*)(Xs + (buf * 2 + 0) * XT + o) = hi;
      *(half8_t*)(Xs + (buf * 2 + 1) * XT + o) = lo;
    }
    if (!WDMA) {
#pragma unroll
      for (int i = 0; i < WROWS; ++i)
        if (WCH >= 256 * (i + 1) || t + 256 * i < WCH) {
          *(half8_t*)(Ws + (buf * 2 + 0) * WT + wlds[i]) = rwh[i];
          *(half8_t*)(Ws + (buf * 2 + 1) * WT + wlds[i]) = rwl[i];
        }
    }
  };

  float16_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_tile(0, 0);
  store_tile(0);
  __syncthreads();     // with LDS-DMA pending the compiler's barrier sequence waits vmcnt(0) first
  const int fl = swz(l31);   // rows of one fragment differ by multiples of 32: same swizzle
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_tile(ks + 1, buf ^ 1);
    const half_t* Xh = Xs + (buf * 2 + 0) * XT + (wm * TM * 32 + l31) * SBK;
    const half_t* Xl = Xh + XT;
    const half_t* Wh = Ws + (buf * 2 + 0) * WT + (wn * TN * 32 + l31) * SBK;
    const half_t* Wl = Wh + WT;
#pragma unroll
    for (int kk = 0; kk < SBK / 16; ++kk) {
      const int co = ((kk * 2 + khalf) ^ fl) * 8;
      half8_t fwh[TN], fwl[TN], fxh[TM], fxl[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        fwh[i] = *(const half8_t*)(Wh + i * 32 * SBK + co);
        fwl[i] = *(const half8_t*)(Wl + i * 32 * SBK + co);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        fxh[j] = *(const half8_t*)(Xh + j * 32 * SBK + co);
        fxl[j] = *(const half8_t*)(Xl + j * 32 * SBK + co);
      }
      // the two small terms first, the leading term last; term-major so that dependent MFMAs are TN*TM apart
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[i], fxh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[i], fxl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[i], fxh[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: undo the weight scale + bias -> LDS tile -> activation (+ residual) -> NHWC, whole lines per wave ----
  // Straight from the accumulators a lane stored 16-B pieces of 32 different pixel rows per instruction, with the
  // activation switched per value and every store variant inlined per register group: 8 200 of the kernel's 8 900
  // instructions, most of a short-K block's life and more code than the instruction cache holds.  The tile goes through
  // LDS instead (the K loop's buffers are free; 16-B chunks XOR-swizzled by the pixel) and a short run-time loop hands
  // every thread 8 consecutive channels of one pixel: residual loads and stores are coalesced 16-B accesses (8 lanes
  // per 128-B line), and the code is a few hundred instructions.
  {
    float* stg = (float*)lds;                          // [SBM][BN] floats <= the K loop's buffers (64 KB at BN = 128)
    static_assert(SBM * BN * 4 <= (4 * XT + 4 * WT) * 2, "staging tile fits the K loop's LDS");
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = (wn * TN + i) * 32 + 4 * hi + 8 * g;
        const float4_t os = *(const float4_t*)(a.oscale + n0 + nl), bs = *(const float4_t*)(a.bias + n0 + nl);   // padded to Npad
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const int p = (wm * TM + j) * 32 + l31;
          float4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] * os[e] + bs[e];   // oscale is a power of two (exact product)
          *(float4_t*)(stg + p * BN + (split_stg_chunk<BN>(p, nl >> 2) << 2)) = v;
        }
      }
    __syncthreads();
    split_store_tile<BN, SBM, 256>(stg, a, n0, t, [&](int p, int, size_t& opix) {
      const int m = m0 + p;
      if (m >= a.M) return false;
      const unsigned tq = (unsigned)(((unsigned long long)(unsigned)m * a.mw_mul) >> a.mw_sh);
      const int ox = m - (int)tq * a.Mw;
      const unsigned bb = (unsigned)(((unsigned long long)tq * a.mh_mul) >> a.mh_sh);
      const int oy = (int)tq - (int)bb * a.Mh;
      opix = ((size_t)bb * a.oH + (oy * a.osy + ooy)) * a.oW + (ox * a.osx + oox);
      return true;
    });
  }
}

// n / d == (uint64(n) * mul) >> sh for every n < 2^31 (Granlund-Montgomery, 31-bit dividend)
void split_magic_div(int d, unsigned& mul, unsigned& sh) {
  int L = 0;
  while ((1ll << L) < d) ++L;
  sh = 31 + L;
  mul = (unsigned)(((1ull << sh) / (unsigned)d) + 1);
}

template <int BN, int SBM, int WGN, int WGM>
void launch_split_cfg(const ConvArgs& a_in, hipStream_t st) {
  ConvArgs a = a_in;
  split_magic_div(a.Mw, a.mw_mul, a.mw_sh);
  split_magic_div(a.Mh, a.mh_mul, a.mh_sh);
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + SBM - 1) / SBM;
  const dim3 grid((unsigned)(ntn * ntm * a.nphase));
  if (a.x_sp) { hipLaunchKernelGGL((conv_split_kernel<BN, SBM, WGN, WGM, true, true>), grid, dim3(256), 0, st, a); return; }
#ifdef CTD_AB_VARIANTS
  if (!g_split_wdma) { hipLaunchKernelGGL((conv_split_kernel<BN, SBM, WGN, WGM, false, false>), grid, dim3(256), 0, st, a); return; }
#endif
  hipLaunchKernelGGL((conv_split_kernel<BN, SBM, WGN, WGM, true, false>), grid, dim3(256), 0, st, a);
}

}  // namespace

#ifdef CTD_AB_VARIANTS    // selftest build only: weight tiles through registers instead of LDS-DMA (7 % slower)
int g_split_wdma = 1;
// ... and 256-pixel blocks for 64-channel N tiles -- measured 3-20 % SLOWER than 128
int g_split_bm256 = 0;    // (profiles/r03_split_selftest.txt): the 80-KB block leaves no LDS for a third block per CU
#endif

// f32 sources / destination with 16-B aligned channel rows, source channel counts multiples of 32
bool conv_split_supported(const ConvArgs& a) {
  const bool stem = a.s0.c == 4 && a.s1.c == 0 && !a.s0.up && a.nphase == 1;   // K = taps x 4, padded to 32 by the packer
  if (!stem && (a.s0.c % SBK || a.s1.c % SBK || a.s0.c == 0)) return false;
  if (a.s0.pitch % 4 || (a.s1.c && a.s1.pitch % 4) || (a.N >= 4 && a.pitchD % 4)) return false;
  if (a.res && a.pitchR % 4) return false;
  if (a.K % SBK || a.Npad % 32) return false;
  if (!a.w2 || !a.oscale) return false;
  // split-plane tensors are addressed in 32-channel groups of 128 B
  if (a.x_sp && (stem || a.s0.pitch % 32 || (a.s1.c && a.s1.pitch % 32))) return false;
  if (a.d_sp && (a.N % 32 || a.pitchD % 32)) return false;
  if (a.r_sp && (!a.res || a.N % 32 || a.pitchR % 32)) return false;
  return a.nphase == 1 || a.nphase == 4;
}

void launch_conv_split(const ConvArgs& a, hipStream_t st) {
  if (conv_split_halo_supported(a)) return launch_conv_split_halo(a, st);
  int bn = a.Npad % 128 == 0 ? 128 : (a.Npad % 64 == 0 ? 64 : 32);
  // small maps: narrower N tiles give 2-4x the blocks (the packing is in 32-row blocks, any multiple of 32 reads it)
  const long long ntm = (a.M + 127) / 128;
  while (bn > 32 && (a.Npad / bn) * ntm * a.nphase < 512) bn >>= 1;
  if (bn == 128) launch_split_cfg<128, 128, 2, 2>(a, st);
#ifdef CTD_AB_VARIANTS
  else if (bn == 64 && g_split_bm256 && (a.Npad / 64) * ((a.M + 255) / 256) * a.nphase >= 1024) launch_split_cfg<64, 256, 1, 4>(a, st);
#endif
  else if (bn == 64) launch_split_cfg<64, 128, 1, 4>(a, st);
  else launch_split_cfg<32, 128, 1, 4>(a, st);
}

// logical weights float [nphase][N][K] (K index = tap * Ctot + c) -> hi plane, lo plane ([nphase][npad/32][K/32][32][32]
// halves each) and the per-channel output scale (npad floats, 1 / the power of two the channel was multiplied by)
void split_pack_weights(const float* logical, int nphase, int N, int K, int npad, std::vector<half_t>& out,
                        std::vector<float>& oscale) {
  const int nk = K / SBK;
  const size_t plane = (size_t)nphase * npad * K;
  out.assign(2 * plane, (half_t)0.f);
  oscale.assign(npad, 1.f);
  for (int n = 0; n < N; ++n) {
    float mx = 0.f;
    for (int ph = 0; ph < nphase; ++ph) {
      const float* src = logical + ((size_t)ph * N + n) * K;
      for (int k = 0; k < K; ++k) mx = std::fmax(mx, std::fabs(src[k]));
    }
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) {
      (void)std::frexp(mx, &e);                 // mx = f * 2^e, f in [0.5, 1)
      e = 10 - e;                               // mx * 2^e in [512, 1024)
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    const float s = std::ldexp(1.f, e);
    oscale[n] = std::ldexp(1.f, -e);
    for (int ph = 0; ph < nphase; ++ph) {
      const float* src = logical + ((size_t)ph * N + n) * K;
      for (int k = 0; k < K; ++k) {
        const float w = src[k] * s;
        const half_t h = (half_t)w;
        const half_t l = (half_t)(w - (float)h);
        const size_t dst = (size_t)ph * npad * K + ((((size_t)(n >> 5) * nk + k / SBK) * 32 + (n & 31)) * SBK + k % SBK);
        out[dst] = h;
        out[plane + dst] = l;
      }
    }
  }
}
