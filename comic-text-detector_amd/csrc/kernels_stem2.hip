// Stem + second convolution in one kernel (gfx950):
//     s   = act0(Conv 6x6/s2/p2, 3 -> 32 (network input))       yolo cfg layer 0 (+ the /255 of inference.py:77-82)
//     out = act1(Conv 3x3/s2/p1, 32 -> 64 (s))                  yolo cfg layer 1
//
// Why: the stem's output is the largest tensor of the network (32 channels at H/2 x W/2: 537 MB per 32 pages at
// 1024x1024) and its only consumer is layer 1.  As two launches it is written to HBM and read back, 1.07 GB of the
// backbone's 5.6 GB and 0.54 ms of its 2.5 ms.  Here a block owns a 16x8 patch of layer 1's output, computes the
// 33x17 stem pixels under it into LDS (3 % more stem work than the unfused tiling) and never writes them.
//
// Arithmetic is the two kernels' own, step for step -- the stem as kernels_fused.hip's stem_mfma_kernel (K = 6 rows x
// 24 (8 columns x 3 channels, columns 6, 7 zero weights), fp16 patch of exact integers for uint8 input with 128/255 in
// the packed weights, sums x 1/128), layer 1 as the implicit-GEMM kernel (K = tap * 32 + channel in 32-channel steps,
// fp32 accumulation) -- so the result is bit-identical to the two launches; the selftest checks exactly that.
//
// Round 4: 512 threads per block instead of 256.  LDS (74 KB) allows two blocks per CU whatever their size, and the kernel is
// a chain of dependent phases (page loads -> fp16 patch -> stem MFMAs + SiLU -> layer-1 MFMAs -> store) with one other block
// to hide behind: eight waves per block = four per SIMD double what is in flight under every wait.  The arithmetic per
// accumulator is unchanged (a wave now owns ONE of layer 1's two N fragments for its two output rows; stem fragments are
// dealt over eight waves), so the result is still bit-identical to the two launches.
//
// LDS (73.9 KB, two blocks per CU): the nine 64x32 weight tiles of layer 1 (36.9 KB; tiles 5-8 share their space with
// the fp16 input patch, which is dead when they are needed) and the stem patch (576 rows x 64 B, XOR-swizzled 16-B
// chunks; later the staged output tile).  Stem pixels outside the stem map are ZERO (layer 1's padding pads the
// stem's output, not the image).
//
// Round 6: the stem patch is stored DE-INTERLEAVED by column parity -- a stem row's 17 even columns, then its 16 odd ones
// (`srow`).  Layer 1 is stride 2: the lanes of a pixel-fragment read walked rows 2 apart, i.e. 2 of the 4 rows of a 256-B
// LDS line.  With the parity planes a tap reads UNIT-stride rows (tx = 0: even plane at c, tx = 1: odd plane at c, tx = 2:
// even plane at c + 1), and the lanes of the fragment's second patch row take their columns rotated by 14, so that the
// 16-lane groups of a ds_read_b128 as MI355X_MICROARCH.md lists them ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}) meet 16
// different rows mod 16 = 16 different 16-B slots (tests/test_stem2_emul.py counts them).  Same values, same K order:
// bit-identical.  MEASURED NEUTRAL (profiles/r06_pmc_stem_c3.txt): 0.337 ms before and after, and SQ_LDS_BANK_CONFLICT did
// not move (3.85e7 vs 3.88e7 per dispatch) -- the conflict cycles of round 5's PMC are the STEM phase's (4-B patch stores,
// 8-B fragment stores), and they do not bound the kernel: its four waves per SIMD issue ~890 VALU instructions each, 52
// SiLUs per lane among them with two quarter-rate transcendentals apiece, ~4.9 k VALU cycles per wave = 84 % of a SIMD's
// cycles over a block's ~19 k-cycle life.  The kernel is VALU-bound on its activations.
#include <type_traits>

#include "kernels.h"

namespace {

constexpr int S2_TW = 16, S2_TH = 8;                     // layer-1 output patch
constexpr int S2_SW = 2 * S2_TW + 1, S2_SH = 2 * S2_TH + 1;   // stem pixels under it: 33 x 17
constexpr int S2_SPX = S2_SW * S2_SH;                    // 561
constexpr int S2_SFRAG = (S2_SPX + 31) / 32;             // 18 MFMA pixel fragments
constexpr int S2_SROWS = S2_SFRAG * 32;                  // 576 LDS rows
constexpr int S2_IW = 2 * S2_SW + 4, S2_IH = 2 * S2_SH + 4;   // input patch 70 x 38
constexpr int S2_PITCH = 72 * 3;                         // input-patch row pitch in halves (72 columns x 3 channels)
constexpr int S2_INH = S2_IH * S2_PITCH + 16;            // halves of the input patch (+ slack for the last fragment read)
constexpr int S2_WTILE = 64 * 32;                        // one tap's weight tile (halves)
constexpr int S2_W0 = 0;                                 // taps 0-4
constexpr int S2_W5 = 5 * S2_WTILE;                      // taps 5-8, aliasing the input patch
constexpr int S2_IN = S2_W5;
constexpr int S2_U = S2_W5 + (S2_INH > 4 * S2_WTILE ? S2_INH : 4 * S2_WTILE);   // end of the union region
constexpr int S2_S = (S2_U + 7) / 8 * 8;                 // stem patch [576][32]; later the output tile [128][72]
constexpr int S2_LDS = S2_S + S2_SROWS * 32;
constexpr int S2_OP = 72;
constexpr int S2_NT = 512;                               // threads per block (8 waves)
static_assert(S2_TW * S2_TH * S2_OP <= S2_SROWS * 32, "output tile fits the stem patch");

// ACT0 is a compile-time constant evaluated by the stem kernel's own `ctd_act` (same arithmetic, no per-element
// switch on a run-time value)
template <int ACT0, int ACT1>
__global__ __launch_bounds__(S2_NT, 4) void stem_conv2_kernel(Stem2Args a) {   // 2 blocks / CU = 4 waves / SIMD
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  __shared__ __attribute__((aligned(16))) half_t lds[S2_LDS + 2 * 96];
  float* bias_s = (float*)(lds + S2_LDS);     // [0,32) stem, [32,96) layer 1
  half_t* patch = lds + S2_IN;

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int H = a.H, W = a.W;
  const int Hs = H / 2, Ws = W / 2;           // stem map
  const int Ho = H / 4, Wo = W / 4;           // layer-1 map

  const int tilesX = (Wo + S2_TW - 1) / S2_TW, tilesY = (Ho + S2_TH - 1) / S2_TH;
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
  const int oy0 = tpy * S2_TH, ox0 = tpx * S2_TW;
  const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;       // first stem pixel of the patch
  const int iy0 = 2 * sy0 - 2, ix0 = 2 * sx0 - 2;       // first input pixel (= 4 * ox0 - 4: 4-B aligned byte rows)

  if (t < 96) bias_s[t] = t < 32 ? a.bias0[t] : a.bias1[t - 32];

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  auto swz = [](int row) { return (row >> 2) & 3; };
  // LDS row of stem pixel (py, px) of the 33 x 17 patch: even columns first, then the odd ones
  auto srow = [](int py, int px) { return py * S2_SW + ((px & 1) ? 17 + (px >> 1) : (px >> 1)); };
  const int pos = t & 3;
  // layer-1 weight tiles [tap][64][32] -> LDS by LDS-DMA, swizzle on the source chunk
  const int wg = w >> 2, w4 = w & 3, t4 = t & 255;      // wave group (0 / 1), wave and thread inside it
  auto dma_w = [&](int pass, half_t* dst_base) {        // pass = 256 chunks = 64 rows = one tap tile, by ONE wave group
    const int row = (pass * 256 + t4) >> 2;
    __builtin_amdgcn_global_load_lds((gptr_t)(a.w1 + row * 32 + ((pos ^ swz(row)) * 8)),
                                     (lptr_t)(dst_base + (w4 * 64) * 8), 16, 0, 0);
  };
  // tiles 0-4: group 0 fetches 0, 2, 4, group 1 fetches 1, 3
#pragma unroll
  for (int i = 0; i < 5; ++i)
    if ((i & 1) == wg) dma_w(i, lds + S2_W0 + i * S2_WTILE);

  // stem A fragments: 9 k-steps x (32 channels x 16 k), one 16-B load each
  half8_t wf[9];
  const bool u8in = a.in_fmt != CTD_IN_NCHW_F32;
  const half_t* wsel = a.wfrag + (u8in ? 9 * 64 * 8 : 0);
#pragma unroll
  for (int s = 0; s < 9; ++s) wf[s] = *(const half8_t*)(wsel + ((size_t)s * 64 + lane) * 8);

  // ---- stage the input patch as fp16 [row][col][channel] -------------------------------------------------------
  const bool interior = iy0 >= 0 && iy0 + S2_IH <= H && ix0 >= 0 && ix0 + 72 <= W;    // block uniform
  if (u8in && interior) {
    // rows are 210 B at a 4-B aligned address: 53 dword loads per row (the last one runs 2 B into column 70, which
    // exists: ix0 + 72 <= W), all issued before the first use
    constexpr int DPR = 53, NDW = S2_IH * DPR, NIT = (NDW + S2_NT - 1) / S2_NT;   // 2014 dwords, 4 per thread
    const uint8_t* src = (const uint8_t*)a.in + ((size_t)b * H * W + (size_t)iy0 * W + ix0) * 3;
    uint32_t vv[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = t + S2_NT * k;
      const int ii = i < NDW ? i : 0;                      // clamped address instead of a branch around the load
      const int r = ii / DPR, d = ii - r * DPR;
      vv[k] = *(const uint32_t*)(src + (size_t)r * W * 3 + d * 4);
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = t + S2_NT * k;
      const int r = i / DPR, d = i - r * DPR;
      if (i < NDW) {
        half_t* p = patch + r * S2_PITCH + d * 4;
        const half2_t lo = {(half_t)(float)(vv[k] & 255u), (half_t)(float)((vv[k] >> 8) & 255u)};
        *(half2_t*)p = lo;
        if (d < DPR - 1) {
          const half2_t hi2 = {(half_t)(float)((vv[k] >> 16) & 255u), (half_t)(float)(vv[k] >> 24)};
          *(half2_t*)(p + 2) = hi2;
        }
      }
    }
  } else {
    // border patches / float input: one element per load with bounds checks.  Clamped addresses instead of branches
    // and the input format as a compile-time tag: all 16 loads of a round are in flight together
    auto stage = [&](auto u8tag) {
      constexpr bool U8 = decltype(u8tag)::value;
      constexpr int NEL = 3 * S2_IH * S2_IW;            // 7980
      constexpr int RND = 16, NR = (NEL + S2_NT * RND - 1) / (S2_NT * RND);   // one round of 16 loads per thread
      const uint8_t* src8 = (const uint8_t*)a.in + (size_t)b * H * W * 3;
      const float* src32 = (const float*)a.in + (size_t)b * 3 * H * W;
#pragma unroll 1
      for (int rd = 0; rd < NR; ++rd) {
        float vf[RND];
        int dsti[RND];
#pragma unroll
        for (int k = 0; k < RND; ++k) {
          const int i = t + S2_NT * (rd * RND + k);
          int r, q, c;
          if (U8) { r = i / (S2_IW * 3); const int j = i - r * (S2_IW * 3); q = j / 3; c = j - 3 * q; }
          else { c = i / (S2_IH * S2_IW); const int j = i - c * (S2_IH * S2_IW); r = j / S2_IW; q = j - r * S2_IW; }
          const int iy = iy0 + r, ix = ix0 + q;
          const bool ok = i < NEL && iy >= 0 && iy < H && ix >= 0 && ix < W;
          if (U8) vf[k] = (float)src8[ok ? ((size_t)iy * W + ix) * 3 + c : 0];
          else vf[k] = src32[ok ? ((size_t)c * H + iy) * W + ix : 0];
          if (!ok) vf[k] = 0.f;
          dsti[k] = i < NEL ? r * S2_PITCH + q * 3 + c : -1;
        }
#pragma unroll
        for (int k = 0; k < RND; ++k)
          if (dsti[k] >= 0) patch[dsti[k]] = (half_t)vf[k];
      }
    };
    if (u8in) stage(std::true_type{});
    else stage(std::false_type{});
  }
  // pad columns 70, 71 (zero weights meet them: they must be finite) and the slack behind the last row
  for (int i = t; i < S2_IH * 6; i += S2_NT) patch[(i / 6) * S2_PITCH + S2_IW * 3 + i % 6] = (half_t)0.f;
  if (t < 16) patch[S2_IH * S2_PITCH + t] = (half_t)0.f;
  __syncthreads();

  // ---- stem: 18 pixel fragments of 32 stem pixels (linear index p = py * 33 + px), 9 k-steps each ----------------
  const float oscale = u8in ? 1.0f / 128.0f : 1.0f;
  half_t* S = lds + S2_S;
#pragma unroll 1
  for (int f = w; f < S2_SFRAG; f += S2_NT / 64) {
    const int p = 32 * f + l31;
    const int pc = p < S2_SPX ? p : S2_SPX - 1;
    const int py = pc / S2_SW, px = pc - py * S2_SW;
    const half_t* prow = patch + (2 * py) * S2_PITCH + px * 6;
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      // k group G = 2s + khalf: row ky = G / 3, first of 8 consecutive (col, channel) entries j0 = (G % 3) * 8
      const int offA = ((2 * s) / 3) * S2_PITCH + ((2 * s) % 3) * 8;
      const int offB = ((2 * s + 1) / 3) * S2_PITCH + ((2 * s + 1) % 3) * 8;
      const uint32_t* q = (const uint32_t*)(prow + (khalf ? offB : offA));   // 4-B aligned
      union { uint32_t u[4]; half8_t h; } fx;
      fx.u[0] = q[0]; fx.u[1] = q[1]; fx.u[2] = q[2]; fx.u[3] = q[3];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], fx.h, acc, 0, 0, 0);
    }
    const int sy = sy0 + py, sx = sx0 + px;
    // Stem pixels outside the stem map are zero: a MASK on the packed halves, not `keep ? act(..) : 0` per value -- that
    // form compiled to an exec-mask branch around every single SiLU (v_exp / v_rcp with their s_nops, no two of the 16
    // values of a fragment in flight together), and this kernel is bound by exactly that VALU work (round 5)
    const unsigned km = (p < S2_SPX && sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) ? 0xffffffffu : 0u;
    const int row = p < S2_SPX ? srow(py, px) : p;          // (the last fragment's lanes beyond the patch keep their own rows)
    float4_t bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *(const float4_t*)(bias_s + 8 * g + 4 * khalf);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)ctd_act_fast_rt(acc[4 * g + e] * oscale + bv[g][e], ACT0);
      uint2 u = __builtin_bit_cast(uint2, o);
      u.x &= km;
      u.y &= km;
      *(uint2*)(S + row * 32 + ((g ^ swz(row)) * 8) + 4 * khalf) = u;
    }
  }
  __syncthreads();   // the stem patch is complete; the input patch is dead
  // tiles 5-8: group 0 fetches 5, 7, group 1 fetches 6, 8
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if ((i & 1) == wg) dma_w(5 + i, lds + S2_W5 + i * S2_WTILE);

  // ---- layer 1: 3x3 / s2 over the stem patch; wave (w4, wg) = output rows 2 w4, 2 w4 + 1, N fragment wg -----------
  const int prow1 = 2 * w4 + (l31 >> 4), pcol1 = l31 < 16 ? l31 : ((l31 - 2) & 15);   // second row: columns rotated by 14
  const int pl = prow1 * S2_TW + pcol1;
  float16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto ld = [&](const half_t* base, int row, int kc) {
    return *(const half8_t*)(base + row * 32 + ((kc ^ swz(row)) * 8));
  };
  auto taps = [&](int t0, int t1) {
#pragma unroll
    for (int tap = t0; tap < t1; ++tap) {
      const int ty = tap / 3, tx = tap - 3 * ty;
      const int row = (2 * prow1 + ty) * S2_SW + (tx == 1 ? 17 + pcol1 : pcol1 + (tx >> 1));   // srow(2 prow1 + ty, 2 pcol1 + tx)
      const half_t* Wb = lds + S2_W0 + tap * S2_WTILE;    // taps 5-8 continue at S2_W5 = 5 tiles
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int kc = kk * 2 + khalf;
        const half8_t fw = ld(Wb, 32 * wg + l31, kc);
        const half8_t fx = ld(S, row, kc);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw, fx, acc, 0, 0, 0);
      }
    }
  };
  taps(0, 5);
  __syncthreads();   // waits this wave's DMAs of taps 5-8 (vmcnt 0), then all waves'
  taps(5, 9);
  __syncthreads();   // every wave is done reading the stem patch: the output tile may overwrite it
  half_t* Os = S;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4_t bv = *(const float4_t*)(bias_s + 32 + 32 * wg + 8 * g + 4 * khalf);
    half4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (half_t)ctd_act_fast<ACT1>(acc[4 * g + e] + bv[e]);
    *(half4_t*)(Os + pl * S2_OP + 32 * wg + 8 * g + 4 * khalf) = o;
  }
  __syncthreads();
  const int cch = t & 7;
#pragma unroll
  for (int it = 0; it < S2_TW * S2_TH / (S2_NT / 8); ++it) {
    const int p = it * (S2_NT / 8) + (t >> 3);
    const int oy = oy0 + (p >> 4), ox = ox0 + (p & 15);
    if (oy < Ho && ox < Wo)
      *(half8_t*)(a.dst + (((size_t)b * Ho + oy) * Wo + ox) * a.pitchD + cch * 8) = *(const half8_t*)(Os + p * S2_OP + cch * 8);
  }
}

}  // namespace

bool stem_conv2_supported(const Stem2Args& a) {
  if (!(g_fuse & 4)) return false;
  if (a.H % 4 || a.W % 4 || a.pitchD % 8) return false;
  if (a.act0 != CTD_ACT_SILU) return false;       // the yolo stem; other activations take the two launches
  if (a.act1 != CTD_ACT_SILU && a.act1 != CTD_ACT_LEAKY && a.act1 != CTD_ACT_RELU) return false;
  return true;
}

void launch_stem_conv2(const Stem2Args& a, hipStream_t st) {
  const int Ho = a.H / 4, Wo = a.W / 4;
  const dim3 grid((unsigned)(((Wo + S2_TW - 1) / S2_TW) * ((Ho + S2_TH - 1) / S2_TH) * a.B), 1, 1);
  switch (a.act1) {
    case CTD_ACT_SILU: hipLaunchKernelGGL((stem_conv2_kernel<CTD_ACT_SILU, CTD_ACT_SILU>), grid, dim3(S2_NT), 0, st, a); break;
    case CTD_ACT_LEAKY: hipLaunchKernelGGL((stem_conv2_kernel<CTD_ACT_SILU, CTD_ACT_LEAKY>), grid, dim3(S2_NT), 0, st, a); break;
    default: hipLaunchKernelGGL((stem_conv2_kernel<CTD_ACT_SILU, CTD_ACT_RELU>), grid, dim3(S2_NT), 0, st, a); break;
  }
}
