// The split-operand kernels' store loop (kernels_split.hip, kernels_split_halo.hip): a block's output tile sits in LDS as
// fp32 [pixels][BN] (16-B chunks XOR-swizzled by the pixel, written from the MFMA accumulators after oscale + bias) and
// every thread takes 8 consecutive channels of one pixel per turn: activation, residual, then either the fp32 row or
// the split-plane row (32 hi halves | 32 lo halves per 32-channel group) -- coalesced 16-B accesses, 8 lanes per 128-B
// line, and a few hundred instructions where the per-register-group epilogue was 8 000.
#pragma once
#include <type_traits>

#include "ctd_common.h"

// chunk position of 16-B chunk `c` of tile row `p`
template <int BN> __device__ __forceinline__ int split_stg_chunk(int p, int c) { return c ^ (p & (BN / 4 - 1)); }

// pix(p, px, opix) -> bool: is tile pixel p inside the output, and which output pixel (index into [B][oH][oW]) is it.
// PAIR (kernels_split_halo.hip): tile columns 64-127 are channels 0-63 of the ConvT phase px = 1 (px = 0 otherwise).
template <int BN, int NPIX, int NTHR, bool PAIR = false, typename PixFn>
__device__ __forceinline__ void split_store_tile(const float* stg, const ConvArgs& a, int n0, int t, PixFn pix) {
  constexpr int UR = BN / 8;                           // 8-channel units per tile row
  auto body = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    for (int u = t; u < NPIX * UR; u += NTHR) {
      const int p = u / UR, cu = u % UR;
      const int n = PAIR ? (cu & (UR / 2 - 1)) * 8 : n0 + cu * 8;
      size_t opix;
      if (n >= a.N || !pix(p, PAIR ? cu / (UR / 2) : 0, opix)) continue;
      const float4_t v0 = *(const float4_t*)(stg + p * BN + (split_stg_chunk<BN>(p, 2 * cu) << 2));
      const float4_t v1 = *(const float4_t*)(stg + p * BN + (split_stg_chunk<BN>(p, 2 * cu + 1) << 2));
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = ACT == CTD_ACT_SILU ? ctd_silu_f32(v0[e]) : ctd_act_f32(v0[e], a.act);
        v[4 + e] = ACT == CTD_ACT_SILU ? ctd_silu_f32(v1[e]) : ctd_act_f32(v1[e], a.act);
      }
      if (n + 7 < a.N) {
        if (a.res) {
          if (a.r_sp) {       // split-plane residual: 8 hi halves + 8 lo halves; hi + lo is exact in fp32
            const char* gp = (const char*)((const float*)a.res + opix * a.pitchR + (n & ~31)) + (n & 31) * 2;
            const half8_t rh = *(const half8_t*)gp, rl = *(const half8_t*)(gp + 64);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rh[e] + (float)rl[e];
          } else {
            const float* rp = (const float*)a.res + opix * a.pitchR + n;
            const float4_t r0 = *(const float4_t*)rp, r1 = *(const float4_t*)(rp + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r0[e], v[4 + e] += r1[e];
          }
        }
        if (a.d_sp) {         // split once here, for every consumer, tap and N tile that will read it
          half8_t oh, ol;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            oh[e] = (half_t)v[e];
            ol[e] = (half_t)(v[e] - (float)oh[e]);
          }
          char* gp = (char*)((float*)a.dst + opix * a.pitchD + (n & ~31)) + (n & 31) * 2;
          *(half8_t*)gp = oh;
          *(half8_t*)(gp + 64) = ol;
        } else {
          float* dp = (float*)a.dst + opix * a.pitchD + n;
          const float4_t o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
          *(float4_t*)dp = o0;
          *(float4_t*)(dp + 4) = o1;
        }
      } else {                // fp32 tensors only (conv_split_supported): the last unit of a channel count like 21
        for (int e = 0; e < 8 && n + e < a.N; ++e) {
          float r = v[e];
          if (a.res) r += ((const float*)a.res)[opix * a.pitchR + n + e];
          ((float*)a.dst)[opix * a.pitchD + n + e] = r;
        }
      }
    }
  };
  if (a.act == CTD_ACT_SILU) body(std::integral_constant<int, CTD_ACT_SILU>{});
  else body(std::integral_constant<int, -1>{});
}
