// Fused head/tail kernels of the fp16 fast path.  These layers have a tiny
// channel count on one side (3 in, or 1 out), so they are HBM/VALU work, not
// GEMMs: no MFMA here (north_star: "MFMA only ... where it is a true GEMM").
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------
// Stem: network input -> Conv 6x6 / s2 / p2 (3 -> 32) + folded BN + SiLU
// (reference yolo cfg layer 0; input conversion inference.py:77-82 fused).
//
// As a direct convolution this layer is VALU bound (3456 FMAs per pixel-lane, 0.78 ms per 32
// pages against 0.2 ms of HBM time), so it runs on the matrix cores like everything else:
// K = 6 rows x 24 (8 columns x 3 channels; columns 6, 7 carry zero weights) = 144 = 9 MFMA
// k-steps.  With the input patch in LDS as fp16 [row][col][channel], the 8 consecutive k of a
// lane's B fragment are 16 contiguous bytes of one patch row, so a fragment is four
// ds_read_b32.  The 32 x 144 weight matrix lives in registers as 9 A fragments per lane.
// Block = 8 x 32 output pixels (4 waves x 2 rows).  uint8 input is staged as the exact
// integers 0..255 with 128/255 folded into the packed weights (sums x 1/128); float input is rounded to fp16.
// ---------------------------------------------------------------------------
constexpr int SM_TW = 32, SM_TH = 8;          // output tile
constexpr int SM_PH = 2 * SM_TH + 4;          // patch rows (20)
constexpr int SM_PW = 2 * SM_TW + 4;          // patch columns (68)
constexpr int SM_PITCH = 72 * 3;              // LDS row pitch in halves: 72 columns (4 zero pad) x 3 channels
constexpr int SM_OP = 40;                     // output staging pitch (32 channels + 8) in halves

__global__ __launch_bounds__(256) void stem_mfma_kernel(const void* __restrict__ in, int in_fmt, half_t* __restrict__ dst,
                                                        int pitchD, int B, int H, int W,
                                                        const half_t* __restrict__ wfrag,
                                                        const float* __restrict__ bias, int act) {
  __shared__ __attribute__((aligned(16))) half_t patch[SM_PH * SM_PITCH + 8];
  __shared__ __attribute__((aligned(16))) half_t ostage[4][32 * SM_OP];
  const int Ho = H / 2, Wo = W / 2;
  const int tiles_x = (Wo + SM_TW - 1) / SM_TW, tiles_y = (Ho + SM_TH - 1) / SM_TH;
  int bid = blockIdx.x;
  const int tx0 = (bid % tiles_x) * SM_TW;
  bid /= tiles_x;
  const int ty0 = (bid % tiles_y) * SM_TH;
  const int b = bid / tiles_y;
  const int iy0 = 2 * ty0 - 2, ix0 = 2 * tx0 - 2;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  // A fragments: 9 k-steps x (32 channels x 16 k), one 16-B load each
  half8_t wf[9];
  const half_t* wsel = wfrag + (in_fmt == CTD_IN_NCHW_F32 ? 0 : 9 * 64 * 8);
#pragma unroll
  for (int s = 0; s < 9; ++s) wf[s] = *(const half8_t*)(wsel + ((size_t)s * 64 + lane) * 8);

  // ---- stage the input patch as fp16 [row][col][c].  All 16 loads of a thread are issued before
  // the first use (clamped addresses instead of branches): a load -> convert -> store loop pays
  // one memory round trip per iteration, 16 x ~2 us per block.
  constexpr int NEL = 3 * SM_PH * SM_PW;            // 4080
  constexpr int NIT = (NEL + 255) / 256;            // 16
  if (in_fmt == CTD_IN_NCHW_F32) {
    const float* src = (const float*)in + (size_t)b * 3 * H * W;
    float v[NIT];
    int dsti[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = t + 256 * k;
      const int c = i / (SM_PH * SM_PW);
      const int r = (i / SM_PW) % SM_PH;
      const int q = i % SM_PW;
      const int iy = iy0 + r, ix = ix0 + q;
      const bool ok = i < NEL && iy >= 0 && iy < H && ix >= 0 && ix < W;
      v[k] = src[ok ? ((size_t)c * H + iy) * W + ix : 0];
      if (!ok) v[k] = 0.f;
      dsti[k] = i < NEL ? r * SM_PITCH + q * 3 + c : -1;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (dsti[k] >= 0) patch[dsti[k]] = (half_t)v[k];
  } else {
    const uint8_t* src = (const uint8_t*)in + (size_t)b * H * W * 3;
    uint8_t v[NIT];
    int dsti[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = t + 256 * k;
      const int r = i / (SM_PW * 3), j = i % (SM_PW * 3);
      const int iy = iy0 + r, ix = ix0 + j / 3;
      const bool ok = i < NEL && iy >= 0 && iy < H && ix >= 0 && ix < W;
      v[k] = src[ok ? ((long long)iy * W + ix0) * 3 + j : 0];
      if (!ok) v[k] = 0;
      dsti[k] = i < NEL ? r * SM_PITCH + j : -1;
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (dsti[k] >= 0) patch[dsti[k]] = (half_t)(float)v[k];
  }
  for (int i = t; i < SM_PH * 12; i += 256) patch[(i / 12) * SM_PITCH + SM_PW * 3 + i % 12] = (half_t)0.f;   // pad columns
  if (t < 8) patch[SM_PH * SM_PITCH + t] = (half_t)0.f;
  __syncthreads();

  const int lx = lane & 31, kg = lane >> 5, hi = lane >> 5;
  const float oscale = in_fmt == CTD_IN_NCHW_F32 ? 1.0f : 1.0f / 128.0f;
  half_t* os = ostage[wave];
#pragma unroll 1
  for (int g = 0; g < 2; ++g) {
    const int ly = 2 * wave + g;
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const half_t* prow = patch + (2 * ly) * SM_PITCH + lx * 6;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      // k group G = 2s + kg: row ky = G / 3, first of 8 consecutive (col, channel) entries j0 = (G % 3) * 8
      const int offA = ((2 * s) / 3) * SM_PITCH + ((2 * s) % 3) * 8;
      const int offB = ((2 * s + 1) / 3) * SM_PITCH + ((2 * s + 1) % 3) * 8;
      const uint32_t* p = (const uint32_t*)(prow + (kg ? offB : offA));   // 4-B aligned
      union { uint32_t u[4]; half8_t h; } fx;
      fx.u[0] = p[0]; fx.u[1] = p[1]; fx.u[2] = p[2]; fx.u[3] = p[3];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s], fx.h, acc, 0, 0, 0);
    }
    // bias + activation; lane = pixel lx, channels 8*q + 4*hi + e
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4_t bv = *(const float4_t*)(bias + 8 * q + 4 * hi);
      half4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (half_t)ctd_act_fast_rt(acc[4 * q + e] * oscale + bv[e], act);
      *(half4_t*)(os + lx * SM_OP + 8 * q + 4 * hi) = o;
    }
    __builtin_amdgcn_wave_barrier();
    const int oy = ty0 + ly;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int qq = it * 64 + lane;
      const int px = qq >> 2, part = qq & 3;
      const int ox = tx0 + px;
      const half8_t v = *(const half8_t*)(os + px * SM_OP + part * 8);
      if (oy < Ho && ox < Wo) *(half8_t*)(dst + (((size_t)b * Ho + oy) * Wo + ox) * pitchD + part * 8) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// Seg final: ConvTranspose2d 4x4/s2/p1 (C -> 1, no bias) + Sigmoid
// (reference basemodel.py:58-61) fused with the f32 `mask` export and the u8
// quantisation of postprocess_mask (reference inference.py:96-99).
// out(2y+py, 2x+px) = sum_{dy,dx} in(y+dy, x+dx) . w[ky = py+1-2dy][kx = px+1-2dx]
// One lane per input pixel -> 2x2 outputs.  w: fp16 pairs packed [C/8][16 taps (ky*4+kx)][8].
// ---------------------------------------------------------------------------
// Block = 16x16 input pixels; the 18x18 halo tile (C fp16 channels per pixel, pixel pitch
// padded by 16 B so the 16-lane ds_read_b128 groups spread over the banks) is staged in LDS
// once, so every input byte is fetched from HBM/L2 once instead of 9 times.
constexpr int SF_T = 16;
template <int C>
__global__ __launch_bounds__(256) void seg_final_kernel(const half_t* __restrict__ src, int pitch, int B, int H, int W,
                                                        const float* __restrict__ w, float bias,
                                                        float* __restrict__ mask, uint8_t* __restrict__ mask_u8) {
  constexpr int TP = SF_T + 2;          // tile edge with halo
  constexpr int PP = C + 8;             // LDS pixel pitch in halves
  __shared__ __attribute__((aligned(16))) half_t tile[TP * TP * PP];
  const int tiles_x = (W + SF_T - 1) / SF_T, tiles_y = (H + SF_T - 1) / SF_T;
  int bid = blockIdx.x;
  const int x0 = (bid % tiles_x) * SF_T;
  bid /= tiles_x;
  const int y0 = (bid % tiles_y) * SF_T;
  const long long b = bid / tiles_y;
  constexpr int CH = C / 8;             // 16-B chunks per pixel
  // all loads of a thread are issued before the first LDS store (clamped addresses, no
  // branches): a load -> store loop pays one memory round trip per iteration
  constexpr int NCH = TP * TP * CH, NIT = (NCH + 255) / 256;
  {
    half8_t v[NIT];
    bool ok[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int c = i % CH, pp = i / CH;
      const int ty = pp / TP, tx = pp % TP;
      const int yy = y0 + ty - 1, xx = x0 + tx - 1;
      ok[k] = i < NCH && yy >= 0 && yy < H && xx >= 0 && xx < W;
      v[k] = *(const half8_t*)(src + (ok[k] ? ((b * H + yy) * W + xx) * pitch + c * 8 : 0));
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = threadIdx.x + 256 * k;
      const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (i < NCH) *(half8_t*)(tile + (i / CH) * PP + (i % CH) * 8) = ok[k] ? v[k] : z;
    }
  }
  __syncthreads();
  const int lx = threadIdx.x % SF_T, ly = threadIdx.x / SF_T;
  const int x = x0 + lx, y = y0 + ly;
  float o[2][2] = {{bias, bias}, {bias, bias}};
  // Weights: fp16 pairs packed [C/8][16 taps][4 dwords]; wave-uniform -> scalar loads.  The
  // channel-group loop is NOT unrolled: unrolling it makes the compiler hoist all 1024 scalar
  // weights and spill SGPRs through v_writelane/v_readlane (first version: 256 VGPRs, 1 wave/SIMD).
  const unsigned* __restrict__ wq = (const unsigned*)w;
#pragma unroll 1
  for (int c8 = 0; c8 < C / 8; ++c8) {
    const unsigned* wc = wq + c8 * 64;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const half8_t v = *(const half8_t*)(tile + ((ly + 1 + dy) * TP + (lx + 1 + dx)) * PP + c8 * 8);
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          const int ky = py + 1 - 2 * dy;
          if (ky < 0 || ky > 3) continue;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int kx = px + 1 - 2 * dx;
            if (kx < 0 || kx > 3) continue;
            float s = o[py][px];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned wb = wc[(ky * 4 + kx) * 4 + q];
              half2_t wv, xv = {v[2 * q], v[2 * q + 1]};
              __builtin_memcpy(&wv, &wb, 4);
              s = __builtin_amdgcn_fdot2(xv, wv, s, false);
            }
            o[py][px] = s;
          }
        }
      }
    }
  }
  if (x >= W || y >= H) return;
  const int Wo = 2 * W;
  const long long Ho = 2LL * H;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    const float s0 = 1.0f / (1.0f + __expf(-o[py][0]));
    const float s1 = 1.0f / (1.0f + __expf(-o[py][1]));
    const long long off = (b * Ho + (2 * y + py)) * Wo + 2 * x;
    if (mask) *(float2*)(mask + off) = make_float2(s0, s1);
    if (mask_u8) {
      uchar2 q;
      q.x = (uint8_t)(s0 * 255.0f);
      q.y = (uint8_t)(s1 * 255.0f);
      *(uchar2*)(mask_u8 + off) = q;
    }
  }
}

// The same layer for the exact-fp32 engine: f32 activations, f32 FMA, precise expf.  On the f32 MFMA kernel this
// 64 -> 1 ConvT filled 1/32 of an N tile (1.9 ms at bs=8, 8 % of the fp32 forward).  16-channel chunks of the
// 18x18 halo tile go through LDS (pixel pitch 20 floats: 16-B reads of consecutive pixels spread over the banks);
// weights (cin,1,4,4) stay in their checkpoint layout and are read with wave-uniform indices (scalar loads).
template <int C>
__global__ __launch_bounds__(256) void seg_final_f32_kernel(const float* __restrict__ src, int pitch, int B, int H, int W,
                                                            const float* __restrict__ w, float* __restrict__ mask,
                                                            uint8_t* __restrict__ mask_u8) {
  constexpr int TP = SF_T + 2, CC = 16, PP = CC + 4;
  __shared__ __attribute__((aligned(16))) float tile[TP * TP * PP];
  const int tiles_x = (W + SF_T - 1) / SF_T, tiles_y = (H + SF_T - 1) / SF_T;
  int bid = blockIdx.x;
  const int x0 = (bid % tiles_x) * SF_T;
  bid /= tiles_x;
  const int y0 = (bid % tiles_y) * SF_T;
  const long long b = bid / tiles_y;
  const int lx = threadIdx.x % SF_T, ly = threadIdx.x / SF_T;
  float o[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  constexpr int NCH = TP * TP * (CC / 4), NIT = (NCH + 255) / 256;
#pragma unroll 1
  for (int cc = 0; cc < C / CC; ++cc) {
    float4_t v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = threadIdx.x + 256 * k;
      const int c4 = i % (CC / 4), pp = i / (CC / 4);
      const int ty = pp / TP, tx = pp % TP;
      const int yy = y0 + ty - 1, xx = x0 + tx - 1;
      const bool ok = i < NCH && yy >= 0 && yy < H && xx >= 0 && xx < W;
      const float4_t z = {0.f, 0.f, 0.f, 0.f};
      v[k] = ok ? *(const float4_t*)(src + ((b * H + yy) * W + xx) * pitch + cc * CC + c4 * 4) : z;
    }
    __syncthreads();                              // the previous chunk's reads are done
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < NCH) *(float4_t*)(tile + (i / (CC / 4)) * PP + (i % (CC / 4)) * 4) = v[k];
    }
    __syncthreads();
#pragma unroll 1
    for (int c4 = 0; c4 < CC / 4; ++c4) {
      const float* wc = w + (size_t)(cc * CC + c4 * 4) * 16;        // [4 channels][ky*4+kx]
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          const float4_t xv = *(const float4_t*)(tile + ((ly + 1 + dy) * TP + (lx + 1 + dx)) * PP + c4 * 4);
#pragma unroll
          for (int py = 0; py < 2; ++py) {
            const int ky = py + 1 - 2 * dy;
            if (ky < 0 || ky > 3) continue;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
              const int kx = px + 1 - 2 * dx;
              if (kx < 0 || kx > 3) continue;
              float s = o[py][px];
#pragma unroll
              for (int e = 0; e < 4; ++e) s = fmaf(xv[e], wc[e * 16 + ky * 4 + kx], s);
              o[py][px] = s;
            }
          }
        }
      }
    }
  }
  const int x = x0 + lx, y = y0 + ly;
  if (x >= W || y >= H) return;
  const int Wo = 2 * W;
  const long long Ho = 2LL * H;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    const float s0 = 1.0f / (1.0f + expf(-o[py][0]));
    const float s1 = 1.0f / (1.0f + expf(-o[py][1]));
    const long long off = (b * Ho + (2 * y + py)) * Wo + 2 * x;
    if (mask) *(float2*)(mask + off) = make_float2(s0, s1);
    if (mask_u8) {
      uchar2 q;
      q.x = (uint8_t)(s0 * 255.0f);
      q.y = (uint8_t)(s1 * 255.0f);
      *(uchar2*)(mask_u8 + off) = q;
    }
  }
}

// ---------------------------------------------------------------------------
// DB tail: for each branch (binarize, thresh):
//   ConvT 2x2/s2 (q->q) + BN + ReLU -> ConvT 2x2/s2 (q->1) -> sigmoid
// (reference basemodel.py:99-102,113,138-142), non-overlapping so every input
// pixel expands independently to a 4x4 output patch.  Fused with the f32
// `lines_map` export and `binarize` pred > thresh (reference db_utils.py:71-72).
// params per branch (f32): W1[c][o][py][px] (q*q*4), b1[q], W2[o][0][py][px] (q*4), b2[1]
// ---------------------------------------------------------------------------
// Device parameter layout per branch (floats): W1p[pp = py*2+px][c][o] (4*Q*Q), b1[Q],
// W2p[qq = qy*2+qx][o] (4*Q), b2, padding to a multiple of 4.  The parameters live in LDS and
// are read with wave-uniform (broadcast) ds_read_b128; as scalar loads the fully unrolled
// kernel spilled >1000 SGPRs through v_writelane/v_readlane.
template <int Q>
struct DbUpLayout {
  static constexpr int W1 = 0, B1 = 4 * Q * Q, W2 = B1 + Q, B2 = W2 + 4 * Q, SIZE = (B2 + 1 + 3) / 4 * 4;
};

template <int Q, typename T>
__global__ __launch_bounds__(256) void db_up_kernel(const T* __restrict__ src, int pitch, int nbr, int B, int H, int W,
                                                    const float* __restrict__ params, float* __restrict__ lines,
                                                    uint8_t* __restrict__ bitmap, float thresh) {
  using Lt = DbUpLayout<Q>;
  __shared__ __attribute__((aligned(16))) float P[2 * Lt::SIZE];
  __shared__ float xs[Q * 256];     // this thread's input channels, [c][tid]: lets the c loop stay rolled
  for (int i = threadIdx.x; i < nbr * Lt::SIZE; i += 256) P[i] = params[i];
  __syncthreads();
  const long long total = (long long)B * H * W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int tid = threadIdx.x;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const long long b = i / ((long long)W * H);
  const T* p = src + i * pitch;
  constexpr bool F32IN = std::is_same<T, float>::value;    // the exact-fp32 engine: f32 activations, precise expf
  const int Wo = 4 * W;
  const long long Ho = 4LL * H;
  // Loops are deliberately NOT unrolled (except the 16-wide output vector): full unrolling made
  // the compiler hoist every parameter load (1000+ registers -> scratch, 1 wave/SIMD).
#pragma unroll 1
  for (int br = 0; br < nbr; ++br) {            // nbr = 1: shrink map only (the threshold branch is not lowered)
    const float* Pb = P + br * Lt::SIZE;
#pragma unroll
    for (int c8 = 0; c8 < Q / 8; ++c8) {
      if constexpr (F32IN) {
        const float4_t v0 = *(const float4_t*)(p + br * Q + c8 * 8), v1 = *(const float4_t*)(p + br * Q + c8 * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[(c8 * 8 + e) * 256 + tid] = v0[e], xs[(c8 * 8 + 4 + e) * 256 + tid] = v1[e];
      } else {
        const half8_t v = *(const half8_t*)(p + br * Q + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xs[(c8 * 8 + e) * 256 + tid] = (float)v[e];
      }
    }
#pragma unroll 1
    for (int pp = 0; pp < 4; ++pp) {
      float h[Q];
#pragma unroll
      for (int o4 = 0; o4 < Q / 4; ++o4) {
        const float4 t = *(const float4*)(Pb + Lt::B1 + o4 * 4);
        h[o4 * 4 + 0] = t.x; h[o4 * 4 + 1] = t.y; h[o4 * 4 + 2] = t.z; h[o4 * 4 + 3] = t.w;
      }
#pragma unroll 1
      for (int c = 0; c < Q; ++c) {
        const float xc = xs[c * 256 + tid];
#pragma unroll
        for (int o4 = 0; o4 < Q / 4; ++o4) {
          const float4 t = *(const float4*)(Pb + Lt::W1 + (pp * Q + c) * Q + o4 * 4);
          h[o4 * 4 + 0] = fmaf(xc, t.x, h[o4 * 4 + 0]);
          h[o4 * 4 + 1] = fmaf(xc, t.y, h[o4 * 4 + 1]);
          h[o4 * 4 + 2] = fmaf(xc, t.z, h[o4 * 4 + 2]);
          h[o4 * 4 + 3] = fmaf(xc, t.w, h[o4 * 4 + 3]);
        }
      }
#pragma unroll
      for (int o = 0; o < Q; ++o) h[o] = fmaxf(h[o], 0.f);   // stays fp32: closer to the fp32 reference
      const float b2 = Pb[Lt::B2];
      float r4[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float s = b2;
#pragma unroll
        for (int o4 = 0; o4 < Q / 4; ++o4) {
          const float4 t = *(const float4*)(Pb + Lt::W2 + qq * Q + o4 * 4);
          s = fmaf(h[o4 * 4 + 0], t.x, s);
          s = fmaf(h[o4 * 4 + 1], t.y, s);
          s = fmaf(h[o4 * 4 + 2], t.z, s);
          s = fmaf(h[o4 * 4 + 3], t.w, s);
        }
        r4[qq] = F32IN ? 1.0f / (1.0f + expf(-s)) : 1.0f / (1.0f + __expf(-s));
      }
      // sub-pixel (py,px) of the first ConvT owns output rows 4y+2py+{0,1}, columns 4x+2px+{0,1}
      const int py = pp >> 1, px = pp & 1;
#pragma unroll
      for (int qy = 0; qy < 2; ++qy) {
        const long long row = 4 * y + 2 * py + qy;
        const long long off = ((b * nbr + br) * Ho + row) * Wo + 4 * x + 2 * px;
        *(float2*)(lines + off) = make_float2(r4[qy * 2], r4[qy * 2 + 1]);
        if (br == 0 && bitmap) {
          uchar2 q;
          q.x = r4[qy * 2] > thresh;
          q.y = r4[qy * 2 + 1] > thresh;
          *(uchar2*)(bitmap + (b * Ho + row) * Wo + 4 * x + 2 * px) = q;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// DB tail on the matrix cores (fp16 engine).  db_up_kernel above is VALU work: 2560 fp32 FMAs per input pixel with
// broadcast LDS reads of the weights, 0.29 ms per 32 pages against a 0.08 ms byte floor.  Its first stage -- ConvT
// 2x2/s2, 16 -> 16 channels, four sub-pixel positions -- is a GEMM with K = 16: D[n = pp * 16 + o][pixel] =
// sum_c W1[pp][c][o] x[pixel][c], i.e. ONE v_mfma_f32_32x32x16_f16 per 32 pixels and 32 rows of n (two per branch).
// The B operand of a lane is 16 contiguous bytes of its pixel's channel row, loaded straight from HBM.  A lane ends
// with 8 of the 16 hidden channels of one pixel for two positions per fragment; it applies bias + ReLU, takes its
// half of the second ConvT's dot products (16 -> 1, four output positions each) and swaps halves with lane ^ 32,
// so that lane (pixel, hi) finishes output rows 4y + 2hi + {0, 1}: 4 contiguous floats per row = one 16-B store
// (the VALU kernel stored 8 B pieces).  W1 is rounded to fp16 like every other MFMA layer of this engine; the
// accumulation, the hidden activations and the second stage stay fp32.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void db_up_mfma_kernel(const half_t* __restrict__ src, int pitch, int nbr, int B, int H, int W,
                                                         const float* __restrict__ params, float* __restrict__ lines,
                                                         uint8_t* __restrict__ bitmap, float thresh, int ngroups) {
  using Lt = DbUpLayout<16>;
  constexpr int Q = 16;
  __shared__ __attribute__((aligned(16))) float P[2 * Lt::SIZE];
  for (int i = threadIdx.x; i < nbr * Lt::SIZE; i += 256) P[i] = params[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const long long total = (long long)B * H * W;
  const int Wo = 4 * W;
  const long long Ho = 4LL * H;
#pragma unroll 1
  for (int br = 0; br < nbr; ++br) {
    const float* Pb = P + br * Lt::SIZE;
    // A fragments: row n = f * 32 + l31 = pp * 16 + o, k = c = 8 * hi + e
    half8_t wa[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int n = f * 32 + l31, pp = n >> 4, o = n & 15;
#pragma unroll
      for (int e = 0; e < 8; ++e) wa[f][e] = (half_t)Pb[Lt::W1 + (pp * Q + 8 * hi + e) * Q + o];
    }
    // this lane's 8 hidden channels (accumulator register j of either half: o = (j & 3) + 8 * ((j >> 2) & 1) + 4 * hi)
    float b1r[8], w2r[4][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = (j & 3) + 8 * ((j >> 2) & 1) + 4 * hi;
      b1r[j] = Pb[Lt::B1 + o];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) w2r[qq][j] = Pb[Lt::W2 + qq * Q + o];
    }
    const float b2 = Pb[Lt::B2];
    for (int g = blockIdx.x * 4 + wave; g < ngroups; g += gridDim.x * 4) {
      const long long i = (long long)g * 32 + l31;
      const bool valid = i < total;
      const long long ic = valid ? i : 0;
      const int x = (int)(ic % W), y = (int)((ic / W) % H);
      const long long b = ic / ((long long)W * H);
      const half8_t xv = *(const half8_t*)(src + ic * pitch + br * Q + 8 * hi);
      float part[4][4];   // [pp][qq]: this lane's half of the sum over the hidden channels
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[f], xv, acc, 0, 0, 0);
        // register r = 8 ph + j: position pp = 2f + ph, hidden channel o(j)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float hv = fmaxf(acc[ph * 8 + j] + b1r[j], 0.f);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) s4[qq] = fmaf(hv, w2r[qq][j], s4[qq]);
          }
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) part[2 * f + ph][qq] = s4[qq];
        }
      }
      // lane hi finishes positions pp = 2 hi + {0, 1} (py = hi): it keeps its halves of those and receives the partner's
      float r8[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const float send = hi ? part[j][qq] : part[2 + j][qq];
          const float mine = hi ? part[2 + j][qq] : part[j][qq];
          const float tot = mine + __shfl_xor(send, 32) + b2;
          r8[j][qq] = 1.0f / (1.0f + __expf(-tot));
        }
      if (valid) {
        // position (py = hi, px = j) owns output rows 4y + 2hi + qy, columns 4x + 2j + qx (qq = qy * 2 + qx)
#pragma unroll
        for (int qy = 0; qy < 2; ++qy) {
          const long long row = 4LL * y + 2 * hi + qy;
          const long long off = ((b * nbr + br) * Ho + row) * Wo + 4 * x;
          float4_t o = {r8[0][qy * 2], r8[0][qy * 2 + 1], r8[1][qy * 2], r8[1][qy * 2 + 1]};
          *(float4_t*)(lines + off) = o;
          if (br == 0 && bitmap) {
            uchar4 q;
            q.x = o[0] > thresh; q.y = o[1] > thresh; q.z = o[2] > thresh; q.w = o[3] > thresh;
            *(uchar4*)(bitmap + (b * Ho + row) * Wo + 4 * x) = q;
          }
        }
      }
    }
  }
}

// The col2im + sigmoid + u8 quantiser of the 64 -> 1 ConvTranspose from the per-tap products of the haloed 18x18 tile in LDS
// (Ps[pixel][tap], out-of-image pixels zero): shared by seg_final_mfma_kernel (which computes the products itself) and
// seg_final_gather_kernel (which reads them from the producing ConvTranspose's launch) -- ONE summation order.
__device__ __forceinline__ void seg_final_gather(const float* Ps, int x0, int y0, long long b, int H, int W, float bias,
                                                 float* __restrict__ mask, uint8_t* __restrict__ mask_u8) {
  constexpr int TP = SF_T + 2;
  const int lx = threadIdx.x % SF_T, ly = threadIdx.x / SF_T;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= W || y >= H) return;
  float o[2][2] = {{bias, bias}, {bias, bias}};
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const float* Pn = Ps + ((ly + 1 + dy) * TP + (lx + 1 + dx)) * 16;
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        const int ky = py + 1 - 2 * dy;
        if (ky < 0 || ky > 3) continue;
#pragma unroll
        for (int px2 = 0; px2 < 2; ++px2) {
          const int kx = px2 + 1 - 2 * dx;
          if (kx < 0 || kx > 3) continue;
          o[py][px2] += Pn[ky * 4 + kx];
        }
      }
    }
  const int Wo = 2 * W;
  const long long Ho = 2LL * H;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    const float s0 = 1.0f / (1.0f + expf(-o[py][0]));
    const float s1 = 1.0f / (1.0f + expf(-o[py][1]));
    const long long off = (b * Ho + (2 * y + py)) * Wo + 2 * x;
    if (mask) *(float2*)(mask + off) = make_float2(s0, s1);
    if (mask_u8) {
      uchar2 q;
      q.x = (uint8_t)(s0 * 255.0f);
      q.y = (uint8_t)(s1 * 255.0f);
      *(uchar2*)(mask_u8 + off) = q;
    }
  }
}

// ---------------------------------------------------------------------------
// Seg final on the matrix cores (fp16 engine).  seg_final_kernel above spends 1024 FMAs per input pixel on a
// 64 -> 1 transposed convolution (0.36 ms per 32 pages against 0.23 ms of bytes).  Per input pixel the layer is a
// 16 x 64 matrix-vector product -- P[pixel][tap = ky * 4 + kx] = sum_c x[pixel][c] w[c][tap] -- followed by a
// col2im: out(2y + py, 2x + px) = sum over the 2x2 neighbours (dy, dx) of P[(y + dy, x + dx)][py + 1 - 2dy][px + 1 - 2dx].
// So: the P of the 18x18 haloed tile by MFMA (rows n = tap, 16 of the 32 used; B operand = 16 contiguous bytes of a
// pixel's channel row straight from HBM, the tile of x is never staged), P through LDS (22 KB instead of 47 KB), and
// 16 LDS reads + 16 adds per pixel for the gather.  Weights are the fp16 pairs the VALU kernel already uses.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void seg_final_mfma_kernel(const half_t* __restrict__ src, int pitch, int B, int H, int W,
                                                             const half_t* __restrict__ w, float bias,
                                                             float* __restrict__ mask, uint8_t* __restrict__ mask_u8) {
  constexpr int C = 64, TP = SF_T + 2, NPX = TP * TP, NFRAG = (NPX + 31) / 32;   // 324 haloed pixels, 11 fragments
  __shared__ __attribute__((aligned(16))) float Ps[NFRAG * 32 * 16];
  const int tiles_x = (W + SF_T - 1) / SF_T, tiles_y = (H + SF_T - 1) / SF_T;
  int bid = blockIdx.x;
  const int x0 = (bid % tiles_x) * SF_T;
  bid /= tiles_x;
  const int y0 = (bid % tiles_y) * SF_T;
  const long long b = bid / tiles_y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  // A fragments: row n = tap (l31 < 16), k = channel 16 ks + 8 hi + e: one 16-B piece of w [C/8][16 taps][8]
  half8_t wa[C / 16];
#pragma unroll
  for (int ks = 0; ks < C / 16; ++ks) {
    const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    wa[ks] = l31 < 16 ? *(const half8_t*)(w + ((size_t)(2 * ks + hi) * 16 + l31) * 8) : z;
  }
  for (int f = wave; f < NFRAG; f += 4) {
    const int p = 32 * f + l31;
    const int ty = p / TP, tx = p - ty * TP;
    const int yy = y0 + ty - 1, xx = x0 + tx - 1;
    const bool ok = p < NPX && yy >= 0 && yy < H && xx >= 0 && xx < W;
    const half_t* px = src + (ok ? ((b * H + yy) * W + xx) * pitch : 0) + 8 * hi;
    half8_t xv[C / 16];
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) xv[ks] = *(const half8_t*)(px + 16 * ks);     // all loads before the first use
    float16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) {
      const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks], ok ? xv[ks] : z, acc, 0, 0, 0);
    }
    // rows n = (r & 3) + 8 (r >> 2) + 4 hi: r = 0..3 -> taps 4hi .. 4hi+3, r = 4..7 -> taps 8 + 4hi .. (r >= 8: the unused rows)
    const float4_t lo = {acc[0], acc[1], acc[2], acc[3]}, hi4 = {acc[4], acc[5], acc[6], acc[7]};
    *(float4_t*)(Ps + p * 16 + 4 * hi) = lo;
    *(float4_t*)(Ps + p * 16 + 8 + 4 * hi) = hi4;
  }
  __syncthreads();
  seg_final_gather(Ps, x0, y0, b, H, W, bias, mask, mask_u8);
}

// The same layer when the producing ConvTranspose (128 -> 64, kernels_halo3.hip SEGP) has already multiplied its output
// tile by the 16 taps: P (B, H, W, 16) fp32 holds, per pixel of the 64-channel map that is never stored, the per-tap
// products seg_final_mfma_kernel computes for itself -- by the same MFMA sequence on the same fp16 values, so the mask is
// bit-identical -- at half the bytes of the map (64 B instead of 128 B per pixel, written once, read once).
__global__ __launch_bounds__(256) void seg_final_gather_kernel(const float* __restrict__ P, int B, int H, int W, float bias,
                                                               float* __restrict__ mask, uint8_t* __restrict__ mask_u8) {
  constexpr int TP = SF_T + 2, NPX = TP * TP;
  __shared__ __attribute__((aligned(16))) float Ps[NPX * 16];
  const int tiles_x = (W + SF_T - 1) / SF_T, tiles_y = (H + SF_T - 1) / SF_T;
  int bid = blockIdx.x;
  const int x0 = (bid % tiles_x) * SF_T;
  bid /= tiles_x;
  const int y0 = (bid % tiles_y) * SF_T;
  const long long b = bid / tiles_y;
  for (int i = threadIdx.x; i < NPX * 4; i += 256) {
    const int p = i >> 2, q = i & 3;
    const int ty = p / TP, tx = p - ty * TP;
    const int yy = y0 + ty - 1, xx = x0 + tx - 1;
    const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
    float4_t v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *(const float4_t*)(P + (((size_t)b * H + yy) * W + xx) * 16 + q * 4);
    *(float4_t*)(Ps + p * 16 + q * 4) = v;
  }
  __syncthreads();
  seg_final_gather(Ps, x0, y0, b, H, W, bias, mask, mask_u8);
}

}  // namespace

void launch_seg_final_gather(const float* P, int B, int H, int W, float bias, float* mask, uint8_t* mask_u8, hipStream_t st) {
  const int g = ((W + SF_T - 1) / SF_T) * ((H + SF_T - 1) / SF_T) * B;
  hipLaunchKernelGGL(seg_final_gather_kernel, dim3(g), dim3(256), 0, st, P, B, H, W, bias, mask, mask_u8);
}

void launch_stem(const void* in, int in_fmt, half_t* dst, int pitchD, int B, int H, int W, int N, const half_t* wfrag,
                 const float* bias, int act, hipStream_t st) {
  const int Ho = H / 2, Wo = W / 2;
  const int tiles = ((Wo + SM_TW - 1) / SM_TW) * ((Ho + SM_TH - 1) / SM_TH) * B;
  // N == 32 only (YOLOv5s); the engine falls back to INPUT + direct conv otherwise
  hipLaunchKernelGGL(stem_mfma_kernel, dim3(tiles), dim3(256), 0, st, in, in_fmt, dst, pitchD, B, H, W, wfrag, bias, act);
  (void)N;
}

// MFMA A fragments of the stem weights: [variant: float input, uint8 input (x 128/255, the kernel
// multiplies the sums by 1/128: keeps the scaled weights out of the fp16 subnormals)][9 k-steps][64 lanes][8];
// lane = (channel n = lane & 31, k group = lane >> 5), k = 16 s + 8 kgroup + e = 24 ky + 3 kx + c (kx < 6).
void stem_pack_weights(const float* W /* (32, 3, 6, 6) */, std::vector<half_t>& out) {
  out.assign((size_t)2 * 9 * 64 * 8, (half_t)0.f);
  for (int var = 0; var < 2; ++var)
    for (int s = 0; s < 9; ++s)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int n = lane & 31, k = 16 * s + 8 * (lane >> 5) + e;
          const int ky = k / 24, j = k % 24;
          if (j >= 18) continue;
          const int kx = j / 3, c = j % 3;
          const float w = W[(((size_t)n * 3 + c) * 6 + ky) * 6 + kx] * (var ? 128.0f / 255.0f : 1.0f);
          out[(((size_t)var * 9 + s) * 64 + lane) * 8 + e] = (half_t)w;
        }
}

void launch_seg_final(const half_t* src, int pitch, int C, int B, int H, int W, const float* w, float bias,
                      float* mask, uint8_t* mask_u8, hipStream_t st) {
  const int g = ((W + SF_T - 1) / SF_T) * ((H + SF_T - 1) / SF_T) * B;
  if (g_seg_final_mfma && C == 64 && pitch % 8 == 0 && (((uintptr_t)src | (uintptr_t)w) & 15) == 0)
    hipLaunchKernelGGL(seg_final_mfma_kernel, dim3(g), dim3(256), 0, st, src, pitch, B, H, W, (const half_t*)w, bias, mask, mask_u8);
  else
    hipLaunchKernelGGL((seg_final_kernel<64>), dim3(g), dim3(256), 0, st, src, pitch, B, H, W, w, bias, mask, mask_u8);
  (void)C;
}

void launch_seg_final_f32(const float* src, int pitch, int C, int B, int H, int W, const float* w, float* mask,
                          uint8_t* mask_u8, hipStream_t st) {
  const int g = ((W + SF_T - 1) / SF_T) * ((H + SF_T - 1) / SF_T) * B;
  hipLaunchKernelGGL((seg_final_f32_kernel<64>), dim3(g), dim3(256), 0, st, src, pitch, B, H, W, w, mask, mask_u8);
  (void)C;
}

int g_seg_final_mfma = 1;   // ctd_tuning_set("seg_final_mfma", 0): the VALU kernel (the fallback for odd pitches; A/B reference)
int g_db_up_mfma = 1;   // ctd_tuning_set("db_up_mfma", 0): the VALU kernel (the fallback for odd pitches; A/B reference)

void launch_db_up(const void* src, bool f32in, int pitch, int q, int nbr, int B, int H, int W, const float* params, float* lines,
                  uint8_t* bitmap, float thresh, hipStream_t st) {
  const long long total = (long long)B * H * W;
  const int g = (int)((total + 255) / 256);
  if (f32in)
    hipLaunchKernelGGL((db_up_kernel<16, float>), dim3(g), dim3(256), 0, st, (const float*)src, pitch, nbr, B, H, W, params,
                       lines, bitmap, thresh);
  else if (g_db_up_mfma && pitch % 8 == 0 && (((uintptr_t)src | (uintptr_t)lines) & 15) == 0) {
    const int ngroups = (int)((total + 31) / 32);
    const int blocks = (int)std::min<long long>((ngroups + 3) / 4, 256 * 16);
    hipLaunchKernelGGL(db_up_mfma_kernel, dim3(blocks), dim3(256), 0, st, (const half_t*)src, pitch, nbr, B, H, W, params, lines,
                       bitmap, thresh, ngroups);
  } else
    hipLaunchKernelGGL((db_up_kernel<16, half_t>), dim3(g), dim3(256), 0, st, (const half_t*)src, pitch, nbr, B, H, W, params,
                       lines, bitmap, thresh);
  (void)q;
}
