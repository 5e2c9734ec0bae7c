// The fp32s engine's first layer, straight from the network input: Conv 6x6 / stride 2 / pad 2, 3 -> 32 channels
// (reference models/yolov5/common.py Conv at model.0, fed by inference.py:78 `astype(np.float32) / 255`), as split-operand
// MFMA work like kernels_split.hip (x = hi + lo in fp16, three v_mfma_f32_32x32x16_f16 per product).
//
// Why its own kernel: through the generic kernel this layer was an INPUT launch (uint8 page -> fp32 NHWC with a zero 4th
// channel, 537 MB per 32 pages written and read back) plus an implicit GEMM that fetches 36 taps x 16 B per output pixel
// through the vector-memory path and splits each value 36 times: 0.18 + 0.85 ms per 32 pages for a layer whose bytes
// (100 MB in, 1.07 GB out) take 0.25 ms.  Here a block owns 16x16 output pixels: the 36x36 input patch is read ONCE from
// the network input (either format), divided by 255, split, and kept in LDS as a hi and a lo plane of 8-B pixels; a lane's
// MFMA operand for K = 16 is two horizontally adjacent taps = 16 contiguous bytes of a plane.  K index = tap * 4 + channel
// (the engine's order for this layer: the packed weights of kernels_split.hip are used as they are), 36 taps padded to 40.
// The weights (20 KB) stay in registers.  Output through the shared store loop (fp32 or split-plane rows).
#include "kernels.h"
#include "split_epilogue.h"

namespace {

constexpr int ST = 16;                 // output tile edge
constexpr int SP = 2 * ST + 4;         // input patch edge: 36
constexpr int KK = 10;                 // K = 160 = 10 x 16 (40 taps x 4 channels)

// IN_U8: (B, H, W, 3) uint8; otherwise (B, 3, H, W) float
template <bool IN_U8>
__global__ __launch_bounds__(256) void stem_split_kernel(ConvArgs a, const void* __restrict__ in) {
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  // patch planes [SP * SP pixels][4 halves] (hi | lo), then the fp32 output tile [256][32]
  __shared__ __attribute__((aligned(16))) char lds[ST * ST * 32 * 4 + 256 * 4];
  static_assert(2 * SP * SP * 8 <= ST * ST * 32 * 4, "patch planes fit the output tile");
  half_t* Ph = (half_t*)lds;
  half_t* Pl = Ph + SP * SP * 4;
  half2_t* lut = (half2_t*)(lds + ST * ST * 32 * 4);     // uint8 v -> (hi, lo) of (float)v / 255.0f

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int tilesX = (a.Mw + ST - 1) / ST, tilesY = (a.Mh + ST - 1) / ST;
  int v = blockIdx.x;
  const int tpx = v % tilesX;
  v /= tilesX;
  const int tpy = v % tilesY;
  const int b = v / tilesY;
  const int y0 = tpy * ST, x0 = tpx * ST;
  const int H = a.Hin, W = a.Win;

  // ---- weights: lane (n = l31, khalf) holds W[n][16 kk + 8 khalf .. + 8] of both planes for all kk
  const half_t* __restrict__ wh = (const half_t*)a.w;
  const half_t* __restrict__ wl = (const half_t*)a.w2;
  half8_t fwh[KK], fwl[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int off = ((kk >> 1) * 32 + l31) * 32 + (kk & 1) * 16 + khalf * 8;   // [K / 32][32 rows][32 halves]
    fwh[kk] = *(const half8_t*)(wh + off);
    fwl[kk] = *(const half8_t*)(wl + off);
  }

  // ---- the input patch: value / 255 (uint8 pages) exactly like the INPUT op, split once per pixel.  All of a thread's
  // loads are issued before the first use (a rolled loop paid one memory round trip per pass: 6 per block), and the 256
  // possible uint8 values come from a table of (hi, lo) pairs built once per block -- no division, no conversion per pixel.
  constexpr int NPASS = (SP * SP + 255) / 256;
  if (IN_U8) {
    const float f = (float)t / 255.0f;
    const half_t h = (half_t)f;
    const half_t l = (half_t)(f - (float)h);
    half2_t e = {h, l};
    lut[t] = e;
    uint8_t px[NPASS][3];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int q = t + 256 * i;
      const int hy = q / SP, hx = q - hy * SP;
      const int iy = 2 * y0 - 2 + hy, ix = 2 * x0 - 2 + hx;
      const bool ok = q < SP * SP && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const uint8_t* s = (const uint8_t*)in + ((size_t)((size_t)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
      px[i][0] = ok ? s[0] : 0; px[i][1] = ok ? s[1] : 0; px[i][2] = ok ? s[2] : 0;     // 0 -> (0, 0): zero padding
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int q = t + 256 * i;
      if (q >= SP * SP) continue;
      const half2_t e0 = lut[px[i][0]], e1 = lut[px[i][1]], e2 = lut[px[i][2]];
      const half4_t hh = {e0[0], e1[0], e2[0], (half_t)0.f}, ll = {e0[1], e1[1], e2[1], (half_t)0.f};
      *(half4_t*)(Ph + q * 4) = hh;
      *(half4_t*)(Pl + q * 4) = ll;
    }
  } else {
    float c[NPASS][3];
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int q = t + 256 * i;
      const int hy = q / SP, hx = q - hy * SP;
      const int iy = 2 * y0 - 2 + hy, ix = 2 * x0 - 2 + hx;
      const bool ok = q < SP * SP && iy >= 0 && iy < H && ix >= 0 && ix < W;
      const float* s = (const float*)in + ((size_t)b * 3 * H + (ok ? iy : 0)) * W + (ok ? ix : 0);
      c[i][0] = ok ? s[0] : 0.f; c[i][1] = ok ? s[(size_t)H * W] : 0.f; c[i][2] = ok ? s[2 * (size_t)H * W] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int q = t + 256 * i;
      if (q >= SP * SP) continue;
      half4_t hh, ll;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        hh[e] = (half_t)c[i][e];
        ll[e] = (half_t)(c[i][e] - (float)hh[e]);
      }
      hh[3] = (half_t)0.f;
      ll[3] = (half_t)0.f;
      *(half4_t*)(Ph + q * 4) = hh;
      *(half4_t*)(Pl + q * 4) = ll;
    }
  }
  __syncthreads();

  // ---- wave w owns tile rows 4w .. 4w + 3 = two 32-pixel fragments (two rows each)
  float16_t acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int base[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int py = 4 * wave + 2 * j + (l31 >> 4), px = l31 & 15;
    base[j] = (2 * py) * SP + 2 * px;                    // patch pixel of tap (0, 0)
  }
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    int tp = kk * 4 + khalf * 2;                         // this lane's two taps: tp, tp + 1 (same row: 6 taps per row)
    if (tp > 34) tp = 34;                                // taps 36..39 pad K: zero weights, any finite operand
    const int ty = tp / 6, tx = tp - ty * 6;
    const int o = (ty * SP + tx) * 4;
    half8_t xh[2], xl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xh[j] = *(const half8_t*)(Ph + base[j] * 4 + o);
      xl[j] = *(const half8_t*)(Pl + base[j] * 4 + o);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[kk], xh[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[kk], xl[j], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[kk], xh[j], acc[j], 0, 0, 0);
  }
  __syncthreads();   // the patch is dead: the output tile takes its place

  float* stg = (float*)lds;
  const int hi = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int nl = 4 * hi + 8 * g;
    const float4_t os = *(const float4_t*)(a.oscale + nl), bs = *(const float4_t*)(a.bias + nl);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = (2 * wave + j) * 32 + l31;           // tile pixel: row 2 (2w + j) + l31 / 16, column l31 % 16
      float4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = acc[j][4 * g + e] * os[e] + bs[e];
      *(float4_t*)(stg + p * 32 + (split_stg_chunk<32>(p, nl >> 2) << 2)) = o;
    }
  }
  __syncthreads();
  split_store_tile<32, ST * ST, 256>(stg, a, 0, t, [&](int p, int, size_t& opix) {
    const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
    if (oy >= a.Mh || ox >= a.Mw) return false;
    opix = ((size_t)b * a.oH + oy) * a.oW + ox;
    return true;
  });
}

}  // namespace

// the first layer as the engine lowers it for fp32 tensors: 6x6 / s2 / p2 over the 4-channel (zero 4th) image, 32 outputs,
// weights packed by split_pack_weights with K = 144 padded to 160
bool stem_split_supported(const ConvArgs& a) {
  if (!a.w2 || !a.oscale || a.nphase != 1 || a.res) return false;
  if (a.s0.c != 4 || a.s1.c != 0 || a.s0.up || a.KH != 6 || a.KW != 6 || a.stride != 2 || a.dy0 != -2 || a.dx0 != -2) return false;
  if (a.N != 32 || a.Npad != 32 || a.K != 160) return false;
  if (a.Mh * 2 != a.Hin || a.Mw * 2 != a.Win || a.oH != a.Mh || a.oW != a.Mw) return false;
  if (a.pitchD % 4 || (a.d_sp && a.pitchD % 32)) return false;
  return true;
}

void launch_stem_split(const ConvArgs& a, const void* input, int in_fmt, hipStream_t st) {
  const int tiles = ((a.Mw + ST - 1) / ST) * ((a.Mh + ST - 1) / ST);
  const dim3 grid((unsigned)(tiles * a.B));
  if (in_fmt == CTD_IN_NHWC_U8) hipLaunchKernelGGL((stem_split_kernel<true>), grid, dim3(256), 0, st, a, input);
  else hipLaunchKernelGGL((stem_split_kernel<false>), grid, dim3(256), 0, st, a, input);
}
