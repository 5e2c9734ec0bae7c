// Pre/post resampling: cv2.resize(INTER_LINEAR) for uint8 with OpenCV's fixed-point
// arithmetic, fused with the letterbox padding (bottom/right zeros).
//   reference utils/imgproc_utils.py:113,116 (letterbox: resize + copyMakeBorder)
//   reference inference.py:165             (mask back to the page size)
// OpenCV (imgproc/resize.cpp, 8u linear): coefficient pairs are shorts round(c * 2048);
//   horizontal: I = S[x0]*a0 + S[x1]*a1
//   vertical  : dst = (((b0 * (I0 >> 4)) >> 16) + ((b1 * (I1 >> 4)) >> 16) + 2) >> 2
// with f = (float)((d + 0.5) * scale - 0.5); x fractions are zeroed outside [0, sw-1),
// y rows are clamped.  HBM-bound byte work: one lane per output pixel.
#include "kernels.h"

namespace {

struct LinCoef { int i0, i1, c0, c1; };

__device__ __forceinline__ LinCoef lin_coef(int d, double scale, int src, bool zero_frac_at_edges) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s0 = (int)floorf(f);
  f -= (float)s0;
  if (zero_frac_at_edges) {
    if (s0 < 0) { f = 0.f; s0 = 0; }
    if (s0 >= src - 1) { f = 0.f; s0 = src - 1; }
  }
  LinCoef c;
  c.c0 = __float2int_rn((1.f - f) * 2048.f);
  c.c1 = __float2int_rn(f * 2048.f);
  c.i0 = min(max(s0, 0), src - 1);
  c.i1 = min(max(s0 + 1, 0), src - 1);
  return c;
}

template <int C>
__global__ void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int sH, int sW, uint8_t* __restrict__ dst,
                                        int dH, int dW, int canvasH, int canvasW, double scale_x, double scale_y) {
  const long long total = (long long)canvasH * canvasW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % canvasW), y = (int)(i / canvasW);
    uint8_t out[C];
    if (x < dW && y < dH) {
      const LinCoef cx = lin_coef(x, scale_x, sW, true);
      const LinCoef cy = lin_coef(y, scale_y, sH, false);
      const uint8_t* r0 = src + (size_t)cy.i0 * sW * C;
      const uint8_t* r1 = src + (size_t)cy.i1 * sW * C;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int h0 = (int)r0[cx.i0 * C + c] * cx.c0 + (int)r0[cx.i1 * C + c] * cx.c1;
        const int h1 = (int)r1[cx.i0 * C + c] * cx.c0 + (int)r1[cx.i1 * C + c] * cx.c1;
        const int v = (((cy.c0 * (h0 >> 4)) >> 16) + ((cy.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out[c] = (uint8_t)min(max(v, 0), 255);
      }
    } else {
#pragma unroll
      for (int c = 0; c < C; ++c) out[c] = 0;      // letterbox border (copyMakeBorder value 0)
    }
#pragma unroll
    for (int c = 0; c < C; ++c) dst[i * C + c] = out[c];
  }
}

}  // namespace

void launch_resize_linear_u8(const uint8_t* src, int sH, int sW, int C, uint8_t* dst, int dH, int dW, int canvasH,
                             int canvasW, hipStream_t st) {
  const long long total = (long long)canvasH * canvasW;
  long long g = (total + 255) / 256;
  if (g > 256LL * 32) g = 256LL * 32;
  // OpenCV: inv_scale = dsize / ssize (double); scale = 1. / inv_scale
  const double scale_x = 1.0 / ((double)dW / sW), scale_y = 1.0 / ((double)dH / sH);
  if (C == 3)
    hipLaunchKernelGGL((resize_linear_u8_kernel<3>), dim3((int)g), dim3(256), 0, st, src, sH, sW, dst, dH, dW, canvasH,
                       canvasW, scale_x, scale_y);
  else
    hipLaunchKernelGGL((resize_linear_u8_kernel<1>), dim3((int)g), dim3(256), 0, st, src, sH, sW, dst, dH, dW, canvasH,
                       canvasW, scale_x, scale_y);
}
