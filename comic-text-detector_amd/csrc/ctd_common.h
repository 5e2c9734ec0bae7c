// Shared device/host declarations for the gfx950 comic-text-detector engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ctd_hip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int CTD_ZEROS_BYTES = 8192;

// One concatenated-source view: NHWC tensor slice.
struct SrcView {
  const void* ptr;  // already offset by the channel offset
  int pitch;        // elements per pixel of the underlying tensor
  int c;            // channels used
  int up;           // 1: tensor is at half the logical resolution (nearest x2)
  int H, W;         // physical spatial size of the tensor
};

// Arguments shared by every conv-like kernel (direct and implicit-GEMM).
struct ConvArgs {
  SrcView s0, s1;          // s1.c == 0 when unused
  int B;
  int Hin, Win;            // logical input size (after upsample)
  int Mh, Mw;              // output grid enumerated by M (conv: Hout,Wout; convT phase: Hin,Win)
  int KH, KW;              // taps walked by the K loop
  int stride;              // input step per output grid step
  int dy0, dx0;            // input y = oy*stride + dy0 + ty
  const void* w;           // packed weights (layout depends on kernel)
  const float* bias;       // padded to the N tile, never null for igemm
  void* dst;               // offset by channel offset
  int pitchD;
  int oH, oW;              // physical output size
  int osy, osx, ooy, oox;  // output pixel = (y*osy+ooy, x*osx+oox)
  const void* res;         // residual (same geometry as dst), may be null
  int pitchR;
  int act;
  int N;                   // true output channels
  int Npad;                // weight rows (multiple of the N tile)
  int K;                   // KH*KW*(s0.c+s1.c)
  int M;                   // B*Mh*Mw
  long long w_phase_stride;  // elements between phases (convT), else 0
  int nphase;              // 1, or 4 for convT 4x4 s2 (blockIdx.z)
  int bk;                  // igemm K step the weights were packed for (32 / 64)
  int w_tiled;             // igemm weights are tile-major [phase][n_tile][k_step][BN][bk]
  const void* zeros;       // CTD_ZEROS_BYTES of zeros in HBM: source of padding rows for LDS-DMA loads (kernels_halo2.hip walks
                           // a padding piece through it in step with the channel chunks: 2 B per input channel)
  long long* dbg;          // selftest only (k_rot & 16): per-block cycle stamps [nblk][8]; null in the product
  unsigned mw_mul, mw_sh;  // igemm: n / Mw == (uint64(n) * mw_mul) >> mw_sh for n < 2^31 (filled by the launcher)
  unsigned mh_mul, mh_sh;  //        same for Mh
  int k_rot;               // igemm: selftest ablation bits (0 in the product): 1 no K-loop loads, 2 no MFMAs, 4 no stores
  const void* w2;          // split kernel (kernels_split.hip): the lo plane of the weights (`w` is the hi plane)
  const float* oscale;     // split kernel: per output channel 1 / (power of two its weights were scaled by), padded to Npad
  int prio;                // 1: the kernel raises its waves' issue priority (s_setprio): the forward's kernels against a co-running tail
  // kernels_halo3.hip, 128-channel ConvTranspose phases only: a pointwise (1x1) conv that is this layer's ONLY consumer runs on
  // the block's output tile before it leaves the CU (post_w != null); this layer's own output is then never stored.
  //   out2 = act2(W2 . [x ; this layer's output] + b2),  x = post_x: an optional second source AHEAD of it in the concat
  const void* post_w;      // W2 in the implicit-GEMM packing [1][(post_x.c + 128) / 32][post_n][32] (post_n = 64 or 128)
  const float* post_bias;  // padded to post_n
  void* post_dst;          // (B, oH, oW, post_pitch) fp16, post_n channels written, already offset by the channel offset
  int post_pitch, post_n, post_act;
  SrcView post_x;          // c == 0 when unused; at the OUTPUT resolution of this layer, not upsampled
  int x_sp, d_sp, r_sp;    // split kernel: sources / destination / residual are SPLIT-PLANE tensors (kernels_split.hip): per pixel
                           // and 32-channel group 32 hi halves then 32 lo halves, in the 128 B the 32 floats would occupy
};

__device__ __forceinline__ float ctd_act(float v, int act) {
  switch (act) {
    case CTD_ACT_SILU: return v / (1.0f + __expf(-v));
    case CTD_ACT_LEAKY: return v > 0.f ? v : 0.1f * v;
    case CTD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CTD_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    default: return v;
  }
}
// compile-time activation for the MFMA epilogue: hardware rcp / exp2 (1 ulp, well inside fp16 rounding)
// The fp32 result is handed back as an OPAQUE value: otherwise the compiler fuses SiLU's last multiply with the fp16
// conversion of the store wherever its register pairing likes it (v_fma_mixlo_f16: the exact product rounded once to
// fp16) and keeps v_mul_f32 + v_cvt (rounded to fp32, then to fp16) elsewhere -- the same layer then rounds 0.006 % of
// its outputs differently from kernel to kernel (found by the fused-vs-unfused selftest).  Every fp16 activation is
// RTNE(fp32 activation) now, in every kernel.
template <int ACT> __device__ __forceinline__ float ctd_act_fast(float v) {
  if (ACT == CTD_ACT_SILU) {
    float r = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    asm("" : "+v"(r));
    return r;
  }
  // leaky / relu as ONE v_max (same value for every input, NaN and -0 included: 0.1 v > v exactly when v < 0) instead of a
  // compare + select: the select's vcc hazard costs an s_nop per value and the pair does not fold into packed code (round 5)
  if (ACT == CTD_ACT_LEAKY) return fmaxf(v, 0.1f * v);
  if (ACT == CTD_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == CTD_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  return v;
}
// run-time activation with the arithmetic of ctd_act_fast (the stem kernels: their SiLU with a true division was a
// dozen VALU instructions per element on 18 k elements per block)
__device__ __forceinline__ float ctd_act_fast_rt(float v, int act) {
  switch (act) {
    case CTD_ACT_SILU: return ctd_act_fast<CTD_ACT_SILU>(v);
    case CTD_ACT_LEAKY: return ctd_act_fast<CTD_ACT_LEAKY>(v);
    case CTD_ACT_RELU: return ctd_act_fast<CTD_ACT_RELU>(v);
    case CTD_ACT_SIGMOID: return ctd_act_fast<CTD_ACT_SIGMOID>(v);
    default: return v;
  }
}
// fp32-grade SiLU / sigmoid for the split-operand engine's epilogue (kernels_split.hip): ~2 ulp in a dozen instructions --
// exp2 of a two-piece product (v_exp_f32 is 1 ulp), reciprocal + one Newton step + a corrected quotient -- where expf and
// an IEEE division are ~35 with their range and scale handling.  The exponent is clamped so that 1 + e stays finite
// (SiLU of v < -87 is below 1e-36 either way).
__device__ __forceinline__ float ctd_exp_neg_f32(float v) {   // exp(-v), -v clamped to [-100, 87]
  const float x = fminf(fmaxf(-v, -100.f), 87.f);
  const float t = x * 1.44269504088896341f;
  const float r = fmaf(x, 1.44269504088896341f, -t) + x * 1.925963033500011e-08f;   // low part of x * log2(e)
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e * 0.6931471805599453f, r, e);                                        // e * 2^r
}
__device__ __forceinline__ float ctd_silu_f32(float v) {
  const float d = 1.0f + ctd_exp_neg_f32(v);
  float rc = __builtin_amdgcn_rcpf(d);
  rc = fmaf(fmaf(-d, rc, 1.0f), rc, rc);
  const float q = v * rc;
  return fmaf(fmaf(-d, q, v), rc, q);
}
__device__ __forceinline__ float ctd_sigmoid_f32(float v) {
  const float d = 1.0f + ctd_exp_neg_f32(v);
  const float rc = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, rc, 1.0f), rc, rc);
}
__device__ __forceinline__ float ctd_act_f32(float v, int act) {
  switch (act) {
    case CTD_ACT_SILU: return ctd_silu_f32(v);
    case CTD_ACT_LEAKY: return v > 0.f ? v : 0.1f * v;
    case CTD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CTD_ACT_SIGMOID: return ctd_sigmoid_f32(v);
    default: return v;
  }
}
// exact-ish variants for the fp32 parity mode (expf instead of the fast exp)
__device__ __forceinline__ float ctd_act_precise(float v, int act) {
  switch (act) {
    case CTD_ACT_SILU: return v / (1.0f + expf(-v));
    case CTD_ACT_LEAKY: return v > 0.f ? v : 0.1f * v;
    case CTD_ACT_RELU: return v > 0.f ? v : 0.f;
    case CTD_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f(float v) { return (T)v; }
