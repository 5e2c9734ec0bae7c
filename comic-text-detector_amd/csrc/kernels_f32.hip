// Exact-fp32 convolution on the matrix cores: implicit GEMM with `v_mfma_f32_32x32x2_f32` (f32 operands,
// f32 accumulate -- the same arithmetic as an fmaf chain, MI355X_MICROARCH.md: 157 TFLOP/s dense, 1/16 of
// the fp16 rate).  This is the engine of BASELINE configs[1] and of `TextDetector(half=False)`: the
// reference itself runs fp32 (reference inference.py:129, basemodel.py:222-244), so this is the precision
// in which "identical boxes / masks" is defined (tests/test_gpu_accept.py).
//
//   D[n][m] = sum_k W[n][k] * X[m][k]      n: output channel, m: output pixel, k = tap * Ctot + c
//
// Covers Conv k x k stride 1/2 (+ folded BN, activation, residual), two concatenated sources, nearest x2
// upsampled sources, and ConvTranspose 4x4/s2/p1 as four 2x2-tap phase GEMMs -- the ops of the fp32
// program whose channel counts are multiples of 16, plus the stem (the image is stored with a zero 4th
// channel so that one tap is one 16-B chunk; K = 36 taps x 4); the 2x2 transposed convs of the DB tail stay
// on the direct kernels (kernels_basic.hip).
//
// Tiling: 256 threads = 4 waves, block tile BN x 128 pixels, K step 16 floats (64-B LDS rows, XOR-swizzled
// 16-B chunks, double buffered, register-staged global loads).  The weights are the MFMA A operand, the
// pixels the B operand, so a lane's 16 accumulators are 4 x 4 consecutive channels of one pixel (16-B
// stores).  Inside a K step lanes 0-31 own k = 0..7 and lanes 32-63 k = 8..15 (two 16-B LDS reads per
// fragment feed eight MFMAs).  With 64 cycles per MFMA the loop has ample issue slack, so the gather
// addresses are simply recomputed per step.
#include <type_traits>

#include "kernels.h"

namespace {

constexpr int FBM = 128;   // pixels per block
constexpr int FBK = 16;    // floats per K step

template <int BN, int WGN, int WGM>
__global__ __launch_bounds__(256) void conv_f32_mfma_kernel(ConvArgs a) {
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  constexpr int TN = BN / (32 * WGN);
  constexpr int TM = FBM / (32 * WGM);
  static_assert(WGN * WGM == 4, "4 waves");
  constexpr int AROWS = FBM * 4 / 256;          // 16-B chunks of the pixel tile per thread (2)
  constexpr int WROWS = (BN * 4 + 255) / 256;   // chunks of the weight tile per thread
  __shared__ __attribute__((aligned(16))) float As[2][FBM * FBK];
  __shared__ __attribute__((aligned(16))) float Ws[2][BN * FBK];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wn = wave % WGN, wm = wave / WGN;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + FBM - 1) / FBM;
  int bid = blockIdx.x;
  const int tile_n = bid % ntn;
  bid /= ntn;
  int phase = 0;
  if (a.nphase == 4) {
    phase = bid & 3;
    bid >>= 2;
  }
  const int tile_m = bid;
  (void)ntm;
  const int n0 = tile_n * BN, m0 = tile_m * FBM;

  int dy0 = a.dy0, dx0 = a.dx0, ooy = a.ooy, oox = a.oox;
  const float* __restrict__ wbase = (const float*)a.w;
  if (a.nphase == 4) {   // ConvTranspose 4x4 s2 p1: sub-pixel phase (py, px), 2x2 taps
    const int py = phase >> 1, px = phase & 1;
    dy0 = py ? 0 : -1;
    dx0 = px ? 0 : -1;
    ooy = py;
    oox = px;
    wbase += (size_t)phase * a.w_phase_stride;
  }
  const int Ct = a.s0.c + a.s1.c;
  const int nk = a.K / FBK;
  auto swz = [](int row) { return (row >> 2) & 3; };

  // this thread's pixel rows (2 chunks: rows t/4 and t/4 + 64, chunk t%4)
  const int seg = t & 3;
  int pb[AROWS], poy[AROWS], pox[AROWS];
  bool pv[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + (t >> 2) + 64 * i;
    pv[i] = m < a.M;
    const int mm = pv[i] ? m : 0;
    pox[i] = mm % a.Mw;
    const int q = mm / a.Mw;
    poy[i] = q % a.Mh;
    pb[i] = q / a.Mh;
  }

  float4_t ra[AROWS], rw[WROWS];
  // (tap, channel offset) of the NEXT tile to load, advanced incrementally: one 32-bit division per step and row
  // was most of a K step's time on the narrow tiles (8 MFMAs per wave and step)
  int nx_cc = 0, nx_ty = 0, nx_tx = 0;
  auto load_tile = [&](int ks) {
    const int k0 = ks * FBK;
    // Ct % 16 == 0: a K step stays inside one tap.  Ct == 4 (the stem's zero-padded image): one tap per 16-B chunk.
    int cc, ty, tx;
    if (Ct == 4) {
      const int tap = ks * 4 + seg;
      cc = 0;
      ty = tap / a.KW, tx = tap - ty * a.KW;
    } else {
      cc = nx_cc, ty = nx_ty, tx = nx_tx;
      nx_cc += FBK;
      if (nx_cc == Ct) {
        nx_cc = 0;
        if (++nx_tx == a.KW) nx_tx = 0, ++nx_ty;
      }
    }
    const bool first = cc < a.s0.c;
    const SrcView& s = first ? a.s0 : a.s1;
    const int ch = Ct == 4 ? 0 : (first ? cc : cc - a.s0.c) + seg * 4;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int iy = poy[i] * a.stride + dy0 + ty, ix = pox[i] * a.stride + dx0 + tx;
      const bool ok = pv[i] && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      const int sy = s.up ? iy >> 1 : iy, sx = s.up ? ix >> 1 : ix;
      float4_t v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *(const float4_t*)((const float*)s.ptr + ((size_t)((size_t)pb[i] * s.H + sy) * s.W + sx) * s.pitch + ch);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int idx = t + 256 * i;
      if (BN * 4 >= 256 * (i + 1) || idx < BN * 4)
        rw[i] = *(const float4_t*)(wbase + (size_t)(n0 + (idx >> 2)) * a.K + k0 + (idx & 3) * 4);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int r = (t >> 2) + 64 * i;
      *(float4_t*)(&As[buf][r * FBK + ((seg ^ swz(r)) * 4)]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int idx = t + 256 * i;
      if (BN * 4 >= 256 * (i + 1) || idx < BN * 4) {
        const int r = idx >> 2;
        *(float4_t*)(&Ws[buf][r * FBK + (((idx & 3) ^ swz(r)) * 4)]) = rw[i];
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

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int fl = swz(l31);   // rows of one fragment differ by multiples of 32: same swizzle
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_tile(ks + 1);
    const float* Ab = &As[buf][(wm * TM * 32 + l31) * FBK];
    const float* Wb = &Ws[buf][(wn * TN * 32 + l31) * FBK];
    float4_t fw[TN][2], fx[TM][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int co = ((khalf * 2 + h) ^ fl) * 4;
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[i][h] = *(const float4_t*)(Wb + i * 32 * FBK + co);
#pragma unroll
      for (int j = 0; j < TM; ++j) fx[j][h] = *(const float4_t*)(Ab + j * 32 * FBK + co);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[i][h][e], fx[j][h][e], acc[i][j], 0, 0, 0);
    if (ks + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias + activation (+ residual) -> NHWC f32, 16 B per lane ----
  const int hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (wm * TM + j) * 32 + l31;
    if (m >= a.M) continue;
    const int ox = m % a.Mw, q = m / a.Mw, oy = q % a.Mh, b = q / a.Mh;
    const size_t opix = ((size_t)b * a.oH + (oy * a.osy + ooy)) * a.oW + (ox * a.osx + oox);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + (wn * TN + i) * 32 + 4 * hi + 8 * g;
        if (n >= a.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ctd_act_precise(acc[i][j][4 * g + e] + a.bias[n + e], a.act);   // bias padded to Npad
        if (n + 3 < a.N) {
          if (a.res) {
            const float4_t rv = *(const float4_t*)((const float*)a.res + opix * a.pitchR + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rv[e];
          }
          float4_t o = {v[0], v[1], v[2], v[3]};
          *(float4_t*)((float*)a.dst + opix * a.pitchD + n) = o;
        } else {
          for (int e = 0; e < 4 && n + e < a.N; ++e) {
            float r = v[e];
            if (a.res) r += ((const float*)a.res)[opix * a.pitchR + n + e];
            ((float*)a.dst)[opix * a.pitchD + n + e] = r;
          }
        }
      }
  }
}

template <int BN, int WGN, int WGM>
void launch_f32_cfg(const ConvArgs& a, hipStream_t st) {
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + FBM - 1) / FBM;
  hipLaunchKernelGGL((conv_f32_mfma_kernel<BN, WGN, WGM>), dim3((unsigned)(ntn * ntm * a.nphase)), dim3(256), 0, st, a);
}

}  // namespace

int f32_mfma_ntile(int N) { return N > 64 ? 128 : (N > 32 ? 64 : 32); }

// f32 sources / destination with 16-B aligned channel rows, channel counts multiples of 16
bool conv_f32_mfma_supported(const ConvArgs& a) {
  if (!(a.s0.c == 4 && a.s1.c == 0) && (a.s0.c % FBK || a.s1.c % FBK)) return false;
  if (a.s0.pitch % 4 || (a.s1.c && a.s1.pitch % 4) || (a.N >= 4 && a.pitchD % 4)) return false;
  if (a.res && a.pitchR % 4) return false;
  if (a.K % FBK) return false;
  return a.nphase == 1 || a.nphase == 4;
}

void launch_conv_f32_mfma(const ConvArgs& a, hipStream_t st) {
  int bn = f32_mfma_ntile(a.N);
  // Small maps (32x32 ... 128x128 at bs <= 8): a 128-channel tile leaves most of the 256 CUs idle while each
  // wave walks a K loop of up to 288 steps x 32 MFMAs of 64 cycles (0.22 ms per layer whatever its size, measured
  // at bs=1).  Narrower N tiles give 2-4x the blocks and a quarter of the MFMAs per wave and step; the weight
  // rows are plain [n][K], so any tile width that divides Npad reads the same packing.  (Tried on top of this
  // and measured slower: a second register set to prefetch two K steps ahead -- with 8 MFMAs per step the
  // loop is bound by its address arithmetic, not by the memory round trip.)
  const long long ntm = (a.M + FBM - 1) / FBM;
  while (bn > 32 && (a.Npad / bn) * ntm * a.nphase < 256 && a.Npad % (bn / 2) == 0) bn >>= 1;
  if (bn == 128) launch_f32_cfg<128, 2, 2>(a, st);
  else if (bn == 64) launch_f32_cfg<64, 1, 4>(a, st);
  else launch_f32_cfg<32, 1, 4>(a, st);
}
