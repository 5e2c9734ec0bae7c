// MFMA implicit-GEMM convolution for gfx950 (CDNA4), fp16 operands, fp32 accumulate.
//
//   D[n][m] = sum_k  Wp[n][k] * X[m][k]          n: output channel, m: output pixel,
//                                                k = (ty*KW + tx) * Ctot + c
//
// The WEIGHTS are the MFMA A operand (rows = n) and the gathered activations the
// B operand (cols = m), so each lane's accumulator registers hold 4 consecutive
// output channels of ONE pixel -> the NHWC epilogue stores 8 B (fp16) / 16 B (fp32)
// per lane instead of 2 B.
//
// One kernel covers (reference layer -> use):
//   * Conv 1x1 / 3x3, stride 1 / 2 + folded BN + SiLU/LeakyReLU/ReLU  (common.py:30-49)
//   * Bottleneck residual add after the activation                    (common.py:104)
//   * torch.cat of two producers and nn.Upsample(x2, nearest) folded into the K loop
//     (yolo cfg layers 11-12, 15-16, 19, 22; basemodel.py:66,71-73,108-109)
//   * ConvTranspose2d 4x4/s2/p1 + BN + ReLU as 4 sub-pixel phase GEMMs (blockIdx.z),
//     each a 2x2-tap conv over the input grid                         (basemodel.py:26-28)
//
// Tiling: 256 threads = 4 waves (64 lanes each).  Block tile BN x BM, K step 32.
// Global -> registers -> LDS staging, double buffered, one barrier per K step.
// LDS rows are 32 halves padded to 40 (80 B) so that both the 8-lane ds_write_b128
// groups and the 16-lane ds_read_b128 groups hit distinct 16-B bank slots.
// v_mfma_f32_32x32x16_f16: A/B fragment = 8 consecutive k of row/col (lane & 31),
// k group = lane >> 5; C/D: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#include "kernels.h"

namespace {

// K step BK = 32 or 64 halves per LDS row, padded by 8 halves (16 B): row pitch 80 B / 144 B.
// Both pitches map 16 consecutive rows (and the non-contiguous 16-lane groups of
// ds_read_b128) onto 16 distinct 16-B slots of the 256-B bank row -> conflict free.
template <int BN, int BM, int WGN, int WGM, int BK, bool DST_F32>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  constexpr int LP = BK + 8;                        // LDS row pitch in halves
  constexpr int SEGS = BK / 8;                      // 16-B chunks per row
  constexpr int RPP = 256 / SEGS;                   // rows staged per pass of the 256 threads
  constexpr int TN = BN / (32 * WGN);
  constexpr int TM = BM / (32 * WGM);
  constexpr int AROWS = BM / RPP;                   // pixel rows each thread stages
  constexpr int WCHUNKS = BN * SEGS;                // 16-B chunks of the weight tile
  constexpr int WROWS = (WCHUNKS + 255) / 256;
  static_assert(WGN * WGM == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) half_t lds[2 * (BM + BN) * LP];
  half_t* As = lds;                 // [2][BM][LP]  pixels
  half_t* Ws = lds + 2 * BM * LP;   // [2][BN][LP]  weights

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wn = wave % WGN, wm = wave / WGN;

  // ---- XCD-aware tile mapping: consecutive block ids land on different XCDs
  // (id % 8); give each XCD a contiguous run of tiles so that the N tiles of one
  // pixel tile (and neighbouring pixel tiles sharing 3x3 halos) share an L2.
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int nblk = ntn * ntm;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, within = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tile_n = bid % ntn;
  const int tile_m = bid / ntn;
  const int n0 = tile_n * BN;
  const int m0 = tile_m * BM;

  // ---- phase (ConvTranspose 4x4 s2 p1 sub-pixel decomposition) --------------
  const int phase = blockIdx.z;
  int dy0 = a.dy0, dx0 = a.dx0, ooy = a.ooy, oox = a.oox;
  const half_t* __restrict__ wbase = (const half_t*)a.w;
  if (a.nphase == 4) {
    const int py = phase >> 1, px = phase & 1;
    dy0 = py ? 0 : -1;
    dx0 = px ? 0 : -1;
    ooy = py;
    oox = px;
    wbase += (size_t)phase * a.w_phase_stride;
  }

  // ---- per-thread staging coordinates ----------------------------------------
  const int seg = t % SEGS;
  const int lrow = t / SEGS;  // 0..RPP-1
  int pb[AROWS], piy[AROWS], pix[AROWS];
  bool pvalid[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + lrow + RPP * i;
    pvalid[i] = m < a.M;
    const int mm = pvalid[i] ? m : 0;
    const int ox = mm % a.Mw;
    const int tq = mm / a.Mw;
    const int oy = tq % a.Mh;
    pb[i] = tq / a.Mh;
    piy[i] = oy * a.stride + dy0;
    pix[i] = ox * a.stride + dx0;
  }
  const int Ct = a.s0.c + a.s1.c;
  const int nk = a.K / BK;

  half8_t ra[AROWS];
  half8_t rw[WROWS];

  int cc = 0, ty = 0, tx = 0;  // K-step cursor of the NEXT tile to load

  auto load_tile = [&](int ks) {
    // activations
    const bool first = cc < a.s0.c;
    const SrcView& s = first ? a.s0 : a.s1;
    const int ch = (first ? cc : cc - a.s0.c) + seg * 8;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const int iy = piy[i] + ty, ix = pix[i] + tx;
      const bool ok = pvalid[i] && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) {
        const int sy = s.up ? (iy >> 1) : iy, sx = s.up ? (ix >> 1) : ix;
        const half_t* p = (const half_t*)s.ptr + ((size_t)((size_t)pb[i] * s.H + sy) * s.W + sx) * s.pitch + ch;
        v = *(const half8_t*)p;
      }
      ra[i] = v;
    }
    // weights
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int chunk = t + 256 * i;
      if (WCHUNKS >= 256 || chunk < WCHUNKS) {
        const int row = chunk / SEGS;
        rw[i] = *(const half8_t*)(wbase + (size_t)(n0 + row) * a.K + (size_t)ks * BK + seg * 8);
      }
    }
    // advance cursor
    cc += BK;
    if (cc == Ct) {
      cc = 0;
      if (++tx == a.KW) { tx = 0; ++ty; }
    }
  };

  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i)
      *(half8_t*)(As + ((size_t)buf * BM + lrow + RPP * i) * LP + seg * 8) = ra[i];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) {
      const int chunk = t + 256 * i;
      if (WCHUNKS >= 256 || chunk < WCHUNKS)
        *(half8_t*)(Ws + ((size_t)buf * BN + (chunk / SEGS)) * LP + seg * 8) = rw[i];
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

  const int l31 = lane & 31, kg = (lane >> 5) * 8;
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_tile(ks + 1);
    const half_t* Ab = As + (size_t)buf * BM * LP + (size_t)(wm * TM * 32 + l31) * LP + kg;
    const half_t* Wb = Ws + (size_t)buf * BN * LP + (size_t)(wn * TN * 32 + l31) * LP + kg;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      half8_t fw[TN], fx[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[i] = *(const half8_t*)(Wb + i * 32 * LP + kk * 16);
#pragma unroll
      for (int j = 0; j < TM; ++j) fx[j] = *(const half8_t*)(Ab + j * 32 * LP + kk * 16);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
    if (ks + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias + activation (+ residual) -> NHWC store -----------------
  const int hi = lane >> 5;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + (wm * TM + j) * 32 + l31;
    if (m >= a.M) continue;
    const int ox = m % a.Mw;
    const int tq = m / a.Mw;
    const int oy = tq % a.Mh;
    const int b = tq / a.Mh;
    const size_t opix = ((size_t)b * a.oH + (oy * a.osy + ooy)) * a.oW + (ox * a.osx + oox);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n0 + (wn * TN + i) * 32 + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g;
        if (n >= a.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ctd_act(acc[i][j][4 * g + e] + a.bias[n + e], a.act);
        if (a.res) {
          const half4_t rv = *(const half4_t*)((const half_t*)a.res + opix * a.pitchR + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        if (n + 3 < a.N) {
          if (DST_F32) {
            float4_t o = {v[0], v[1], v[2], v[3]};
            *(float4_t*)((float*)a.dst + opix * a.pitchD + n) = o;
          } else {
            half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)((half_t*)a.dst + opix * a.pitchD + n) = o;
          }
        } else {
          for (int e = 0; e < 4 && n + e < a.N; ++e) {
            if (DST_F32) ((float*)a.dst)[opix * a.pitchD + n + e] = v[e];
            else ((half_t*)a.dst)[opix * a.pitchD + n + e] = (half_t)v[e];
          }
        }
      }
    }
  }
}

template <int BN, int BM, int WGN, int WGM, int BK>
void launch_cfg(const ConvArgs& a, bool dst_f32, hipStream_t st) {
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + BM - 1) / BM;
  dim3 grid(ntn * ntm, 1, a.nphase);
  if (dst_f32)
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, true>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, false>), grid, dim3(256), 0, st, a);
}

// probe: one wave, D = A(32x16) * B(16x32) with the fragment convention used above
__global__ void mfma_probe_kernel(const half_t* __restrict__ A, const half_t* __restrict__ B, float* __restrict__ D) {
  const int lane = threadIdx.x;
  const int l31 = lane & 31, kg = (lane >> 5) * 8;
  half8_t fa, fb;
  for (int e = 0; e < 8; ++e) {
    fa[e] = A[l31 * 16 + kg + e];   // A[i][k] row-major 32x16
    fb[e] = B[(kg + e) * 32 + l31];  // B[k][j] row-major 16x32
  }
  float16_t acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    D[row * 32 + l31] = acc[r];
  }
}

}  // namespace

int igemm_ntile(int N) {
  if (N % 128 == 0) return 128;
  if (N % 64 == 0) return 64;
  return 32;
}

int g_igemm_force_bk = 0;  // selftest / tuning: 0 = heuristic, 32 or 64 = forced

static int pick_bk(const ConvArgs& a) {
  const bool can64 = (a.s0.c % 64 == 0) && (a.s1.c % 64 == 0);
  if (g_igemm_force_bk == 32 || !can64) return 32;
  if (g_igemm_force_bk == 64) return 64;
  return 64;
}

bool igemm_supported(const ConvArgs& a) {
  constexpr int BK = 32;
  if (a.s0.c % BK || a.s1.c % BK) return false;
  if (a.s0.pitch % 8 || (a.s1.c && a.s1.pitch % 8)) return false;
  if (a.pitchD % 4 || (a.res && a.pitchR % 4)) return false;
  if (a.K % BK) return false;
  return true;
}

void launch_conv_igemm(const ConvArgs& a, bool dst_f32, hipStream_t st) {
  const int bn = igemm_ntile(a.N);
  if (pick_bk(a) == 64) {
    if (bn == 128) launch_cfg<128, 128, 2, 2, 64>(a, dst_f32, st);
    else if (bn == 64) launch_cfg<64, 128, 2, 2, 64>(a, dst_f32, st);
    else launch_cfg<32, 128, 1, 4, 64>(a, dst_f32, st);
  } else {
    if (bn == 128) launch_cfg<128, 128, 2, 2, 32>(a, dst_f32, st);
    else if (bn == 64) launch_cfg<64, 128, 2, 2, 32>(a, dst_f32, st);
    else launch_cfg<32, 128, 1, 4, 32>(a, dst_f32, st);
  }
}

void launch_mfma_probe(const half_t* a, const half_t* b, float* out, hipStream_t st) {
  hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, st, a, b, out);
}
