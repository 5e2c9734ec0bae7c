// Halo-tile split-operand convolution for gfx950: the fp32s engine's stride-1 3x3 convolutions and the four 2x2-tap
// phases of ConvTranspose 4x4/s2/p1 on SPLIT-PLANE tensors (kernels_split.hip: x = hi + lo in fp16, three
// v_mfma_f32_32x32x16_f16 per product; reference: Conv / C3 bottlenecks of models/yolov5/common.py and the
// ConvTranspose2d of basemodel.py double_conv_up_c3, fp32 there: inference.py:129).
//
// Why: the 128-pixel kernel of kernels_split.hip moves a 16-KB pixel tile for EVERY tap plus a 16-KB weight tile through
// L2 -> LDS per 24 MFMAs of a wave -- 17 GB per launch of ConvT 256 -> 128 (whose tensors are 1.6 GB), at the 9 TB/s that
// path gives: matrix cores 40 % busy.  Here, as in kernels_halo.hip for the fp16 engine,
//   * a block owns a 16x16 pixel patch and stages its haloed input patch (hi and lo planes) in LDS ONCE per
//     32-channel chunk; all taps read it at shifted LDS rows;
//   * the patch is 256 pixels (8 waves), so a weight tile is amortised over twice the pixels:
// 27 KB per 256-pixel tap step instead of 64 KB (ConvT, 128 output channels).  118 KB of LDS, one block per CU.
//
// Sources must be split-plane (the planner stores conv-to-conv tensors that way), not upsampled, channel counts multiples
// of 32; destination fp32 or split-plane; weights in kernels_split.hip's packing (hi plane | lo plane,
// [phase][Npad / 32][K / 32][32][32], K index = tap * Ctot + channel), shared with the 128-pixel kernel.
#include <type_traits>

#include "kernels.h"
#include "split_epilogue.h"

namespace {

constexpr int TWP = 16;               // pixel patch width; the height THP is 16 (256 pixels, 8 waves) or 8 (128 pixels, 4 waves)
constexpr int BKH = 32;               // channels per K chunk (64-B LDS rows per plane)

// PAIR (ConvT 4x4/s2 with 64 output channels, BN = 128): a block computes the two sub-pixel phases (py, px = 0) and
// (py, px = 1) of its patch -- N columns 0-63 are phase px = 0, 64-127 phase px = 1.  The two phases read the same input
// rows and overlapping columns (dx in {-1,0} and {0,+1}), so ONE 17x18 patch serves both, and the wave tile is the
// 2x2-fragment tile of the 128-channel layers (as in kernels_halo.hip).
// THP = 8 (selftest build only, measured equal to the implicit-GEMM kernel): a 16x8 patch, 256 threads, 64 KB of LDS at
// BN = 64 -- two blocks per CU for the 64-channel 3x3 layers.
template <int BN, int WGN, int WGM, bool PAIR = false, int THP = 16>
__global__ __launch_bounds__(32 * THP, 2) void conv_split_halo_kernel(ConvArgs a) {
  constexpr int BMH = TWP * THP;                  // pixels per block
  constexpr int NTHR = 2 * BMH;                   // 512 / 256 threads
  constexpr int AROWS_PAD = THP == 16 ? 336 : 192;   // 18 x (THP + 2) haloed rows, rounded up to whole waves of the third DMA pass
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  constexpr int TN = BN / (32 * WGN);
  constexpr int TM = BMH / (32 * WGM);
  static_assert(WGN * WGM == NTHR / 64, "one wave per 32 x 32-pixel-by-channel fragment group");
  constexpr int A_BUF = AROWS_PAD * BKH;          // halves of one plane of one patch buffer
  constexpr int W_TILE = BN * BKH;                // halves of one plane of one tap's weight tile
  constexpr int LDS_STAGE = 4 * A_BUF + 4 * W_TILE;            // [2 buffers][hi, lo] each
  constexpr int LDS_OUT = BMH * BN * 2;                        // the fp32 output tile, in halves
  constexpr int LDS_MAIN = LDS_STAGE > LDS_OUT ? LDS_STAGE : LDS_OUT;
  __shared__ __attribute__((aligned(16))) half_t lds[LDS_MAIN];   // one LDS object (see kernels_igemm.hip)
  half_t* As = lds;                 // [2][2][AROWS_PAD][32]
  half_t* Ws = lds + 4 * A_BUF;     // [2][2][BN][32]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave_u % WGN, wm = wave_u / WGN;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- block -> (batch, patch, phase, N tile); XCD-aware: each XCD gets a contiguous run so the N tiles / phases /
  // neighbouring patches that share input pixels share an L2
  const int ntn = PAIR ? 1 : a.Npad / BN;
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + THP - 1) / THP;
  const int nblk = ntn * (PAIR ? 2 : a.nphase) * tilesX * tilesY * a.B;
  int v = blockIdx.x;
  {
    const int xcd = v & 7, within = v >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tile_n = v % ntn;
  v /= ntn;
  int phase = 0;
  if (PAIR) {
    phase = (v & 1) * 2;      // py; the block covers px = 0 and 1
    v >>= 1;
  } else if (a.nphase == 4) {
    phase = v & 3;
    v >>= 2;
  }
  const int tpx = v % tilesX;
  v /= tilesX;
  const int tpy = v % tilesY;
  const int b = v / tilesY;
  const int n0 = tile_n * BN;
  const int y0 = tpy * THP, x0 = tpx * TWP;

  int dy0 = a.dy0, dx0 = a.dx0, ooy = a.ooy, oox = a.oox;
  const half_t* __restrict__ wh = (const half_t*)a.w;
  const half_t* __restrict__ wl = (const half_t*)a.w2;
  if (a.nphase == 4) {   // ConvTranspose 4x4 s2 p1: sub-pixel phase (py, px), 2x2 taps
    const int py = phase >> 1, px = phase & 1;
    dy0 = py ? 0 : -1;
    dx0 = px ? 0 : -1;
    ooy = py;
    oox = px;
    wh += (size_t)phase * a.w_phase_stride;   // PAIR: phase (py, 0); (py, 1) is one phase stride further (woff below)
    wl += (size_t)phase * a.w_phase_stride;
  }
  if (PAIR) dx0 = -1;                                    // the patch spans dx = -1 .. +1
  const int HW = TWP + (PAIR ? 3 : a.KW) - 1, HH = THP + a.KH - 1;   // haloed patch
  const int taps = a.KH * a.KW;
  const int Ct = a.s0.c + a.s1.c;
  const int nchunk = Ct / BKH;
  const int nk = a.K / BKH;                             // K steps of the weight packing (= taps * nchunk)

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  // LDS rows are 64 B, 16-B chunk c of row r sits at chunk position c ^ ((r >> 2) & 3).  The DMA writes lane-linear (chunk
  // position = lane % 4), so the swizzle goes on the SOURCE chunk.
  auto swz = [](int row) { return (row >> 2) & 3; };

  // ---- this thread's three haloed-patch rows (one 16-B chunk of each plane of each) ----
  // byte offsets of the row's 128-B group 0 in the two sources (fp32-sized tensors pass 2 GiB: 64-bit)
  long long aoff0[3], aoff1[3];
  bool aok[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = i * NTHR + t;
    const int r = q >> 2, pos = q & 3;
    const int hy = r / HW, hx = r - hy * HW;
    const int iy = y0 + hy + dy0, ix = x0 + hx + dx0;
    aok[i] = r < HH * HW && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
    const int gs = (pos ^ swz(r)) * 16;
    aoff0[i] = aok[i] ? ((long long)((long long)b * a.s0.H + iy) * a.s0.W + ix) * a.s0.pitch * 4 + gs : 0;
    aoff1[i] = aok[i] ? ((long long)((long long)b * a.s1.H + iy) * a.s1.W + ix) * a.s1.pitch * 4 + gs : 0;
  }
  // weights: one 16-B chunk per thread, plane and K step (threads beyond the tile idle)
  constexpr int WCHUNKS = BN * 4;
  const int wr = (t >> 2) % BN;
  // PAIR: LDS weight rows 0-63 come from phase (py, 0)'s 64 rows, rows 64-127 from phase (py, 1)'s
  const size_t woff = PAIR ? (size_t)(wr >> 6) * a.w_phase_stride + ((((wr & 63) >> 5) * nk * 32 + (wr & 31)) * BKH + ((t & 3) ^ swz(wr)) * 8)
                           : (size_t)((((n0 + wr) >> 5) * nk * 32 + (wr & 31)) * BKH + ((t & 3) ^ swz(wr)) * 8);   // halves; + ks * 1024 per K step

  auto dma_a = [&](int chunk, int i) {   // pass i (0..2) of the haloed patch of channel chunk `chunk`, both planes
    const int cc = chunk * BKH;
    const bool first = cc < a.s0.c;
    const char* base = first ? (const char*)a.s0.ptr + (size_t)cc * 4 : (const char*)a.s1.ptr + (size_t)(cc - a.s0.c) * 4;
    const long long off = first ? aoff0[i] : aoff1[i];
    const char* g = base + off;
    half_t* dst = As + (size_t)(chunk & 1) * 2 * A_BUF + (size_t)(i * NTHR + wave_u * 64) * 8;
    if ((i * NTHR + wave_u * 64) / 4 < AROWS_PAD) {   // the last pass is partial (THP = 16: waves 0..4 cover rows 256..335)
      __builtin_amdgcn_global_load_lds((gptr_t)(aok[i] ? (const void*)g : a.zeros), (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(aok[i] ? (const void*)(g + 64) : a.zeros), (lptr_t)(dst + A_BUF), 16, 0, 0);
    }
  };
  auto dma_w = [&](int chunk, int tap, int buf) {
    if (WCHUNKS >= NTHR || t < WCHUNKS) {
      const size_t kofs = (size_t)(tap * nchunk + chunk) * (32 * BKH);
      half_t* dst = Ws + (size_t)buf * 2 * W_TILE + (size_t)(wave_u * 64) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)(wh + woff + kofs), (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(wl + woff + kofs), (lptr_t)(dst + W_TILE), 16, 0, 0);
    }
  };

  float16_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment j of this wave = patch rows 2f, 2f+1 (f = wm*TM + j); row of tap (0,0) per lane.  Lane 16+i takes column
  // (i - (HW - 16)) mod 16 of the second row so that the 16 lanes of a ds_read_b128 group hit 16 different LDS rows mod 16
  // (kernels_halo.hip).
  const int xrot = (l31 < 16) ? l31 : ((l31 - (HW - 16)) & 15);
  int row0[TM];
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int f = wm * TM + j;
    row0[j] = (2 * f + (l31 >> 4)) * HW + xrot;
  }
  const int flw = swz(l31);   // weight rows of one fragment differ by multiples of 32

  // prologue: patch of chunk 0 + weights of step 0
  dma_a(0, 0);
  dma_a(0, 1);
  if (HH * HW > 2 * (NTHR / 4)) dma_a(0, 2);
  dma_w(0, 0, 0);
  __syncthreads();

  int step = 0;
  for (int c = 0; c < nchunk; ++c) {
    const half_t* Ah = As + (size_t)(c & 1) * 2 * A_BUF;
    const half_t* Al = Ah + A_BUF;
    for (int tap = 0; tap < taps; ++tap, ++step) {
      // next step's weight tile, and the next chunk's patch spread over this chunk's first steps
      const bool last_t = tap + 1 == taps;
      if (!(last_t && c + 1 == nchunk)) dma_w(last_t ? c + 1 : c, last_t ? 0 : tap + 1, (step + 1) & 1);
      if (c + 1 < nchunk) {   // static pass indices: the row tables stay in registers
        if (tap == 0) dma_a(c + 1, 0);
        else if (tap == 1) dma_a(c + 1, 1);
        else if (tap == 2) dma_a(c + 1, 2);
      }
      const int ty = tap / a.KW, tx = tap - ty * a.KW;
      const half_t* Wh = Ws + (size_t)(step & 1) * 2 * W_TILE + (size_t)(wn * TN * 32 + l31) * BKH;
      const half_t* Wl = Wh + W_TILE;
      const int tapoff = ty * HW + tx + (PAIR ? wn : 0);   // PAIR: phase px = 1 (wn = 1) reads one column further right
#pragma unroll
      for (int kk = 0; kk < BKH / 16; ++kk) {
        half8_t fwh[TN], fwl[TN], fxh[TM], fxl[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int co = ((kk * 2 + khalf) ^ flw) * 8;
          fwh[i] = *(const half8_t*)(Wh + i * 32 * BKH + co);
          fwl[i] = *(const half8_t*)(Wl + i * 32 * BKH + co);
        }
#pragma unroll
        for (int jj = 0; jj < TM; ++jj) {
          const int row = row0[jj] + tapoff;
          const int co = row * BKH + (((kk * 2 + khalf) ^ swz(row)) * 8);
          fxh[jj] = *(const half8_t*)(Ah + co);
          fxl[jj] = *(const half8_t*)(Al + co);
        }
        // the two small terms first, the leading term last; term-major so that dependent MFMAs are TN*TM apart
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int jj = 0; jj < TM; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl[i], fxh[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int jj = 0; jj < TM; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[i], fxl[jj], acc[i][jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int jj = 0; jj < TM; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[i], fxh[jj], acc[i][jj], 0, 0, 0);
      }
      __syncthreads();   // waits the DMAs (vmcnt 0) and fences the LDS buffers for reuse
    }
  }

  // ---- epilogue: undo the weight scale + bias -> fp32 LDS tile -> the shared store loop (split_epilogue.h) ----
  float* stg = (float*)lds;
  const int hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nl = (wn * TN + i) * 32 + 4 * hi + 8 * g;
      const int nc = PAIR ? (nl & 63) : n0 + nl;
      const float4_t os = *(const float4_t*)(a.oscale + nc), bs = *(const float4_t*)(a.bias + nc);   // padded to Npad
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int p = (wm * TM + j) * 32 + (l31 & 16) + xrot;   // the pixel this lane's MFMA column stands for
        float4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * g + e] * os[e] + bs[e];
        *(float4_t*)(stg + p * BN + (split_stg_chunk<BN>(p, nl >> 2) << 2)) = o;
      }
    }
  __syncthreads();
  // PAIR: staged columns 64-127 are channels 0-63 of phase px = 1
  split_store_tile<BN, BMH, NTHR, PAIR>(stg, a, n0, t, [&](int p, int px, size_t& opix) {
    const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
    if (oy >= a.Mh || ox >= a.Mw) return false;
    opix = ((size_t)b * a.oH + (oy * a.osy + ooy)) * a.oW + (ox * a.osx + (PAIR ? px : oox));
    return true;
  });
}

void launch_split_halo_pair(const ConvArgs& a, hipStream_t st) {
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + 15) / 16;
  dim3 grid((unsigned)(2 * tilesX * tilesY * a.B), 1, 1);
  hipLaunchKernelGGL((conv_split_halo_kernel<128, 2, 4, true>), grid, dim3(512), 0, st, a);
}

template <int BN, int WGN, int WGM, int THP>
void launch_split_halo_cfg(const ConvArgs& a, hipStream_t st) {
  const int ntn = a.Npad / BN;
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + THP - 1) / THP;
  dim3 grid((unsigned)(ntn * a.nphase * tilesX * tilesY * a.B), 1, 1);
  hipLaunchKernelGGL((conv_split_halo_kernel<BN, WGN, WGM, false, THP>), grid, dim3(32 * THP), 0, st, a);
}

}  // namespace

int g_split_halo = 1;                       // 0: everything through the 128-pixel kernel ("split_halo")
#ifdef CTD_AB_VARIANTS   // selftest build only: 64-channel 3x3 layers on 16x8 patches, two blocks per CU -- measured equal to the 128-pixel
int g_split_halo_small = 1;   // implicit-GEMM kernel (0.223 vs 0.225 ms; forward 24.69 vs 24.78 ms): those layers are not traffic-bound
#endif
long long g_split_halo_min_patches = 512;   // fewer 256-pixel patches (x phases x N tiles): the 128-pixel kernel ("split_halo_min_patches")

static bool split_halo_pair(const ConvArgs& a) { return a.nphase == 4 && a.N == 64 && a.Npad == 64; }

// Stride-1 3x3 / 2x2 windows over split-plane sources whose M grid equals the input grid, not upsampled
bool conv_split_halo_supported(const ConvArgs& a) {
  if (!g_split_halo || !conv_split_supported(a) || !a.x_sp) return false;
  if (a.stride != 1 || a.s0.up || (a.s1.c && a.s1.up)) return false;
  if (!((a.KH == 3 && a.KW == 3) || (a.KH == 2 && a.KW == 2))) return false;
  if (a.Mh != a.Hin || a.Mw != a.Win) return false;
  if (a.s0.H != a.Hin || a.s0.W != a.Win || (a.s1.c && (a.s1.H != a.Hin || a.s1.W != a.Win))) return false;
  if (a.N % 8 || a.K != a.KH * a.KW * (a.s0.c + a.s1.c)) return false;
  // 128-channel N tiles on 256-pixel patches; the two px phases of a 64-channel ConvT as one such tile; other 64-channel
  // layers on 128-pixel patches with two blocks per CU (on 256-pixel patches a 64-row weight tile leaves a wave 12 MFMAs
  // per tap step with one block per CU: measured 8-11 % slower than the 128-pixel implicit-GEMM kernel)
  const bool pair = split_halo_pair(a);
#ifdef CTD_AB_VARIANTS
  const bool small = !pair && a.Npad == 64 && g_split_halo_small;
#else
  const bool small = false;
#endif
  if (!pair && !small && a.Npad % 128) return false;
  const int thp = small ? 8 : 16;
  const long long patches = (long long)a.B * ((a.Mh + thp - 1) / thp) * ((a.Mw + TWP - 1) / TWP) * (pair ? 2 : a.nphase * (small ? 1 : a.Npad / 128));
  return patches >= g_split_halo_min_patches * (small ? 2 : 1);
}

void launch_conv_split_halo(const ConvArgs& a, hipStream_t st) {
  if (split_halo_pair(a)) launch_split_halo_pair(a, st);
#ifdef CTD_AB_VARIANTS
  else if (a.Npad == 64) launch_split_halo_cfg<64, 1, 4, 8>(a, st);
#endif
  else launch_split_halo_cfg<128, 2, 4, 16>(a, st);
}
