// Bottleneck (+ cv3) of a C3 block with 64 or 128 hidden channels in ONE kernel (gfx950) -- the part of `C3.forward`
// (reference models/yolov5/common.py:125-138, Bottleneck :94-106; the heads' C3s: basemodel.py:21-45) behind the merged
// cv1 | cv2 GEMM:
//
//     t   = act(m.cv1 y1)                        1x1, CH -> CH
//     b   = [y1 +] act(m.cv2 t)                  3x3, CH -> CH, zero padding, shortcut when the block has one
//     out = act(cv3 [b ; y2])                    1x1, 2 CH -> 2 CH                      (CV3; else the kernel stores b)
//
// Why: as three launches these layers move 9 CH channel rows per pixel through HBM (y1 in, t out / in, y1 in again as the
// residual, b out, [b ; y2] in, out) where the fused form moves 4 CH (+ the halo overlap, served by L2); their 1x1
// launches already run at 3.2-4.4 TB/s (`upconv5.conv.0.cv3`, `.m.0.cv1`), so only removing bytes shortens them.  The
// 32-channel kernel (kernels_c3.hip) also folds cv1 | cv2 in; here the block's input has 128-768 channels and that GEMM
// stays its own launch (it is a plain byte-bound 1x1 with no reuse to win).
//
// A block owns a 16x8 pixel patch.  LDS holds ONE haloed image [CH / 32 planes][192 rows][32 channels] (rows 64 B, 16-B
// chunks XOR-swizzled by (row >> 2) & 3): y1 arrives in it by LDS-DMA, m.cv1 turns it into t IN PLACE (a wave owns whole
// pixel fragments: it reads its rows for every K before it writes them), the shortcut values were lifted into registers
// in accumulator layout before that, the nine taps read t at shifted rows, and b overwrites the first 128 rows for cv3.
// y2 never touches LDS: it is cv3's B operand straight from HBM into registers (fetched under the 3x3).  Weights stream
// through a 3-slot ring (one tile of CH rows x 32 channels per step, one LDS-DMA instruction per thread) with a counted
// `s_waitcnt vmcnt` and one barrier per step, as in kernels_halo3.hip -- the ring runs through all three stages, so the
// first tiles of the next stage are in flight while a stage finishes.
//
//   CH =  64: 256 threads, 37 KB of LDS -> 4 blocks / CU;   CH = 128: 512 threads, 74 KB -> 2 blocks / CU   (16 waves / CU)
//   wave (wn, wm): S2 pixel fragment(s) w [, w + NW] of the 6 haloed ones x all N; S3 / S4 patch rows 2 wm, 2 wm + 1 x
//   64 channels wn (S4: of each 128-row weight tile)
//
// Arithmetic is the unfused path's, step for step: fp16 operands, fp32 MFMA accumulation over the same K order (the 3x3's
// walk is the one the unfused dispatch would use for the layer: channel chunk outer / tap inner on kernels_halo.hip,
// K-linear on kernels_igemm.hip -- `tap_major`), every intermediate rounded to fp16 where the unfused kernels store it,
// the shortcut added to the ROUNDED conv output.  Bit-identical to the launches it replaces
// (tests/test_gpu_edge.py::test_fused_blocks_equal_the_layer_per_launch_program_bit_for_bit).
#include <string>
#include <type_traits>
#include <utility>

#include "kernels.h"

long long g_c3b_min_patches = 1024;   // fewer patches (blocks) than this: the per-layer kernels ("c3b_min_patches")
int g_c3b_max_ch = 128;               // widest hidden width the kernel takes ("c3b_max_ch": 0 / 64 / 128)
// Tiling per hidden width (A/B knobs, "c3b_cfg64" / "c3b_cfg128"): 0 = 16x8 patch, wave tile 64 channels x 32 pixels (16
// waves per CU); 1 = wave tile 64 channels x 64 pixels on a 16x16 patch (64 channels) / on the 16x8 patch with four waves
// (128 channels): 8 waves per CU, a third fewer LDS reads per MFMA; 2 (64 channels only) = 16x8 patch, two waves of 64 x 64
int g_c3b_cfg64 = 0, g_c3b_cfg128 = 1;

namespace {

constexpr int BW = 16;                             // patch width; the patch height BH is a template parameter
constexpr int HW = BW + 2;                         // haloed patch width

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {
  static_assert(N >= 0 && N <= 2, "LDS-DMA instructions of one step");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
}

// the activations of ctd_act_fast (same values for every input), written without a compare + select: leaky / relu as one
// v_max -- the epilogues below are VALU-bound (PMC: 13 VALU instructions per MFMA before this)
template <int ACT> __device__ __forceinline__ float act_c3b(float v) {
  if (ACT == CTD_ACT_LEAKY) return fmaxf(v, 0.1f * v);
  if (ACT == CTD_ACT_RELU) return fmaxf(v, 0.f);
  return ctd_act_fast<ACT>(v);
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// Geometry of one instantiation.  CH hidden channels, BH patch rows, PXF pixel fragments (of 32) per wave in S3 / S4.
template <int CH, int BH, int PXF> struct Geo {
  static constexpr int HH = BH + 2, HROWS = HW * HH;             // haloed patch: 18 x 10 = 180 or 18 x 18 = 324 rows
  static constexpr int NFH = (HROWS + 31) / 32;                  // haloed MFMA pixel fragments: 6 / 11
  static constexpr int NROW = NFH * 32;                          // padded rows of a plane: 192 / 352
  static constexpr int PX = BW * BH, NPF = PX / 32;              // patch pixels, patch fragments: 4 / 8
  static constexpr int WN = CH / 64, WM = NPF / PXF, NW = WN * WM, NTHR = 64 * NW;
  static constexpr int NCH = CH / 32;
  static constexpr int PLANE = NROW * 32;                        // halves of one 32-channel plane of the haloed image
  static constexpr int SLOT = CH * 32;                           // halves of one ring slot: CH weight rows x 32 channels
  static constexpr int RING = NCH * PLANE, LDS_MAIN = RING + 3 * SLOT;
  static constexpr int NDW = CH * 4 / NTHR;                      // LDS-DMA instructions per thread and weight step
  static constexpr int NDY = NCH * NROW * 4 / NTHR;              // ... for the haloed patch
  static constexpr int MAXF = (NFH + NW - 1) / NW;               // haloed fragments a wave owns in S2
  static constexpr int LDS_BYTES = (LDS_MAIN + 8 * CH) * 2;
  static constexpr int BLOCKS = (160 * 1024) / LDS_BYTES;        // blocks per CU the LDS allows
  static constexpr int OCC = BLOCKS * NW / 4 < 1 ? 1 : (BLOCKS * NW / 4 > 4 ? 4 : BLOCKS * NW / 4);   // waves per SIMD
  static_assert(CH * 4 % NTHR == 0 && (NCH * NROW * 4) % NTHR == 0 && NPF % PXF == 0, "pieces divide evenly");
};

template <int CH, int BH, int PXF, int ACT, bool CV3>
__global__ __launch_bounds__((Geo<CH, BH, PXF>::NTHR), (Geo<CH, BH, PXF>::OCC)) void c3b_kernel(C3bArgs a) {
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  using G = Geo<CH, BH, PXF>;
  constexpr int NTHR = G::NTHR, NW = G::NW, WM = G::WM, NCH = G::NCH, NFH = G::NFH, NROW = G::NROW, HROWS = G::HROWS;
  constexpr int PX = G::PX, NPF = G::NPF, PLANE = G::PLANE, SLOT = G::SLOT, RING = G::RING, LDS_MAIN = G::LDS_MAIN;
  constexpr int NDW = G::NDW, NDY = G::NDY, MAXF = G::MAXF;
  constexpr int OC = CV3 ? 2 * CH : CH;            // channels this block stores
  constexpr int OP = OC + 8;                       // pitch of the staged output tile
  constexpr int NOS = PX * OP <= LDS_MAIN ? 1 : 2; // the output tile is staged in this many passes of PX / NOS pixels
  constexpr int PXH = PX / NOS;
  static_assert(PXH * OP <= LDS_MAIN && (NPF / NOS) % PXF == 0, "output tile fits the staging buffers");
  constexpr int N2 = NCH, N3 = 9 * NCH, N4 = CV3 ? 4 * NCH : 0;   // weight steps of the three stages
  constexpr int NSTEP = N2 + N3 + N4;
  __shared__ __attribute__((aligned(16))) half_t lds[LDS_MAIN + 8 * CH];
  float* bias_s = (float*)(lds + LDS_MAIN);        // [0,CH) m.cv1, [CH,2CH) m.cv2, [2CH,4CH) cv3

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = w / WM, wm = w % WM;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- block -> (page, patch); XCD-aware: each XCD gets a contiguous run of patches (shared halos share an L2)
  const int tilesX = (a.W + BW - 1) / BW, tilesY = (a.H + BH - 1) / BH;
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
  const int y0 = tpy * BH, x0 = tpx * BW;

#pragma unroll
  for (int i = t; i < 4 * CH; i += NTHR)
    bias_s[i] = i < CH ? a.bm1[i] : i < 2 * CH ? a.bm2[i - CH] : (CV3 ? a.bc3[i - 2 * CH] : 0.f);

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  auto swz = [](int row) { return (row >> 2) & 3; };
  auto dma = [&](const void* g, half_t* dst) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)dst, 16, 0, 0); };
  // haloed row r (18 per patch row): r / 18 == (r * 3641) >> 16 for r < 4096 (no integer division: ~40 VALU each)
  auto row_yx = [&](int r, int& iy, int& ix) {
    const int hy = (r * 3641) >> 16, hx = r - hy * HW;
    iy = y0 - 1 + hy;
    ix = x0 - 1 + hx;
  };
  auto inside = [&](int r) {   // ... lies inside the image
    int iy, ix;
    row_yx(r, iy, ix);
    return r < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
  };

  // ---- weights: NDW 16-B pieces per thread and step; steps run m.cv1's K chunks, the 3x3's (chunk, tap) tiles in the
  // order of the unfused dispatch, then cv3's tiles -- all CH-row pieces of the implicit-GEMM packing
  int woff[NDW];
#pragma unroll
  for (int d = 0; d < NDW; ++d) {
    const int piece = d * NTHR + t, wrow = piece >> 2, wpos = piece & 3;
    woff[d] = wrow * 32 + ((wpos ^ swz(wrow)) * 8);
  }
  half_t* const ring = lds + RING;
  auto dma_w = [&](auto g_tag) {
    constexpr int g = decltype(g_tag)::value;
    const half_t* src;
    if constexpr (g < N2) src = a.wm1 + (size_t)g * SLOT;
    else if constexpr (g < N2 + N3) {
      constexpr int s = g - N2;
      const int tile = a.tap_major ? s : (s % 9) * NCH + s / 9;
      src = a.wm2 + (size_t)tile * SLOT;
    } else src = a.wc3 + (size_t)(g - N2 - N3) * SLOT;
#pragma unroll
    for (int d = 0; d < NDW; ++d) dma(src + woff[d], ring + (g % 3) * SLOT + (d * NTHR + w * 64) * 8);
  };

  // ---- prologue: the haloed patch of y1 (NDY pieces per thread), the first two weight tiles
  if constexpr ((NROW * 4) % NTHR == 0) {
    // a plane is a whole number of passes of the block: a thread fetches the SAME rows of every plane -- row / validity /
    // address once per pass, the planes 64 B apart
    constexpr int PPP = NROW * 4 / NTHR;
#pragma unroll
    for (int i = 0; i < PPP; ++i) {
      const int q = i * NTHR + t, r = q >> 2, pos = q & 3;
      int iy, ix;
      row_yx(r, iy, ix);
      const bool ok = r < HROWS && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const half_t* g0 = (const half_t*)a.y1.ptr + ((size_t)(b * a.H + iy) * a.W + ix) * a.y1.pitch + ((pos ^ swz(r)) * 8);
#pragma unroll
      for (int pn = 0; pn < NCH; ++pn)
        dma(ok ? (const void*)(g0 + pn * 32) : a.zeros, lds + pn * PLANE + (i * NTHR + w * 64) * 8);
    }
  } else {
#pragma unroll
    for (int j = 0; j < NDY; ++j) {
      const int Q = j * NTHR + t;
      const int plane = Q / (NROW * 4), q = Q - plane * (NROW * 4);
      const int r = q >> 2, pos = q & 3;
      int iy, ix;
      row_yx(r, iy, ix);
      const void* g = inside(r) ? (const void*)((const half_t*)a.y1.ptr + ((size_t)(b * a.H + iy) * a.W + ix) * a.y1.pitch +
                                                 plane * 32 + ((pos ^ swz(r)) * 8))
                                : a.zeros;
      const int Q0 = j * NTHR + w * 64;             // the wave's first piece: a whole wave lies inside one plane
      const int plane0 = Q0 / (NROW * 4), q0 = Q0 - plane0 * (NROW * 4);
      dma(g, lds + plane0 * PLANE + q0 * 8);
    }
  }
  dma_w(std::integral_constant<int, 0>{});
  dma_w(std::integral_constant<int, 1>{});
  __syncthreads();   // waits for the LDS-DMAs (vmcnt 0) first

  // ---- fragment addressing
  // patch fragment j of this wave = patch rows 2 (wm PXF + j), + 1; the second row's lanes are rotated by HW - 16 columns
  // so the 16-lane ds_read_b128 groups meet 16 distinct bank slots (kernels_halo.hip)
  const int pcol = (l31 < 16) ? l31 : ((l31 - (HW - 16)) & 15);
  int prow[PXF], pl[PXF], rowIn[PXF];
#pragma unroll
  for (int j = 0; j < PXF; ++j) {
    prow[j] = 2 * (wm * PXF + j) + (l31 >> 4);
    pl[j] = prow[j] * BW + pcol;                          // pixel index in the patch
    rowIn[j] = (prow[j] + 1) * HW + pcol + 1;             // its row in haloed coordinates (centre tap)
  }
  auto ld = [&](const half_t* base, int row, int kc) { return *(const half8_t*)(base + row * 32 + ((kc ^ swz(row)) * 8)); };
  auto zero16 = [](float16_t& x) {
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = 0.f;
  };

  // the step protocol (kernels_halo3.hip): a step's operands are read first, then the tile two steps ahead is requested,
  // the wait retires the tile of the NEXT step (issued one step ago) and this wave's reads, the barrier makes both true
  // for every wave -- so slot (g + 2) % 3, last read in step g - 1, is free, and slot (g + 1) % 3 is complete
  auto step_sync = [&](auto g_tag) {
    constexpr int g = decltype(g_tag)::value;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g + 2 < NSTEP) {
      dma_w(std::integral_constant<int, g + 2>{});
      wait_vm_lgkm0<NDW>();
    } else {
      wait_vm_lgkm0<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lds_fence = [&]() {   // LDS writes of every wave visible to every wave (the DMAs in flight are not waited for)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ================= S2: t = act(Wm1 y1) on the haloed patch, in place, zero outside the image =====================
  // the shortcut first: y1 at this wave's S3 pixels in accumulator layout (4 consecutive channels per lane and group)
  half4_t sc[2][PXF][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PXF; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        sc[i][j][g] = *(const half4_t*)(lds + (wn * 2 + i) * PLANE + rowIn[j] * 32 + ((g ^ swz(rowIn[j])) * 8) + 4 * khalf);

  {
    bool has[MAXF];                                          // wave-uniform: haloed fragments w, w + NW, ... of NFH
    int rowA[MAXF];
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
      has[f] = w + f * NW < NFH;
      rowA[f] = 32 * (w + f * NW) + l31;
    }
    float16_t acc[MAXF][NCH];
#pragma unroll
    for (int f = 0; f < MAXF; ++f)
#pragma unroll
      for (int n = 0; n < NCH; ++n) zero16(acc[f][n]);
    static_for<0, N2>([&](auto kc_tag) {
      constexpr int kc = decltype(kc_tag)::value;
      const half_t* Wb = ring + (kc % 3) * SLOT;
      half8_t fw[NCH][2], fx[MAXF][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int n = 0; n < NCH; ++n) fw[n][kk] = ld(Wb, n * 32 + l31, kk * 2 + khalf);
#pragma unroll
        for (int f = 0; f < MAXF; ++f)
          if (has[f]) fx[f][kk] = ld(lds + kc * PLANE, rowA[f], kk * 2 + khalf);
      }
      step_sync(kc_tag);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int n = 0; n < NCH; ++n)
#pragma unroll
          for (int f = 0; f < MAXF; ++f)
            if (has[f]) acc[f][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[n][kk], fx[f][kk], acc[f][n], 0, 0, 0);
    });
    // every wave's shortcut reads completed before the barrier of step 0; the rows written here are this wave's own.
    // Straight-line code: out-of-image rows are zeroed by a mask on the packed halves (a `keep ? x : 0` per value became an
    // exec-mask branch per value)
#pragma unroll
    for (int f = 0; f < MAXF; ++f) {
      if (!has[f]) continue;
      const int row = rowA[f];
      const unsigned km = inside(row) ? 0xffffffffu : 0u;   // the 3x3's zero padding pads t, not y1
      const int sw = swz(row);
#pragma unroll
      for (int n = 0; n < NCH; ++n) {
        float4_t bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *(const float4_t*)(bias_s + n * 32 + 8 * g + 4 * khalf);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)act_c3b<ACT>(acc[f][n][4 * g + e] + bv[g][e]);
          uint2 u = __builtin_bit_cast(uint2, o);
          u.x &= km;
          u.y &= km;
          *(uint2*)(lds + n * PLANE + row * 32 + ((g ^ sw) * 8) + 4 * khalf) = u;
        }
      }
    }
  }
  lds_fence();

  // ================= S3: b = [y1 +] act(Wm2 * t)  (3x3 over the haloed t) ==========================================
  // y2 (cv3's second K half) straight into registers: B-operand layout, 16 B per lane and K step of 16
  half8_t y2r[CV3 ? NCH : 1][2][PXF];
  if constexpr (CV3) {
#pragma unroll
    for (int j = 0; j < PXF; ++j) {
      const int oy = min(y0 + prow[j], a.H - 1), ox = min(x0 + pcol, a.W - 1);
      const half_t* p = (const half_t*)a.y2.ptr + ((size_t)(b * a.H + oy) * a.W + ox) * a.y2.pitch + khalf * 8;
#pragma unroll
      for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) y2r[kc][kk][j] = *(const half8_t*)(p + kc * 32 + kk * 16);
    }
  }
  float16_t acc3[2][PXF];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PXF; ++j) zero16(acc3[i][j]);
  static_for<0, N3>([&](auto s_tag) {
    constexpr int s = decltype(s_tag)::value;
    constexpr int g = N2 + s;
    // (chunk, tap) of this step: K-linear (tap outer) or channel chunk outer, as the unfused kernel of this layer walks
    const int c = a.tap_major ? s % NCH : s / 9;
    const int tap = a.tap_major ? s / NCH : s % 9;
    const int ty = tap / 3, tx = tap - 3 * ty;
    const half_t* Wb = ring + (g % 3) * SLOT + wn * 64 * 32;
    half8_t fw[2][2], fx[PXF][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fw[i][kk] = ld(Wb, i * 32 + l31, kk * 2 + khalf);
#pragma unroll
      for (int j = 0; j < PXF; ++j) fx[j][kk] = ld(lds + c * PLANE, (prow[j] + ty) * HW + pcol + tx, kk * 2 + khalf);
    }
    step_sync(std::integral_constant<int, g>{});
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PXF; ++j)
          acc3[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i][kk], fx[j][kk], acc3[i][j], 0, 0, 0);
  });
  // Every wave's reads of t completed before the last barrier: the image may be overwritten.  Shortcut: conv output
  // rounded to fp16, then added to y1 and rounded (the reference's half-precision `x + cv2(cv1(x))`, kernels_halo.hip).
  half_t* const Os = lds;   // [PX / NOS][OP]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < PXF; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = (wn * 2 + i) * 32 + 8 * g + 4 * khalf;
        const float4_t bv = *(const float4_t*)(bias_s + CH + n);
        half4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half_t u = (half_t)act_c3b<ACT>(acc3[i][j][4 * g + e] + bv[e]);
          o[e] = a.add ? (half_t)((float)u + (float)sc[i][j][g][e]) : u;
        }
        if constexpr (CV3) *(half4_t*)(lds + (wn * 2 + i) * PLANE + pl[j] * 32 + ((g ^ swz(pl[j])) * 8) + 4 * khalf) = o;
        else *(half4_t*)(Os + pl[j] * OP + n) = o;
      }

  // 16-B channel-row stores of staged pixels [p0, p0 + PXH): OC / 8 lanes per pixel
  auto store_tile = [&](int p0) {
    constexpr int CPP = OC / 8, PPI = NTHR / CPP;
    const int cch = t % CPP;
#pragma unroll
    for (int it = 0; it < PXH / PPI; ++it) {
      const int p = it * PPI + t / CPP;
      const int oy = y0 + ((p0 + p) >> 4), ox = x0 + ((p0 + p) & 15);
      if (oy < a.H && ox < a.W)
        *(half8_t*)((half_t*)a.dst + ((size_t)(b * a.H + oy) * a.W + ox) * a.pitchD + cch * 8) =
            *(const half8_t*)(Os + p * OP + cch * 8);
    }
  };

  // ================= S4: out = act(Wc3 [b ; y2]) ==================================================================
  if constexpr (CV3) {
    lds_fence();
    float16_t acc4[2][2][PXF];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PXF; ++j) zero16(acc4[p][i][j]);
    static_for<0, N4>([&](auto s_tag) {
      constexpr int s = decltype(s_tag)::value;
      constexpr int g = N2 + N3 + s;
      // CH = 64: one 128-row tile per K chunk, two 64-row pieces p; CH = 128: two 128-row tiles p, K chunks inside
      constexpr int kc = CH == 64 ? s >> 1 : s & 7;
      constexpr int p = CH == 64 ? s & 1 : s >> 3;
      const half_t* Wb = ring + (g % 3) * SLOT + wn * 64 * 32;
      half8_t fw[2][2], fx[PXF][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fw[i][kk] = ld(Wb, i * 32 + l31, kk * 2 + khalf);
#pragma unroll
        for (int j = 0; j < PXF; ++j) {
          if constexpr (kc < NCH) fx[j][kk] = ld(lds + kc * PLANE, pl[j], kk * 2 + khalf);
          else fx[j][kk] = y2r[kc - NCH][kk][j];
        }
      }
      step_sync(std::integral_constant<int, g>{});
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < PXF; ++j)
            acc4[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i][kk], fx[j][kk], acc4[p][i][j], 0, 0, 0);
    });
    // every wave's reads of b and of the ring completed before the last barrier: the output tile may take their place
    // (in NOS passes of PXH pixels when the whole patch does not fit)
#pragma unroll
    for (int h = 0; h < NOS; ++h) {
      if (h > 0) __syncthreads();
      if ((wm * PXF) / (NPF / NOS) == h) {              // wave-uniform: this wave's pixel fragments lie in pass h
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < PXF; ++j)
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int n = (CH == 64 ? (p * 2 + i) * 32 : p * 128 + wn * 64 + i * 32) + 8 * g + 4 * khalf;
                const float4_t bv = *(const float4_t*)(bias_s + 2 * CH + n);
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (half_t)act_c3b<ACT>(acc4[p][i][j][4 * g + e] + bv[e]);
                *(half4_t*)(Os + (pl[j] - h * PXH) * OP + n) = o;
              }
      }
      __syncthreads();
      store_tile(h * PXH);
    }
  } else {
    static_assert(CV3 || NOS == 1, "the bottleneck's own output is staged in one pass");
    __syncthreads();
    store_tile(0);
  }
}

template <int CH, int BH, int PXF, bool CV3>
void launch_act(const C3bArgs& a, hipStream_t st) {
  using G = Geo<CH, BH, PXF>;
  const int tilesX = (a.W + BW - 1) / BW, tilesY = (a.H + BH - 1) / BH;
  const dim3 grid((unsigned)(tilesX * tilesY * a.B), 1, 1), blk(G::NTHR);
  switch (a.act) {
    case CTD_ACT_SILU: hipLaunchKernelGGL((c3b_kernel<CH, BH, PXF, CTD_ACT_SILU, CV3>), grid, blk, 0, st, a); break;
    case CTD_ACT_LEAKY: hipLaunchKernelGGL((c3b_kernel<CH, BH, PXF, CTD_ACT_LEAKY, CV3>), grid, blk, 0, st, a); break;
    default: hipLaunchKernelGGL((c3b_kernel<CH, BH, PXF, CTD_ACT_RELU, CV3>), grid, blk, 0, st, a); break;
  }
}

template <int CH, int BH, int PXF>
void launch_cv3(const C3bArgs& a, hipStream_t st) {
  if (a.cv3) launch_act<CH, BH, PXF, true>(a, st);
  else launch_act<CH, BH, PXF, false>(a, st);
}

int patch_rows(const C3bArgs& a) { return a.ch == 64 && g_c3b_cfg64 == 1 ? 16 : 8; }

}  // namespace

bool c3b_supported(const C3bArgs& a) {
  if (!(g_fuse & 8)) return false;
  if (!(a.ch == 64 || a.ch == 128) || a.ch > g_c3b_max_ch) return false;
  if (a.y1.c != a.ch || a.y1.up || a.y1.pitch % 8 || a.pitchD % 8) return false;
  if (a.y1.H != a.H || a.y1.W != a.W) return false;
  if (a.cv3 && (a.y2.c != a.ch || a.y2.up || a.y2.pitch % 8 || a.y2.H != a.H || a.y2.W != a.W)) return false;
  if (a.act != CTD_ACT_SILU && a.act != CTD_ACT_LEAKY && a.act != CTD_ACT_RELU) return false;
  const int bh = patch_rows(a);
  const long long patches = (long long)a.B * ((a.H + bh - 1) / bh) * ((a.W + BW - 1) / BW);
  return patches >= g_c3b_min_patches * 8 / bh;   // the threshold counts 128-pixel patches
}

void launch_c3b(const C3bArgs& a, hipStream_t st) {
  if (a.ch == 64) {
    if (g_c3b_cfg64 == 1) launch_cv3<64, 16, 2>(a, st);
    else if (g_c3b_cfg64 == 2) launch_cv3<64, 8, 2>(a, st);
    else launch_cv3<64, 8, 1>(a, st);
  } else {
    if (g_c3b_cfg128 == 1) launch_cv3<128, 8, 2>(a, st);
    else launch_cv3<128, 8, 1>(a, st);
  }
}
