// ConvTranspose 4x4/s2 phases on 256-pixel x 128-column tiles with FOUR waves per block and two blocks per CU (gfx950) --
// kernels_halo2.hip's K loop (128-register wave tiles of 128 pixels x 64 columns, fragment addresses computed under the
// MFMAs, 3-slot weight ring, counted s_waitcnt vmcnt, one LDS-DMA pointer add per piece) in a shape whose prologue and
// epilogue do not stop the CU.
//
// Why.  kernels_halo2.hip runs its K loop at the matrix-pipe occupancy of the guide's 256x256 GEMM template (0.58 busy), but
// with ONE block per CU nothing overlaps a block's prologue (first DMA round trip) and epilogue (a 128-KB tile leaves at
// ~10 B per clock and CU): 13 % of a block's life at 512 -> 256, 27 % at 256 -> 128, 45 % at 128 -> 64 channels
// (ST_H2_ABL ablations).  Here a block has one wave per SIMD and 72 KB of LDS, so a CU holds two blocks with the
// same 2 waves per SIMD and 256 registers per wave: while one block stores its tile the other one computes, and the two
// waves of a SIMD (one from each block) alternate between LOAD and MFMA segments without being told to.
//
//   N = 256 channels : half a phase's channels per block     (17x17 haloed patch)
//   N = 128 channels : one phase per block                    (17x17)
//   N =  64 channels : both px phases of a py                 (17x18 patch serves both)
//
// A block's four waves run in step: LOAD(k) [12 ds_read_b128 of step k, LDS-DMA of the weight tile of step k + 2 and a
// share of the next chunk's patch, s_waitcnt vmcnt(issued now) lgkmcnt(0)] | s_barrier | MFMA(k) [16 MFMAs + the address
// arithmetic of step k + 1].  One barrier per step is enough: a ring slot is rewritten two LOAD segments after its last
// read, and every wave's reads are complete (lgkmcnt 0) before it passes the barrier in between.
// Arithmetic: K walk and MFMA order per accumulator as in kernels_halo.hip / kernels_halo2.hip -- bit-identical results.
#include <string>
#include <type_traits>

#include "kernels.h"

long long g_halo3 = 1;                 // "halo3": 0 leaves the ConvT layers to kernels_halo2.hip / kernels_halo.hip
long long g_halo3_min_blocks = 1024;   // "halo3_min_blocks"

namespace {

constexpr int TWP = 16, THP = 16;
constexpr int BMH = TWP * THP;         // 256 pixels per block
constexpr int BN3 = 128;               // columns per block
constexpr int BKH = 32;
constexpr int NTHR = 256;
constexpr int A_ROWS = 384;            // 6 DMA passes of 64 rows; 17 x 18 = 306 are used at most
constexpr int A_BUF = A_ROWS * BKH;    // halves per patch buffer (24 KB)
constexpr int W_TILE = BN3 * BKH;      // halves per weight tile (8 KB)
constexpr int NRING = 3;

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {
  static_assert(N >= 0 && N <= 4, "DMA instructions of one LOAD segment");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
  else if (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else if (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
}

// NPH: phases per block (2: N = 64, the px pair; 1: N >= 128).  NT: 128-column tiles per phase (2: N = 256).
// POST (NPH = 1, NT = 1 only): the layer's single consumer, a 1x1 conv with N2 = 64 or 128 output channels (optionally over
// the concat [x ; this layer's output]), runs on the block's staged output tile: the tile is the B operand of a second
// GEMM whose A fragments come straight from HBM / L2 (the weight matrix is 16-48 KB and every block reads all of it), and
// only out2 leaves the CU.  K order and epilogue arithmetic are the implicit-GEMM kernel's: bit-identical to the two
// launches (tests/test_gpu_edge.py).
// SEGP (NPH = 2, NT = 1: the 64-channel layer whose block holds both px phases): the layer's single consumer is the network's
// LAST layer, ConvTranspose 4x4/s2 64 -> 1 + sigmoid (kernels_fused.hip seg_final_mfma_kernel), which starts by multiplying every
// pixel's 64 channels with the 16 taps.  That product is a GEMM on this block's output tile: it runs here (the MFMA sequence of
// seg_final_mfma_kernel on the same fp16 values: bit-identical products), and the block stores P (16 fp32 per pixel = 64 B)
// instead of the 64-channel map (128 B); seg_final_gather_kernel does the col2im + sigmoid from P.  1.07 GB less written and
// 1.07 GB less read per 32 pages.
template <int NPH, int NT, int N2 = 0, bool SEGP = false>
__global__ __launch_bounds__(NTHR, 2) void conv_halo3_kernel(ConvArgs a) {   // 2 blocks / CU = 2 waves / SIMD
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  constexpr int CP = BN3 / NPH;                       // channels of a phase inside this block's columns (64 or 128)
  constexpr int HW = NPH == 2 ? 18 : 17, HH = 17;     // haloed patch
  constexpr int LDS_STAGE = 2 * A_BUF + NRING * W_TILE;
  constexpr int OP = BN3 + 8;
  constexpr int LDS_OUT = BMH * OP;
  constexpr int LDS_MAIN = LDS_STAGE > LDS_OUT ? LDS_STAGE : LDS_OUT;
  __shared__ __attribute__((aligned(16))) half_t lds[LDS_MAIN + 2 * BN3];
  float* bias_s = (float*)(lds + LDS_MAIN);
  half_t* As = lds;                    // [2][A_ROWS][32]
  half_t* Ws = lds + 2 * A_BUF;        // [3][128][32]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave_u & 1, wm = wave_u >> 1;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- block -> (batch, patch, py [, px], column tile); XCD-aware
  constexpr int NPG = (4 / NPH) * NT;                 // blocks per patch
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + THP - 1) / THP;
  const int nblk = NPG * tilesX * tilesY * a.B;
  int v = blockIdx.x;
  {
    const int xcd = v & 7, within = v >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tile_n = v % NT;
  v /= NT;
  const int pg = v % (4 / NPH);                       // NPH = 1: the phase; NPH = 2: py
  v /= (4 / NPH);
  const int tpx = v % tilesX;
  v /= tilesX;
  const int tpy = v % tilesY;
  const int b = v / tilesY;
  const int y0 = tpy * THP, x0 = tpx * TWP;
  if (t < BN3) bias_s[t] = a.bias[tile_n * BN3 + t % CP];

  const int py_b = NPH == 2 ? pg : (pg >> 1);
  const int dy0 = py_b ? 0 : -1;
  const int dx0 = NPH == 2 ? -1 : ((pg & 1) ? 0 : -1);
  const int Ct = a.s0.c + a.s1.c;
  const int nchunk = Ct / BKH;

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  auto swz = [](int row) { return (row >> 2) & 3; };

  // ---- six haloed-patch pieces per thread and chunk, as running pointers (kernels_halo2.hip)
  const char* ap[6];
  auto patch_ptrs = [&](const SrcView& sv) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int q = i * NTHR + t;
      const int r = q >> 2, pos = q & 3;
      const int hy = r / HW, hx = r - hy * HW;
      const int iy = y0 + hy + dy0, ix = x0 + hx + dx0;
      const bool ok = r < HH * HW && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      const int gs = (pos ^ swz(r)) * 8;
      ap[i] = ok ? (const char*)sv.ptr + ((size_t)((b * sv.H + iy) * sv.W + ix) * sv.pitch + gs) * 2 : (const char*)a.zeros;
    }
  };
  patch_ptrs(a.s0);
  const int nchunk0 = a.s0.c / BKH;
  // ---- two weight pieces per thread and step: LDS rows r (pass 0) and r + 64 (pass 1)
  constexpr int BNp = NPH == 2 ? 64 : 128;            // rows of one packed weight tile (igemm_ntile(N))
  const int nkt = a.K / BKH;
  const char* wp;
  {
    const int r = t >> 2, pos = t & 3;                // 0..63
    const int ps = r / CP, n = r - ps * CP;           // NPH = 2: rows 0-63 are phase px = 0 (pass 1: px = 1)
    const int phase = NPH == 2 ? pg * 2 + ps : pg;
    wp = (const char*)((const half_t*)a.w + (size_t)phase * a.w_phase_stride + ((size_t)tile_n * nkt * BNp + n) * BKH +
                       ((pos ^ swz(r)) * 8));
  }
  // pass 1 = rows 64..127: the px = 1 phase (N = 64) or 64 rows further down the same packed tile
  const long long wpass = NPH == 2 ? a.w_phase_stride * 2 : (long long)64 * BKH * 2;
  const long long wstep = (long long)BNp * BKH * 2;
  const long long wd_tap = wstep * nchunk, wd_wrap = wstep - 3 * wstep * nchunk;

  auto dma_a = [&](int buf, int i) {
    half_t* dst = As + (size_t)buf * A_BUF + (size_t)(i * NTHR + wave_u * 64) * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)ap[i], (lptr_t)dst, 16, 0, 0);
    ap[i] += BKH * 2;
  };
  auto dma_w = [&](int slot, bool wrap) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      half_t* dst = Ws + (size_t)slot * W_TILE + (size_t)(j * NTHR + wave_u * 64) * 8;
      __builtin_amdgcn_global_load_lds((gptr_t)(wp + (j ? wpass : 0)), (lptr_t)dst, 16, 0, 0);
    }
    wp += wrap ? wd_wrap : wd_tap;
  };

  float16_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int xrot = (l31 < 16) ? l31 : ((l31 - (HW - 16)) & 15);
  const int sx = NPH == 2 ? wn : 0;                   // N = 64: this wave's 64 columns are phase px = wn
  const int row_base = (2 * (wm * 4) + (l31 >> 4)) * HW + xrot + sx;
  const int flw = swz(l31);
  const int wrow = (wn * 64 + l31) * BKH;

  // ---- prologue
  if (nchunk0 == 0) patch_ptrs(a.s1);
#pragma unroll
  for (int i = 0; i < 6; ++i) dma_a(0, i);
  if (nchunk0 == 1 && nchunk > 1) patch_ptrs(a.s1);
  dma_w(0, false);
  dma_w(1, false);
  __syncthreads();

  int slot = 0;
  int ra[2][4], wa[2];
  auto read_addrs = [&](int cn, int tapn, int slotn) {
    int rb = row_base;
    asm volatile("" : "+v"(rb));
    const int tapoff = (tapn >> 1) * HW + (tapn & 1);
    const int abase = (cn & 1) * (A_BUF * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = rb + j * 2 * HW + tapoff;
      const int sw = (row >> 2) & 3, base = abase + row * (BKH * 2);
      ra[0][j] = base + ((khalf ^ sw) << 4);
      ra[1][j] = base + (((2 + khalf) ^ sw) << 4);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) wa[kk] = 2 * A_BUF * 2 + slotn * (W_TILE * 2) + wrow * 2 + (((kk * 2 + khalf) ^ flw) << 4);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      asm volatile("" : "+v"(wa[kk]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ra[kk][j]));
    }
  };
  read_addrs(0, 0, 0);

  auto chunk_steps = [&](int c, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    auto one_step = [&](auto tap_tag) {
      constexpr int tap = decltype(tap_tag)::value;
      // ================= LOAD(k) =================
      half8_t fw[2][2], fx[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fw[kk][i] = *(const half8_t*)((const char*)lds + wa[kk] + i * 32 * BKH * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) fx[kk][j] = *(const half8_t*)((const char*)lds + ra[kk][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int s2 = slot == 0 ? 2 : slot - 1;        // (k + 2) % 3
      constexpr bool more_w = !LAST || tap < 2;
      // the next chunk's six patch passes go out with taps 0-2 (two each): the last one is then retired by the wait of
      // tap 3's LOAD segment, one barrier before the chunk's first read
      constexpr int na = (LAST || tap == 3) ? 0 : 2;
      if (more_w) dma_w(s2, ((tap + 2) & 3) == 3);
      if (na) {
        dma_a((c + 1) & 1, 2 * tap);
        dma_a((c + 1) & 1, 2 * tap + 1);
        if (tap == 2 && c + 2 == nchunk0 && c + 2 < nchunk) patch_ptrs(a.s1);   // chunk c + 2 is the second source's first
      }
      wait_vm_lgkm0<(more_w ? 2 : 0) + na>();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ================= MFMA(k) =================
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk][i], fx[kk][j], acc[i][j], 0, 0, 0);
      read_addrs(tap == 3 ? c + 1 : c, (tap + 1) & 3, slot == 2 ? 0 : slot + 1);
      __builtin_amdgcn_sched_barrier(0);
      slot = slot == 2 ? 0 : slot + 1;
    };
    one_step(std::integral_constant<int, 0>{});
    one_step(std::integral_constant<int, 1>{});
    one_step(std::integral_constant<int, 2>{});
    one_step(std::integral_constant<int, 3>{});
  };
  for (int c = 0; c + 1 < nchunk; ++c) chunk_steps(c, std::false_type{});
  chunk_steps(nchunk - 1, std::true_type{});
  __syncthreads();

  // ---- epilogue
  const int hi = lane >> 5;
  half_t* Os = lds;   // [256][OP]
  auto epilogue = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = (wm * 4 + j) * 32 + (l31 & 16) + xrot;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int nl = (wn * 2 + i) * 32 + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4_t bv = *(const float4_t*)(bias_s + nl + 8 * g);
          float vv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[e] = ctd_act_fast<ACT>(acc[i][j][4 * g + e] + bv[e]);
          half4_t o = {(half_t)vv[0], (half_t)vv[1], (half_t)vv[2], (half_t)vv[3]};
          *(half4_t*)(Os + (size_t)pl * OP + nl + 8 * g) = o;
        }
      }
    }
  };
  // POST: everything the second GEMM needs that does not depend on the staged tile is set up -- and its first fragments
  // requested -- BEFORE the tile is written, so that their round trip hides under the epilogue and its barrier
  static_assert(N2 == 0 || (NPH == 1 && NT == 1), "the post conv needs every channel of the layer in one block");
  constexpr int N2S = N2 > 0 ? N2 : 32;               // (stand-in sizes keep the declarations legal when N2 == 0)
    constexpr int NF2 = N2S / 32;
    constexpr int OP2 = N2S + 8;
    static_assert(BMH * OP2 <= LDS_MAIN, "out2 tile fits the staging region");
    // wave w: tile rows (= patch pixels) 64 w .. 64 w + 63, every output channel
    const int K0 = a.post_x.c;                              // channels of the concat ahead of this layer's: 0 or a multiple of 32
    const int nk0 = K0 / BKH, nk = nk0 + BN3 / BKH;
    const half_t* W2 = (const half_t*)a.post_w;
    int r2[2];
    size_t opx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r2[j] = (wave_u * 2 + j) * 32 + l31;
      const int oy = (y0 + (r2[j] >> 4)) * 2 + py_b, ox = (x0 + (r2[j] & 15)) * 2 + (pg & 1);
      opx[j] = ((size_t)b * a.oH + oy) * a.oW + ox;
    }
    auto load_a = [&](int kc, half8_t (&fa)[NF2][2]) {
#pragma unroll
      for (int i = 0; i < NF2; ++i)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          fa[i][kk] = *(const half8_t*)(W2 + ((size_t)kc * N2S + i * 32 + l31) * BKH + (kk * 2 + khalf) * 8);
    };
    auto load_b = [&](int kc, half8_t (&fb)[2][2]) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          if (kc < nk0)   // the concat's first source, straight from HBM (wave-uniform branch)
            fb[j][kk] = *(const half8_t*)((const half_t*)a.post_x.ptr + opx[j] * a.post_x.pitch + kc * BKH + (kk * 2 + khalf) * 8);
          else
            fb[j][kk] = *(const half8_t*)(Os + (size_t)r2[j] * OP + (kc - nk0) * BKH + (kk * 2 + khalf) * 8);
        }
    };
  half8_t wseg[4];                                    // SEGP: A fragments = the 16 taps (rows 16-31 zero) x 16 channels per k step
  if constexpr (SEGP) {
    static_assert(NPH == 2 && NT == 1 && N2 == 0, "SEGP is the 64-channel layer's variant");
    const half_t* w6 = (const half_t*)a.post_w;       // [64 / 8][16 taps][8 channels] (engine.hip, the seg-final packing)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      wseg[ks] = l31 < 16 ? *(const half8_t*)(w6 + ((size_t)(2 * ks + hi) * 16 + l31) * 8) : z;
    }
  }
  half8_t fa[2][NF2][2], fb[2][2][2];
  if constexpr (N2 > 0) {
    load_a(0, fa[0]);
    if (nk0 > 0) load_b(0, fb[0]);                    // the concat's first source does not wait for the tile either
  }
  switch (a.act) {
    case CTD_ACT_SILU: epilogue(std::integral_constant<int, CTD_ACT_SILU>{}); break;
    case CTD_ACT_LEAKY: epilogue(std::integral_constant<int, CTD_ACT_LEAKY>{}); break;
    case CTD_ACT_RELU: epilogue(std::integral_constant<int, CTD_ACT_RELU>{}); break;
    case CTD_ACT_SIGMOID: epilogue(std::integral_constant<int, CTD_ACT_SIGMOID>{}); break;
    default: epilogue(std::integral_constant<int, CTD_ACT_NONE>{}); break;
  }
  __syncthreads();
  if constexpr (SEGP) {
    // wave w: tile rows (= patch pixels) 64 w .. 64 w + 63, both px phases (columns 0-63 / 64-127 of the tile)
    half8_t xb[2][2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          xb[j][ps][ks] = *(const half8_t*)(Os + (size_t)((wave_u * 2 + j) * 32 + l31) * OP + ps * 64 + 16 * ks + 8 * hi);
    __syncthreads();                                  // every wave holds its operands: the tile's space is free
    float* Pst = (float*)lds;                         // [16 patch rows][32 output columns][16 taps] fp32 = 32 KB
    static_assert(16 * 32 * 16 * 4 <= LDS_MAIN * 2, "P tile fits the staging region");
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        float16_t pa;
#pragma unroll
        for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(wseg[ks], xb[j][ps][ks], pa, 0, 0, 0);
        // rows n = (r & 3) + 8 (r >> 2) + 4 hi: r = 0..3 -> taps 4 hi .. 4 hi + 3, r = 4..7 -> taps 8 + 4 hi .. (r >= 8: unused rows)
        const int r = (wave_u * 2 + j) * 32 + l31;
        float* d = Pst + (size_t)(((r >> 4) * 32 + 2 * (r & 15) + ps) * 16);
        const float4_t lo = {pa[0], pa[1], pa[2], pa[3]}, hi4 = {pa[4], pa[5], pa[6], pa[7]};
        *(float4_t*)(d + 4 * hi) = lo;
        *(float4_t*)(d + 8 + 4 * hi) = hi4;
      }
    __syncthreads();
    // 512 pixels x 64 B: a patch row's 32 output columns are 2 KB contiguous in P
    float* Pg = (float*)a.post_dst;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * NTHR + t, up = idx >> 2, q = idx & 3;
      const int row = up >> 5, col = up & 31;
      const size_t opx = ((size_t)b * a.oH + (y0 + row) * 2 + py_b) * a.oW + x0 * 2 + col;
      *(float4_t*)(Pg + opx * 16 + q * 4) = *(const float4_t*)(Pst + (size_t)up * 16 + q * 4);
    }
    return;
  }
  if constexpr (N2 > 0) {
    float16_t acc2[NF2][2];
#pragma unroll
    for (int i = 0; i < NF2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    if (nk0 == 0) load_b(0, fb[0]);
    for (int kc = 0; kc < nk; kc += 2) {                   // two chunks per trip: the double buffers keep static indices
      if (kc + 1 < nk) {
        load_a(kc + 1, fa[1]);
        load_b(kc + 1, fb[1]);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < NF2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][i][kk], fb[0][j][kk], acc2[i][j], 0, 0, 0);
      if (kc + 1 < nk) {
        if (kc + 2 < nk) {
          load_a(kc + 2, fa[0]);
          load_b(kc + 2, fb[0]);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
          for (int i = 0; i < NF2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][i][kk], fb[1][j][kk], acc2[i][j], 0, 0, 0);
      }
    }
    __syncthreads();                                        // every wave has read its rows of the layer's tile
    half_t* Os2 = lds;                                      // [256][OP2]
    auto epilogue2 = [&](auto act_tag) {
      constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < NF2; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int nl = i * 32 + 4 * hi + 8 * g;
            const float4_t bv = *(const float4_t*)(a.post_bias + nl);
            float vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = ctd_act_fast<ACT>(acc2[i][j][4 * g + e] + bv[e]);
            half4_t o = {(half_t)vv[0], (half_t)vv[1], (half_t)vv[2], (half_t)vv[3]};
            *(half4_t*)(Os2 + (size_t)r2[j] * OP2 + nl) = o;
          }
    };
    switch (a.post_act) {
      case CTD_ACT_SILU: epilogue2(std::integral_constant<int, CTD_ACT_SILU>{}); break;
      case CTD_ACT_LEAKY: epilogue2(std::integral_constant<int, CTD_ACT_LEAKY>{}); break;
      case CTD_ACT_RELU: epilogue2(std::integral_constant<int, CTD_ACT_RELU>{}); break;
      default: epilogue2(std::integral_constant<int, CTD_ACT_NONE>{}); break;
    }
    __syncthreads();
    constexpr int CPP2 = N2 / 8, PPI2 = NTHR / CPP2;        // 16-B chunks per pixel row; pixels per pass
    const int c2 = t % CPP2;
#pragma unroll
    for (int it = 0; it < BMH / PPI2; ++it) {
      const int r = it * PPI2 + t / CPP2;
      const int oy = (y0 + (r >> 4)) * 2 + py_b, ox = (x0 + (r & 15)) * 2 + (pg & 1);
      *(half8_t*)((half_t*)a.post_dst + (((size_t)b * a.oH + oy) * a.oW + ox) * a.post_pitch + c2 * 8) =
          *(const half8_t*)(Os2 + (size_t)r * OP2 + c2 * 8);
    }
    return;
  }
  constexpr int CPP = BN3 / 8;         // 16 chunks of 16 B per pixel row of the tile
  constexpr int PPI = NTHR / CPP;      // 16 pixels per pass = one patch row
  const int cch = t % CPP;
  const int col = cch * 8;
  const int ps_o = col / CP, n = tile_n * BN3 + col - ps_o * CP;
  const int px = NPH == 2 ? ps_o : (pg & 1);
  {
    const int pcol = t / CPP;
    half_t* dp = (half_t*)a.dst + (((size_t)b * a.oH + (y0 * 2 + py_b)) * a.oW + ((x0 + pcol) * 2 + px)) * a.pitchD + n;
    const size_t dstep = (size_t)2 * a.oW * a.pitchD;
    const half_t* sp = Os + (size_t)pcol * OP + col;
    half8_t vv[BMH / PPI];
#pragma unroll
    for (int it = 0; it < BMH / PPI; ++it) vv[it] = *(const half8_t*)(sp + (size_t)it * PPI * OP);
#pragma unroll
    for (int it = 0; it < BMH / PPI; ++it) *(half8_t*)(dp + it * dstep) = vv[it];
  }
}

template <int NPH, int NT, int N2 = 0, bool SEGP = false>
void launch_cfg(const ConvArgs& a, hipStream_t st) {
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + THP - 1) / THP;
  dim3 grid((unsigned)((4 / NPH) * NT * tilesX * tilesY * a.B), 1, 1);
  hipLaunchKernelGGL((conv_halo3_kernel<NPH, NT, N2, SEGP>), grid, dim3(NTHR), 0, st, a);
}

}  // namespace

bool conv_halo3_supported(const ConvArgs& a, bool dst_f32) {
  if (!g_halo3 || dst_f32 || a.res) return false;
  if (a.nphase != 4 || a.KH != 2 || a.KW != 2 || a.stride != 1 || a.osy != 2 || a.osx != 2) return false;
  if (!(a.N == 64 || a.N == 128 || a.N == 256) || a.Npad != a.N) return false;
  if (a.s0.up || (a.s1.c && a.s1.up)) return false;
  if (a.Mh != a.Hin || a.Mw != a.Win || a.Mh % THP || a.Mw % TWP) return false;
  if (a.s0.c % BKH || a.s1.c % BKH || a.bk != BKH || !a.w_tiled) return false;
  if (a.pitchD % 8) return false;
  if (a.s0.H != a.Hin || a.s0.W != a.Win || (a.s1.c && (a.s1.H != a.Hin || a.s1.W != a.Win))) return false;
  if (a.k_rot) return false;
  if ((a.s0.c + a.s1.c) * 2 + 16 > CTD_ZEROS_BYTES) return false;
  const long long blocks = (long long)a.B * (a.Mh / THP) * (a.Mw / TWP) * (a.N == 64 ? 2 : (a.N == 128 ? 4 : 8));
  return blocks >= g_halo3_min_blocks;
}

// the fused form of (ConvTranspose with 128 output channels, its single 1x1 consumer): `a` carries post_*
bool conv_halo3_post_supported(const ConvArgs& a) {
  if (!a.post_w || a.N != 128 || !(a.post_n == 64 || a.post_n == 128)) return false;
  // the second epilogue knows these activations only (a sigmoid consumer stays its own launch)
  if (a.post_act != CTD_ACT_NONE && a.post_act != CTD_ACT_SILU && a.post_act != CTD_ACT_LEAKY && a.post_act != CTD_ACT_RELU) return false;
  if (a.post_pitch % 8 || a.post_x.c % BKH || a.post_x.c > 64 || a.post_x.up) return false;
  if (a.post_x.c && (a.post_x.pitch % 8 || a.post_x.H != a.oH || a.post_x.W != a.oW)) return false;
  return conv_halo3_supported(a, false);
}

// the fused form of (ConvTranspose with 64 output channels, the 64 -> 1 seg-final ConvTranspose's tap products): post_n == -16
bool conv_halo3_segp_supported(const ConvArgs& a) {
  if (!a.post_w || !a.post_dst || a.N != 64 || a.post_n != -16) return false;
  if ((((uintptr_t)a.post_w | (uintptr_t)a.post_dst) & 15) != 0) return false;
  return conv_halo3_supported(a, false);
}

void launch_conv_halo3(const ConvArgs& a, hipStream_t st) {
  if (a.post_w && a.N == 64 && a.post_n == -16) return launch_cfg<2, 1, 0, true>(a, st);
  if (a.post_w && a.N == 128) {
    if (a.post_n == 64) launch_cfg<1, 1, 64>(a, st);
    else launch_cfg<1, 1, 128>(a, st);
    return;
  }
  if (a.N == 64) launch_cfg<2, 1>(a, st);
  else if (a.N == 128) launch_cfg<1, 1>(a, st);
  else launch_cfg<1, 2>(a, st);
}

int halo3_tuning_set(const char* key, long long value) {
  const std::string k(key ? key : "");
  if (k == "halo3") g_halo3 = value;
  else if (k == "halo3_min_blocks") g_halo3_min_blocks = value;
  else return -1;
  return 0;
}
