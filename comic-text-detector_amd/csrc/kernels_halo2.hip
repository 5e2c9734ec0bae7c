// Big-tile halo convolution for gfx950: the sub-pixel phases of ConvTranspose 4x4/s2/p1 (reference basemodel.py
// double_conv_up_c3 :26-28) as ONE 256-pixel x 256-column tile per block, columns = (phase, output channel):
//
//   N = 256 channels : one phase per block          (17x17 haloed patch)
//   N = 128 channels : both px phases of a py        (17x18 patch serves both)
//   N =  64 channels : all four phases               (18x18 patch serves all four)
//
// Why a third MFMA convolution kernel.  kernels_halo.hip (256 pixels x 128 columns, 2 blocks per CU, one
// __syncthreads per tap step) sits at 800-895 TFLOP/s on these layers whatever is done to its schedule:
// that is the ceiling of the "barrier + drain per K step, several blocks per CU" structure.  This kernel is the other
// structure (cdna_hip_programming.md, 256x256 template): ONE block of 8 waves per CU, every wave a 128-pixel x 64-column
// tile (128 accumulator registers: 6 fragment reads per 8 MFMAs instead of 4 per 4), and the two waves of a SIMD take
// turns -- while one issues the 16 MFMAs of a tap step, the other reads its fragments of the next step from LDS and
// issues the LDS-DMA of a step further on.  The turns are made by plain s_barriers with the wave halves staggered by one
// barrier; DMAs stay in flight across them and are retired by counted s_waitcnt vmcnt (never 0 inside the loop):
//
//   half A (waves 0-3):  LOAD(0) | MFMA(0) | LOAD(1) | MFMA(1) | ...
//   half B (waves 4-7):          | LOAD(0) | MFMA(0) | LOAD(1) | ...        ("|" = s_barrier)
//
//   LOAD(k): 12 ds_read_b128 of step k (weights ring slot k % 3, patch buffer chunk % 2); LDS-DMA of the weight tile of
//            step k + 2 (slot (k + 2) % 3, last read in LOAD(k - 1) of both halves, i.e. two barriers ago) and one third
//            of the NEXT chunk's patch; s_waitcnt vmcnt(issued in this segment) = everything issued in LOAD(k - 1) has
//            landed, lgkmcnt(0) = this segment's reads are done before anybody may overwrite what they read.
//   Data of step k + 1 is therefore complete (own part waited for, barrier passed by everybody) one full turn before
//   LOAD(k + 1) reads it.
//
// Arithmetic: the K walk (channel chunk outer, tap inner, two K = 16 halves per step) and the MFMA issue order per
// accumulator are those of kernels_halo.hip, so the results are BIT-IDENTICAL to it (ctd_selftest compares them).
// Weights: the implicit-GEMM tile-major packing [phase][N/BNp][K/32][BNp][32] shared with the other two kernels.
#include <string>
#include <type_traits>

#include "kernels.h"

// The PRODUCT library ships kernels_halo3.hip for these layers; this kernel -- the investigation's first structure, with its
// cycle stamps and ablations -- is compiled into the selftest build only (-DCTD_AB_VARIANTS), like the other A/B variants.
long long g_halo2 = 1;                 // "halo2": 0 sends the ConvT layers back to kernels_halo.hip
long long g_halo2_min_blocks = 1024;   // "halo2_min_blocks": fewer 256x256 tiles than 4 per CU -> the smaller-tile kernels fill the chip
                                       // better (B = 5: 0.114 vs 0.107 ms on 256 -> 128, 0.134 vs 0.121 on 128 -> 64)

#ifndef CTD_AB_VARIANTS
bool conv_halo2_supported(const ConvArgs&, bool) { return false; }
void launch_conv_halo2(const ConvArgs&, hipStream_t) {}
int halo2_tuning_set(const char* key, long long value) {
  const std::string k(key ? key : "");
  if (k == "halo2") g_halo2 = value;
  else if (k == "halo2_min_blocks") g_halo2_min_blocks = value;
  else return halo3_tuning_set(key, value);
  return 0;
}
#else
namespace {

constexpr int TWP = 16, THP = 16;      // pixel patch
constexpr int BMH = TWP * THP;         // 256 pixels per block
constexpr int BN2 = 256;               // columns per block = phases x channels
constexpr int BKH = 32;                // channels per K chunk (64-B LDS rows)
constexpr int NTHR = 512;
constexpr int A_ROWS = 384;            // 3 DMA passes of 128 rows; 18 x 18 = 324 are used at most
constexpr int A_BUF = A_ROWS * BKH;    // halves per patch buffer (24 KB)
constexpr int W_TILE = BN2 * BKH;      // halves per weight tile (16 KB)
constexpr int NRING = 3;

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {   // counted vmcnt + all LDS reads of this wave done
  if (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
  else if (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
}

// PROF (selftest build only): cycle stamps of waves 0 and 4 -- where a turn's time goes (a.dbg: [block][2][8])
// ABL (selftest build only, timing only -- wrong results by construction): 1 no stagger (both halves in step), 2 no DMA in
// the loop, 4 no MFMAs, 8 no fragment reads, 16 no barriers in the loop
template <int NPH, bool PRIO, bool PROF = false, int ABL = 0>
__global__ __launch_bounds__(NTHR, 2) void conv_halo2_kernel(ConvArgs a) {   // 2 waves / SIMD = 1 block / CU
  if (a.prio) __builtin_amdgcn_s_setprio(3);
  auto stamp = [&]() -> long long { return PROF ? (long long)__builtin_readcyclecounter() : 0ll; };
  long long p_rd = 0, p_dma = 0, p_wait = 0, p_bar1 = 0, p_mfma = 0, p_bar2 = 0;
  const long long P0 = stamp();
  constexpr int CP = BN2 / NPH;                       // channels per phase = a.N
  constexpr int HW = NPH == 1 ? 17 : 18;              // haloed patch: columns
  constexpr int HH = NPH == 4 ? 18 : 17;              //               rows
  constexpr int LDS_STAGE = 2 * A_BUF + NRING * W_TILE;
  constexpr int OP = BN2 + 8;
  constexpr int LDS_OUT = BMH * OP;
  constexpr int LDS_MAIN = LDS_STAGE > LDS_OUT ? LDS_STAGE : LDS_OUT;
  __shared__ __attribute__((aligned(16))) half_t lds[LDS_MAIN + 2 * BN2];   // one LDS object: staging / output tile, biases
  float* bias_s = (float*)(lds + LDS_MAIN);
  half_t* As = lds;                    // [2][A_ROWS][32]
  half_t* Ws = lds + 2 * A_BUF;        // [3][256][32]

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wn = wave_u & 3, wm = wave_u >> 2;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- block -> (batch, patch, phase group); XCD-aware: each XCD gets a contiguous run of blocks
  constexpr int NPG = 4 / NPH;                        // phase groups per patch
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + THP - 1) / THP;
  const int nblk = NPG * tilesX * tilesY * a.B;
  int v = blockIdx.x;
  {
    const int xcd = v & 7, within = v >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int pg = v % NPG;
  v /= NPG;
  const int tpx = v % tilesX;
  v /= tilesX;
  const int tpy = v % tilesY;
  const int b = v / tilesY;
  const int y0 = tpy * THP, x0 = tpx * TWP;
  if (t < BN2) bias_s[t] = a.bias[t % CP];            // visible after the prologue barrier

  // patch origin relative to the output pixel: phase (py, px) reads input rows y + (py ? 0 : -1) + ty, same in x
  const int dy0 = NPH == 4 ? -1 : ((NPH == 2 ? pg : (pg >> 1)) ? 0 : -1);
  const int dx0 = NPH == 1 ? ((pg & 1) ? 0 : -1) : -1;
  const int Ct = a.s0.c + a.s1.c;
  const int nchunk = Ct / BKH;

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  // LDS rows are 64 B; 16-B chunk c of row r sits at chunk position c ^ ((r >> 2) & 3).  The DMA writes lane-linear
  // (chunk position = lane % 4), so the swizzle goes on the SOURCE chunk.
  auto swz = [](int row) { return (row >> 2) & 3; };

  // ---- this thread's three haloed-patch pieces (one 16-B chunk each) as running pointers: ap[i] = source of pass i of
  // the NEXT chunk to fetch (a padding / out-of-image piece reads `zeros` and never moves), so a DMA in the loop costs
  // one 64-bit add besides the instruction itself (the first version recomputed source selection, chunk base and
  // validity per DMA: ~40 scalar / vector instructions per tap step, 420 cycles of a 740-cycle LOAD segment)
  const char* ap[3];
  auto patch_ptrs = [&](const SrcView& sv) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int q = i * NTHR + t;
      const int r = q >> 2, pos = q & 3;
      const int hy = r / HW, hx = r - hy * HW;
      const int iy = y0 + hy + dy0, ix = x0 + hx + dx0;
      const bool ok = r < HH * HW && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      const int gs = (pos ^ swz(r)) * 8;
      // a padding / out-of-image piece walks through the zeros buffer instead (ZEROS_BYTES >= 2 * channels + 16)
      ap[i] = ok ? (const char*)sv.ptr + ((size_t)((b * sv.H + iy) * sv.W + ix) * sv.pitch + gs) * 2 : (const char*)a.zeros;
    }
  };
  patch_ptrs(a.s0);
  const int nchunk0 = a.s0.c / BKH;                   // chunks that come from the first source
  // ---- this thread's two weight pieces: LDS row r = column (phase slot, channel); wp[j] = source of the next tile to fetch
  const int BNp = CP < 128 ? CP : 128;                // rows of one packed weight tile (igemm_ntile)
  const int nkt = a.K / BKH;                          // K steps per packed N tile
  // rows r (pass 0) and r + 128 (pass 1) of the tile differ by a uniform distance: the second packed N tile (N = 256),
  // the px = 1 phase (N = 128), two phases on (N = 64) -- one running pointer per thread
  const char* wp;
  {
    const int r = t >> 2, pos = t & 3;
    const int ps = r / CP, n = r - ps * CP;
    const int phase = NPH == 1 ? pg : (NPH == 2 ? pg * 2 + ps : ps);
    wp = (const char*)((const half_t*)a.w + (size_t)phase * a.w_phase_stride + (size_t)n * BKH + ((pos ^ swz(r)) * 8));
  }
  const long long wpass = NPH == 1 ? (long long)nkt * BNp * BKH * 2 : (NPH == 2 ? a.w_phase_stride * 2 : a.w_phase_stride * 4);
  // packed K step index of step (chunk, tap) = tap * nchunk + chunk: the next tap is nchunk tiles on, the next chunk's
  // tap 0 is 3 * nchunk - 1 tiles back
  const long long wstep = (long long)BNp * BKH * 2;   // bytes per K step of a packed tile
  const long long wd_tap = wstep * nchunk, wd_wrap = wstep - 3 * wstep * nchunk;

  auto dma_a = [&](int buf, int i) {                  // pass i (0..2) of the next chunk's haloed patch into buffer `buf`
    half_t* dst = As + (size_t)buf * A_BUF + (size_t)(i * NTHR + wave_u * 64) * 8;
    __builtin_amdgcn_global_load_lds((gptr_t)ap[i], (lptr_t)dst, 16, 0, 0);
    ap[i] += BKH * 2;
  };
  auto dma_w = [&](int slot, bool wrap) {             // the next weight tile into ring slot `slot`; `wrap`: it was a tap 3
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

  // fragment j of this wave = patch rows 2f, 2f+1 (f = wm*4 + j).  Lane 16+i reads column (i - (HW - 16)) mod 16 of
  // the second row so that the 16 LDS rows of a ds_read_b128 lane group differ mod 16 (kernels_halo.hip).
  const int xrot = (l31 < 16) ? l31 : ((l31 - (HW - 16)) & 15);
  // this wave's 64 columns belong to ONE phase: its shift inside the shared patch
  const int ps_w = (wn * 64) / CP;
  const int sy = NPH == 4 ? (ps_w >> 1) : 0, sx = NPH == 1 ? 0 : (ps_w & 1);
  const int row_base = (2 * (wm * 4) + (l31 >> 4) + sy) * HW + xrot + sx;   // fragment j starts 2 * j patch rows further down
  const int flw = swz(l31);            // weight rows of one fragment differ by multiples of 32
  const int wrow = (wn * 64 + l31) * BKH;

  // ---- prologue: patch of chunk 0, weight tiles of steps 0 and 1
  if (nchunk0 == 0) patch_ptrs(a.s1);
  dma_a(0, 0);
  dma_a(0, 1);
  dma_a(0, 2);
  if (nchunk0 == 1 && nchunk > 1) patch_ptrs(a.s1);  // the pointers now stand for chunk 1
  dma_w(0, false);                     // step 0 = (chunk 0, tap 0)
  dma_w(1, false);                     // step 1 = (chunk 0, tap 1); nsteps >= 4
  __syncthreads();                     // vmcnt(0) + barrier
  if (wm == 1 && !(ABL & 1)) __builtin_amdgcn_s_barrier();          // half B runs one turn behind half A

  int slot = 0;                        // k % 3
  // Byte addresses (inside `lds`) of the fragment reads of the COMING step: patch rows ra[K half][fragment], weight rows
  // wa[K half] (the second 32-row fragment is 2 KB on).  They are computed among the MFMAs of the step before, by the wave
  // that issues those MFMAs: a LOAD segment then consists of LDS, vector-memory and scalar instructions only.  (With the
  // ~22 VALU instructions of the address arithmetic inside it, a LOAD segment could not overlap the partner's MFMA segment:
  // both waves of a SIMD feed one VALU issue port, and ablations -- ST_H2_ABL -- showed LOAD-only + MFMA-only = the whole
  // loop.)
  int ra[2][4], wa[2];
  auto read_addrs = [&](int cn, int tapn, int slotn) {
    int rb = row_base;                 // opaque: otherwise all 32 (tap, fragment, K half) addresses are hoisted into registers
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
    // finished values, here: the compiler otherwise leaves the last operation of each address to the point of use
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      asm volatile("" : "+v"(wa[kk]));
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(ra[kk][j]));
    }
  };
  read_addrs(0, 0, 0);
  // One channel chunk = four tap steps.  LAST (the last chunk) is a compile-time flag so that every step's DMA count --
  // the literal of its s_waitcnt -- is known without run-time selection: before the last chunk a step issues the weight
  // tile of step k + 2 (two instructions) and, for taps 0-2, a third of the next chunk's patch; in the last chunk only
  // taps 0-1 still have a weight tile to fetch.
  auto chunk_steps = [&](int c, auto last_tag) {
    constexpr bool LAST = decltype(last_tag)::value;
    auto one_step = [&](auto tap_tag) {
      constexpr int tap = decltype(tap_tag)::value;
      // ================= LOAD(k) =================
      const long long q0 = stamp();
      half8_t fw[2][2], fx[2][4];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (!(ABL & 8)) fw[kk][i] = *(const half8_t*)((const char*)lds + wa[kk] + i * 32 * BKH * 2);
          else asm volatile("" : "=v"(fw[kk][i]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!(ABL & 8)) fx[kk][j] = *(const half8_t*)((const char*)lds + ra[kk][j]);
          else asm volatile("" : "=v"(fx[kk][j]));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const long long q1 = stamp();
      const int s2 = slot == 0 ? 2 : slot - 1;        // (k + 2) % 3
      constexpr bool more_w = !LAST || tap < 2;       // step k + 2 exists
      constexpr bool more_a = !LAST && tap < 3;
      if (more_w && !(ABL & 2)) dma_w(s2, ((tap + 2) & 3) == 3);    // step k + 2 has tap (tap + 2) & 3; after a tap 3 the pointer wraps
      if (more_a && !(ABL & 2)) {
        dma_a((c + 1) & 1, tap);
        if (tap == 2 && c + 2 == nchunk0 && c + 2 < nchunk) patch_ptrs(a.s1);   // chunk c + 2 is the second source's first
      }
      const long long q2 = stamp();
      // everything issued in LOAD(k - 1) has landed; this segment's own DMAs stay in flight
      wait_vm_lgkm0<(ABL & 2) ? 0 : (more_w ? 2 : 0) + (more_a ? 1 : 0)>();
      __builtin_amdgcn_sched_barrier(0);
      const long long q3 = stamp();
      if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const long long q4 = stamp();
      // ================= MFMA(k) =================
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!(ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[kk][i], fx[kk][j], acc[i][j], 0, 0, 0);
            else asm volatile("" : "+v"(acc[i][j]) : "v"(fw[kk][i]), "v"(fx[kk][j]));
          }
      // the coming step's read addresses, spread over the gaps between these MFMAs (2 VALU per MFMA)
      read_addrs(tap == 3 ? c + 1 : c, (tap + 1) & 3, slot == 2 ? 0 : slot + 1);
      if (PRIO) {
        if (a.prio) __builtin_amdgcn_s_setprio(3);
        else __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      const long long q5 = stamp();
      if (!(ABL & 16)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (PROF) {
        const long long q6 = stamp();
        if (a.dbg && blockIdx.x == 0 && (t == 0 || t == 256) && c < 3) {      // absolute timeline of the first 12 steps of block 0
          long long* tl = a.dbg + (size_t)gridDim.x * 16 + ((size_t)(t >> 8) * 12 + c * 4 + tap) * 6;
          tl[0] = q0, tl[1] = q3, tl[2] = q4, tl[3] = q5, tl[4] = q6, tl[5] = P0;
        }
        p_rd += q1 - q0, p_dma += q2 - q1, p_wait += q3 - q2, p_bar1 += q4 - q3, p_mfma += q5 - q4, p_bar2 += q6 - q5;
      }
      slot = slot == 2 ? 0 : slot + 1;
    };
    one_step(std::integral_constant<int, 0>{});
    one_step(std::integral_constant<int, 1>{});
    one_step(std::integral_constant<int, 2>{});
    one_step(std::integral_constant<int, 3>{});
  };
  for (int c = 0; c + 1 < nchunk; ++c) chunk_steps(c, std::false_type{});
  chunk_steps(nchunk - 1, std::true_type{});
  if (wm == 0 && !(ABL & 1)) __builtin_amdgcn_s_barrier();          // half A waits for half B's last turn
  __syncthreads();
  const long long P1 = stamp();

  // ---- epilogue: bias + activation, transposed through LDS for 16-B channel-row stores
  const int hi = lane >> 5;
  half_t* Os = lds;   // [256][OP]
  auto epilogue = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pl = (wm * 4 + j) * 32 + (l31 & 16) + xrot;   // the pixel this lane's MFMA column stands for
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
  switch (a.act) {
    case CTD_ACT_SILU: epilogue(std::integral_constant<int, CTD_ACT_SILU>{}); break;
    case CTD_ACT_LEAKY: epilogue(std::integral_constant<int, CTD_ACT_LEAKY>{}); break;
    case CTD_ACT_RELU: epilogue(std::integral_constant<int, CTD_ACT_RELU>{}); break;
    case CTD_ACT_SIGMOID: epilogue(std::integral_constant<int, CTD_ACT_SIGMOID>{}); break;
    default: epilogue(std::integral_constant<int, CTD_ACT_NONE>{}); break;
  }
  __syncthreads();
  constexpr int CPP = BN2 / 8;         // 16-B chunks per pixel row of the tile (32)
  constexpr int PPI = NTHR / CPP;      // pixels covered by one pass of the block (16)
  const int cch = t % CPP;
  const int col = cch * 8;
  const int ps_o = col / CP, n = col - ps_o * CP;     // phase slot and channel of this thread's chunk
  const int py = NPH == 4 ? (ps_o >> 1) : (NPH == 2 ? pg : (pg >> 1));
  const int px = NPH == 1 ? (pg & 1) : (ps_o & 1);
  // Every tile is full (conv_halo2_supported demands map sizes that are multiples of the patch): pass `it` is patch row
  // `it`, this thread's patch column is t / CPP -- one running pointer, all 16 LDS reads in flight before the first
  // store.  (The first version recomputed a bounds-checked 64-bit address per pass and waited for each LDS read before its
  // store: ~15 k cycles per tile, 13-45 % of a block's life with nothing to overlap it at one block per CU.)
  {
    const int pcol = t / CPP;
    half_t* dp = (half_t*)a.dst + (((size_t)b * a.oH + (y0 * 2 + py)) * a.oW + ((x0 + pcol) * 2 + px)) * a.pitchD + n;
    const size_t dstep = (size_t)2 * a.oW * a.pitchD;
    const half_t* sp = Os + (size_t)pcol * OP + col;
    half8_t vv[BMH / PPI];
#pragma unroll
    for (int it = 0; it < BMH / PPI; ++it) vv[it] = *(const half8_t*)(sp + (size_t)it * PPI * OP);
#pragma unroll
    for (int it = 0; it < BMH / PPI; ++it) *(half8_t*)(dp + it * dstep) = vv[it];
  }
  if (PROF && a.dbg && (t == 0 || t == 256)) {
    const long long P2 = stamp();
    long long* d = a.dbg + ((size_t)blockIdx.x * 2 + (t >> 8)) * 8;
    const unsigned hwid = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID: [5:4] SIMD, [11:8] CU
    d[0] = p_rd, d[1] = p_dma, d[2] = p_wait, d[3] = p_bar1, d[4] = p_mfma, d[5] = p_bar2, d[6] = ((P2 - P1) << 8) | (hwid & 0xff),
    d[7] = P2 - P0;
  }
}

template <int NPH>
void launch_cfg(const ConvArgs& a, hipStream_t st) {
  const int tilesX = (a.Mw + TWP - 1) / TWP, tilesY = (a.Mh + THP - 1) / THP;
  dim3 grid((unsigned)((4 / NPH) * tilesX * tilesY * a.B), 1, 1);
#ifdef CTD_AB_VARIANTS
  if (a.dbg) {
    hipLaunchKernelGGL((conv_halo2_kernel<NPH, true, true>), grid, dim3(NTHR), 0, st, a);
    return;
  }
  if (g_halo2 == 2) {                                 // A/B: no s_setprio around the MFMA segments
    hipLaunchKernelGGL((conv_halo2_kernel<NPH, false>), grid, dim3(NTHR), 0, st, a);
    return;
  }
  switch (g_halo2 >> 4) {                             // timing-only ablations (ST_H2_ABL)
#define H2ABL(x) case x: hipLaunchKernelGGL((conv_halo2_kernel<NPH, true, false, x>), grid, dim3(NTHR), 0, st, a); return;
    H2ABL(1) H2ABL(2) H2ABL(4) H2ABL(8) H2ABL(6) H2ABL(10) H2ABL(12) H2ABL(14) H2ABL(17) H2ABL(30) H2ABL(26) H2ABL(22)
#undef H2ABL
    default: break;
  }
#endif
  hipLaunchKernelGGL((conv_halo2_kernel<NPH, true>), grid, dim3(NTHR), 0, st, a);
}

}  // namespace

// ConvTranspose 4x4/s2 phases (nphase = 4, 2x2 taps) with 64 / 128 / 256 output channels on non-upsampled fp16 sources
// whose channel counts are multiples of 32; fp16 destination, no residual.
bool conv_halo2_supported(const ConvArgs& a, bool dst_f32) {
  if (!g_halo2 || dst_f32 || a.res) return false;
  if (a.nphase != 4 || a.KH != 2 || a.KW != 2 || a.stride != 1 || a.osy != 2 || a.osx != 2) return false;
  if (!(a.N == 64 || a.N == 128 || a.N == 256) || a.Npad != a.N) return false;
  if (a.s0.up || (a.s1.c && a.s1.up)) return false;
  if (a.Mh != a.Hin || a.Mw != a.Win || a.Mh % THP || a.Mw % TWP) return false;
  if (a.s0.c % BKH || a.s1.c % BKH || a.bk != BKH || !a.w_tiled) return false;
  if (a.pitchD % 8) return false;
  if (a.s0.H != a.Hin || a.s0.W != a.Win || (a.s1.c && (a.s1.H != a.Hin || a.s1.W != a.Win))) return false;
  if (a.k_rot) return false;                          // selftest ablation / profiling bits belong to the other kernels
  if ((a.s0.c + a.s1.c) * 2 + 16 > CTD_ZEROS_BYTES) return false;   // padding pieces walk through the zeros buffer
  const long long tiles = (long long)a.B * ((a.Mh + THP - 1) / THP) * ((a.Mw + TWP - 1) / TWP) * (a.N / 64);
  return tiles >= g_halo2_min_blocks;
}

void launch_conv_halo2(const ConvArgs& a, hipStream_t st) {
  if (a.N == 256) launch_cfg<1>(a, st);
  else if (a.N == 128) launch_cfg<2>(a, st);
  else launch_cfg<4>(a, st);
}

int halo2_tuning_set(const char* key, long long value) {
  const std::string k(key ? key : "");
  if (k == "halo2") g_halo2 = value;
  else if (k == "halo2_min_blocks") g_halo2_min_blocks = value;
  else return halo3_tuning_set(key, value);
  return 0;
}
#endif  // CTD_AB_VARIANTS
