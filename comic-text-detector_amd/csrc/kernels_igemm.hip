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
// Tiling: 256 threads = 4 waves (64 lanes each).  Block tile BN x BM, K step 32 / 64.
// Global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, swizzle on the source address),
// double buffered, one barrier per K step; LDS rows are unpadded with XOR-swizzled 16-B
// chunks so the 16-lane ds_read_b128 groups hit distinct bank slots.
// v_mfma_f32_32x32x16_f16: A/B fragment = 8 consecutive k of row/col (lane & 31),
// k group = lane >> 5; C/D: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#include <type_traits>
#include <vector>

#include "kernels.h"

int g_igemm_occ_lo = 0;  // selftest build (-DCTD_AB_VARIANTS) only: 1 = register-staged loads instead of LDS-DMA

namespace {

// K step BK = 32 or 64 halves per LDS row (64 / 128 B), XOR-swizzled 16-B chunks.
// ABL: selftest-only instantiation that honours the ablation bits of a.k_rot; in the product
// instantiations (ABL = false) every hook below is a compile-time zero.
template <int BN, int BM, int WGN, int WGM, int BK, bool DST_F32, int MINW, int PF, bool PROF = false, bool ABL = false>
__global__ __launch_bounds__(256, MINW) void conv_igemm_kernel(ConvArgs a) {
  if (a.prio) __builtin_amdgcn_s_setprio(3);   // ahead of a co-running tail's waves in the issue arbiter
  constexpr int LP = BK;                            // LDS row pitch in halves (XOR swizzled, no pad)
  constexpr int SEGS = BK / 8;                      // 16-B chunks per row
  constexpr int RPP = 256 / SEGS;                   // rows staged per pass of the 256 threads
  constexpr int TN = BN / (32 * WGN);
  constexpr int TM = BM / (32 * WGM);
  constexpr int AROWS = BM / RPP;                   // pixel rows each thread stages
  constexpr int WCHUNKS = BN * SEGS;                // 16-B chunks of the weight tile
  constexpr int WROWS = (WCHUNKS + 255) / 256;
  static_assert(WGN * WGM == 4, "4 waves");

  constexpr int LDS_STAGE = 2 * (BM + BN) * LP, LDS_OUT = BM * (BN + 8);
  constexpr int LDS_MAIN = LDS_STAGE > LDS_OUT ? LDS_STAGE : LDS_OUT;
  // ONE LDS object: staging buffers / output tile, then this tile's biases (fetched while the K loop runs: as
  // global loads in the epilogue they cost a cold round trip right before the stores).  A second __shared__
  // array made the compiler's LDS-DMA alias tracking put `s_waitcnt vmcnt(0)` in front of the first ds_read of
  // every K step (a block no longer overlapped its own DMAs with its MFMAs).
  __shared__ __attribute__((aligned(16))) half_t lds[LDS_MAIN + 2 * BN];
  float* bias_s = (float*)(lds + LDS_MAIN);
  const int krot = ABL ? a.k_rot : 0;
  half_t* As = lds;                 // [2][BM][LP]  pixels
  half_t* Ws = lds + 2 * BM * LP;   // [2][BN][LP]  weights

  const long long T0 = PROF ? (long long)__builtin_readcyclecounter() : 0ll;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wn = wave % WGN, wm = wave / WGN;

  // ---- XCD-aware tile mapping: consecutive block ids land on different XCDs
  // (id % 8); give each XCD a contiguous run of tiles so that the N tiles of one
  // pixel tile (and neighbouring pixel tiles sharing 3x3 halos) share an L2.
  // The 4 sub-pixel phases of a ConvTranspose read the same input pixels, so they sit next to
  // each other in that order too (phase-major launch order streamed the input 4x from HBM).
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int nblk = ntn * ntm * a.nphase;
  int bid = blockIdx.x;
  {
    const int xcd = bid & 7, within = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tile_n = bid % ntn;
  const int bq = bid / ntn;
  int phase = a.nphase == 4 ? (bq & 3) : 0;
  int tile_m = a.nphase == 4 ? (bq >> 2) : bq;
  if ((krot & 8) && a.nphase == 4) {   // selftest A/B: phase-major order
    phase = bq / ntm;
    tile_m = bq % ntm;
  }
  const int n0 = tile_n * BN;
  const int m0 = tile_m * BM;
  if (t < BN) bias_s[t] = a.bias[n0 + t];   // visible after the first barrier of the K loop

  // ---- phase (ConvTranspose 4x4 s2 p1 sub-pixel decomposition) --------------
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

  // ---- per-thread staging state ----------------------------------------------
  // The K loop is instruction-issue bound if addresses are rebuilt per load (PMC:
  // SQ_ACTIVE_INST_ANY ~4x the MFMA cycles).  So everything per-thread is hoisted:
  //   aoffN[i] : signed byte offset of this thread's 16-B chunk of row i at tap (0,0)
  //   vmask[i] : bit t set <=> tap t of row i lies inside the image (zero padding otherwise)
  // and each K step only adds a wave-uniform (scalar) tap/channel offset.
  const int seg = t % SEGS;
  const int lrow = t / SEGS;  // 0..RPP-1
  const int Ct = a.s0.c + a.s1.c;
  const int nk = a.K / BK;
  // LDS image: rows of BK halves, NO padding; the 16-B chunk c of row r lives at chunk
  // position c ^ f(r), f(r) = (r / rows-per-256B) % chunks-per-row.  With it both the 8-lane
  // ds_write_b128 groups and the 16-lane ds_read_b128 groups hit distinct 16-B bank slots
  // (PMC on the padded layout: ds_write 2-way conflicts = 33 % of LDS cycles).
  // Register-staged mode: this thread loads global chunk `seg` and stores it at seg ^ f(r).
  // LDS-DMA mode (PF == 2): the DMA writes lane-linear, i.e. at position `seg`, so the thread
  // fetches global chunk seg ^ f(r) instead (swizzle on the SOURCE address).
  constexpr int RPW = 256 / (BK * 2);
  auto swz = [&](int row) { return (row / RPW) % SEGS; };
  constexpr bool GLDS = PF == 2;
  // m -> (batch, grid row, grid column) with the launcher's magic multipliers: an integer
  // division costs ~40 VALU instructions, and the per-block set-up / epilogue, not the K loop,
  // is what the shallow (K <= 128) layers spend their issue slots on (PMC: 1200 VALU per wave
  // for 8 MFMAs on the 1x1 64->64 layer before this).
  auto split = [&](int m, int& b, int& oy, int& ox) {
    const unsigned tq = (unsigned)(((unsigned long long)(unsigned)m * a.mw_mul) >> a.mw_sh);
    ox = m - (int)tq * a.Mw;
    const unsigned bb = (unsigned)(((unsigned long long)tq * a.mh_mul) >> a.mh_sh);
    oy = (int)tq - (int)bb * a.Mh;
    b = (int)bb;
  };
  int aoff0[AROWS], aoff1[AROWS];
  int pb[AROWS], piy[AROWS], pix[AROWS], gseg[AROWS];
  unsigned vmask[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    gseg[i] = GLDS ? (seg ^ swz(lrow + RPP * i)) : seg;
    const int m = m0 + lrow + RPP * i;
    const bool pv = m < a.M;
    int b, oy, ox;
    split(pv ? m : 0, b, oy, ox);
    const int iy0 = oy * a.stride + dy0, ix0 = ox * a.stride + dx0;
    pb[i] = b; piy[i] = iy0; pix[i] = ix0;
    // tensors are < 2 GiB (engine plan guard), so byte offsets fit 32 bits
    aoff0[i] = (((b * a.s0.H + iy0) * a.s0.W + ix0) * a.s0.pitch + gseg[i] * 8) * 2;
    aoff1[i] = (((b * a.s1.H + iy0) * a.s1.W + ix0) * a.s1.pitch + gseg[i] * 8) * 2;
    // tap validity is separable: rows inside x columns inside
    unsigned ym = 0, xm = 0, vm = 0;
    for (int ty = 0; ty < a.KH; ++ty) ym |= (unsigned)((unsigned)(iy0 + ty) < (unsigned)a.Hin) << ty;
    for (int tx = 0; tx < a.KW; ++tx) xm |= (unsigned)((unsigned)(ix0 + tx) < (unsigned)a.Win) << tx;
    for (int ty = 0; ty < a.KH; ++ty) vm |= ((ym >> ty) & 1u) ? xm << (ty * a.KW) : 0u;
    vmask[i] = pv ? vm : 0u;
  }
  // wave-uniform: all rows staged by this wave have all their taps inside the image
  bool interior;
  {
    const unsigned full = (1u << (a.KH * a.KW)) - 1u;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) ok = ok && vmask[i] == full;
    interior = __builtin_amdgcn_ballot_w64(ok) == ~0ull && !(krot & 32);
  }
  // weights: tile-major [n_tile][k_step][BN][BK]; per-thread constant part of the address
  int woff[WROWS];
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int r = (t + 256 * i) / SEGS;
    woff[i] = (r * BK + (GLDS ? (seg ^ swz(r)) : seg) * 8) * 2;
  }
  const char* wtile = (const char*)(wbase + (size_t)tile_n * nk * BN * BK);

  // One staged K step held in registers between its global loads and its LDS store.
  struct Stage {
    half8_t ra[AROWS];
    half8_t rw[WROWS];
    unsigned rok;   // bit i: staged row i is inside the image (else zero padding)
  };

  int cc = 0, ty = 0, tx = 0, kp = 0;  // K-step cursor of the NEXT tile to load (wave uniform)
  long long tb0 = 0, tb1 = 0;          // byte offset of tap (ty, tx) in source 0 / 1, updated on tap changes only

  using gptr_t = const __attribute__((address_space(1))) void*;
  using lptr_t = __attribute__((address_space(3))) void*;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform for the LDS-DMA base
  // LDS-DMA: a wave's 64 lanes write 64 consecutive 16-B chunks starting at a wave-uniform base
  auto dma = [&](const void* g, half_t* tile_base, int i) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(tile_base + (i * 256 + wave_u * 64) * 8), 16, 0, 0);
  };

  auto load_tile = [&](Stage& sg, int dbuf) {
    half8_t (&ra)[AROWS] = sg.ra;
    half8_t (&rw)[WROWS] = sg.rw;
    unsigned& rok = sg.rok;
    half_t* Ad = As + (size_t)dbuf * BM * LP;
    half_t* Wd = Ws + (size_t)dbuf * BN * LP;
    // activations
    const bool first = cc < a.s0.c;
    const SrcView& s = first ? a.s0 : a.s1;
    const int ch = first ? cc : cc - a.s0.c;
    const int tap = ty * a.KW + tx;
    rok = 0;
    if ((krot & 128) && kp > 0) {
      // selftest ablation: no activation loads after the first K step
    } else if (GLDS && interior && !s.up) {
      // interior pixel tile: every tap of every row is inside the image.  One scalar base per
      // K step + this thread's constant 32-bit row offset -> no per-row VALU work at all
      // (the K loop spends its issue slots on address arithmetic, not on MFMAs, otherwise).
      const char* sb = (const char*)s.ptr + ((first ? tb0 : tb1) + ch * 2);
#pragma unroll
      for (int i = 0; i < AROWS; ++i) dma(sb + (size_t)(unsigned)(first ? aoff0[i] : aoff1[i]), Ad, i);
    } else if (!s.up) {
      const long long tapoff = (first ? tb0 : tb1) + ch * 2;   // uniform
      const char* sb = (const char*)s.ptr + tapoff;
      const int back = (int)-tapoff;                                            // -> s.ptr (always valid)
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const bool ok = (vmask[i] >> tap) & 1u;
        if (GLDS) {
          dma(ok ? (const void*)(sb + (first ? aoff0[i] : aoff1[i])) : a.zeros, Ad, i);   // padding rows DMA zeros
        } else {
          const int off = ok ? (first ? aoff0[i] : aoff1[i]) : back;
          ra[i] = *(const half8_t*)(sb + off);      // padding rows read s.ptr; zeroed at store time
          rok |= (unsigned)ok << i;
        }
      }
    } else {
      // nearest x2 upsampled producer (yolo neck): source pixel = (iy >> 1, ix >> 1)
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const bool ok = (vmask[i] >> tap) & 1u;
        const int sy = ok ? ((piy[i] + ty) >> 1) : 0, sx = ok ? ((pix[i] + tx) >> 1) : 0;
        const half_t* p = (const half_t*)s.ptr +
                          ((size_t)((size_t)(ok ? pb[i] : 0) * s.H + sy) * s.W + sx) * s.pitch + ch + gseg[i] * 8;
        if (GLDS) {
          dma(ok ? (const void*)p : a.zeros, Ad, i);
        } else {
          ra[i] = *(const half8_t*)p;
          rok |= (unsigned)ok << i;
        }
      }
    }
    // weights
    const char* wk = wtile + (size_t)kp * (BN * BK * 2);
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      if ((WCHUNKS >= 256 || t + 256 * i < WCHUNKS) && !((krot & 64) && kp > 0)) {
        if (GLDS) dma(wk + woff[i], Wd, i);
        else rw[i] = *(const half8_t*)(wk + woff[i]);
      }
    // advance cursor
    ++kp;
    cc += BK;
    if (cc == Ct) {
      cc = 0;
      if (++tx == a.KW) { tx = 0; ++ty; }
      tb0 = (((long long)ty * a.s0.W + tx) * a.s0.pitch) * 2;
      tb1 = (((long long)ty * a.s1.W + tx) * a.s1.pitch) * 2;
    }
  };

  int sa_off[AROWS], sw_off[WROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int r = lrow + RPP * i;
    sa_off[i] = r * BK + ((seg ^ swz(r)) * 8);
  }
#pragma unroll
  for (int i = 0; i < WROWS; ++i) {
    const int r = (t + 256 * i) / SEGS;
    sw_off[i] = r * BK + ((seg ^ swz(r)) * 8);
  }
  auto store_tile = [&](const Stage& sg, int buf) {
    const half8_t (&ra)[AROWS] = sg.ra;
    const half8_t (&rw)[WROWS] = sg.rw;
    const unsigned rok = sg.rok;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      // the zero-fill select sits here, after the MFMAs, so the loads stay in flight during compute
      const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      *(half8_t*)(As + (size_t)buf * BM * LP + sa_off[i]) = ((rok >> i) & 1u) ? ra[i] : z;
    }
#pragma unroll
    for (int i = 0; i < WROWS; ++i)
      if (WCHUNKS >= 256 || t + 256 * i < WCHUNKS) *(half8_t*)(Ws + (size_t)buf * BN * LP + sw_off[i]) = rw[i];
  };

  float16_t acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // selftest instrumentation (k_rot & 16): cycle stamps of wave 0 -> where a block's time goes
  constexpr bool prof = PROF;   // selftest instantiation only; compiled out of the product kernels
  auto stamp = [&]() -> long long { return PROF ? (long long)__builtin_readcyclecounter() : 0ll; };
  long long t_issue = 0, t_comp = 0, t_wait = 0;
  const long long T1 = stamp();
  Stage sA;
  load_tile(sA, 0);
  if (!GLDS) store_tile(sA, 0);
  __syncthreads();     // with LDS-DMA pending the compiler's barrier sequence waits vmcnt(0) first
  const long long T2 = stamp();

  const int l31 = lane & 31, khalf = lane >> 5;
  const int fl = swz(l31);   // rows of one fragment differ by multiples of 32 -> same swizzle
  auto compute = [&](int buf) {
    const half_t* Ab = As + (size_t)buf * BM * LP + (size_t)(wm * TM * 32 + l31) * LP;
    const half_t* Wb = Ws + (size_t)buf * BN * LP + (size_t)(wn * TN * 32 + l31) * LP;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int co = ((kk * 2 + khalf) ^ fl) * 8;
      half8_t fw[TN], fx[TM];
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[i] = *(const half8_t*)(Wb + i * 32 * LP + co);
#pragma unroll
      for (int j = 0; j < TM; ++j) fx[j] = *(const half8_t*)(Ab + j * 32 * LP + co);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
  };
  const int abl = krot;   // selftest ablation bits: 1 = no loads in the K loop, 2 = no MFMAs, 4 = no stores
  // One K step in flight: the LDS-DMAs (or loads) of step k+1 are issued before the MFMAs of step k.
  // (A 3-deep ring with two steps in flight and a counted vmcnt measured 3-6 % SLOWER on every
  // layer shape: the K loop is not short of bytes in flight, see DESIGN.md.)
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    const long long s0 = stamp();
    if (ks + 1 < nk && !(abl & 1)) load_tile(sA, buf ^ 1);
    const long long s1 = stamp();
    if (!(abl & 2)) compute(buf);
    if (!GLDS && ks + 1 < nk) store_tile(sA, buf ^ 1);
    const long long s2 = stamp();
    __syncthreads();
    const long long s3 = stamp();
    t_issue += s1 - s0; t_comp += s2 - s1; t_wait += s3 - s2;
  }
  const long long T3 = stamp();

  // ---- epilogue: bias + activation (+ residual) -> NHWC store -----------------
  const int hi = lane >> 5;
  constexpr int OP = BN + 8;   // LDS pitch (halves) of the staged output tile, +16 B against conflicts
  half_t* Os = lds;            // [BM][OP], reuses the (now idle) staging buffers
  const bool staged = !DST_F32 && (a.pitchD % 8 == 0) && (a.N % 8 == 0);
  auto out_pixel = [&](int m) {
    int b, oy, ox;
    split(m, b, oy, ox);
    return ((size_t)b * a.oH + (oy * a.osy + ooy)) * a.oW + (ox * a.osx + oox);
  };
  auto epilogue = [&](auto act_tag) {
    constexpr int ACT = decltype(act_tag)::value;
    if (staged) {
      // The common case (fp16 destination with 16-B channel rows) as STRAIGHT-LINE code: a tile's biases first, then bias +
      // activation + rounding of its 16 values, then four 8-B LDS writes -- no branch, no wait between the groups.  The
      // general path below tests `staged`, the residual and the channel tail per group of four values: 3-4 scalar branches
      // and an `s_waitcnt lgkmcnt(0)` per group, 16 groups per wave tile in series -- ~3 k cycles per wave where a short-K 1x1
      // layer's whole K loop is ~1 k (round 5, found in the ISA after the same defect in kernels_c3b.hip / kernels_stem2.hip)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const int pl = (wm * TM + j) * 32 + l31;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          const int nl = (wn * TN + i) * 32 + 4 * hi;
          float4_t bv[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) bv[g] = *(const float4_t*)(bias_s + nl + 8 * g);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)ctd_act_fast<ACT>(acc[i][j][4 * g + e] + bv[g][e]);
            *(half4_t*)(Os + (size_t)pl * OP + nl + 8 * g) = o;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int pl = (wm * TM + j) * 32 + l31;   // pixel inside the tile
      const int m = m0 + pl;
      const bool mv = m < a.M;
      const size_t opix = out_pixel(mv ? m : 0);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int nl = (wn * TN + i) * 32 + 4 * hi;   // channel inside the tile
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + nl + 8 * g;
          const float4_t bv = *(const float4_t*)(bias_s + nl + 8 * g);   // bias is padded to the N tile
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ctd_act_fast<ACT>(acc[i][j][4 * g + e] + bv[e]);
          if (!staged && a.res && mv && n < a.N) {   // staged tiles add the residual on whole rows below
            const half4_t rv = *(const half4_t*)((const half_t*)a.res + opix * a.pitchR + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
          }
          if (staged) {
            half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)(Os + (size_t)pl * OP + nl + 8 * g) = o;
          } else if (mv && n < a.N) {
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
  };
  switch (a.act) {   // block-uniform: one specialised copy of the epilogue arithmetic runs
    case CTD_ACT_SILU: epilogue(std::integral_constant<int, CTD_ACT_SILU>{}); break;
    case CTD_ACT_LEAKY: epilogue(std::integral_constant<int, CTD_ACT_LEAKY>{}); break;
    case CTD_ACT_RELU: epilogue(std::integral_constant<int, CTD_ACT_RELU>{}); break;
    case CTD_ACT_SIGMOID: epilogue(std::integral_constant<int, CTD_ACT_SIGMOID>{}); break;
    default: epilogue(std::integral_constant<int, CTD_ACT_NONE>{}); break;
  }
  const long long T4 = stamp();
  if (staged) {
    // The MFMA C layout gives each lane 4 channels of one pixel: storing that directly makes
    // 16-B write requests scattered over 32 cache lines per instruction (PMC: TCP_TCC_WRITE_REQ
    // = bytes / 16).  Transposing through LDS lets 8-16 consecutive lanes write one pixel's
    // whole channel row with 16 B each.
    __syncthreads();
    constexpr int CPP = BN / 8;          // 16-B chunks per pixel row of the tile
    constexpr int PPI = 256 / CPP;       // pixels covered by one pass of the block
    const int c = t % CPP;
    const int n = n0 + c * 8;
    // The residual (C3 shortcut) joins here on whole 16-B channel rows: coalesced loads, all
    // issued before the first use, rounded like the reference's half-precision x + cv2(cv1(x)).
    constexpr int NIT = BM / PPI;
    size_t opx[NIT];
    bool okp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = m0 + it * PPI + t / CPP;
      okp[it] = m < a.M && n < a.N && !(abl & 4);
      opx[it] = okp[it] ? out_pixel(m) : 0;
    }
    if (a.res) {
      half8_t rv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        rv[it] = *(const half8_t*)((const half_t*)a.res + opx[it] * a.pitchR + (okp[it] ? n : 0));
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const half8_t o = *(const half8_t*)(Os + (size_t)(it * PPI + t / CPP) * OP + c * 8);
        half8_t sum;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = (half_t)((float)o[e] + (float)rv[it][e]);
        if (okp[it]) *(half8_t*)((half_t*)a.dst + opx[it] * a.pitchD + n) = sum;
      }
    } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (okp[it])
          *(half8_t*)((half_t*)a.dst + opx[it] * a.pitchD + n) = *(const half8_t*)(Os + (size_t)(it * PPI + t / CPP) * OP + c * 8);
    }
  }
  if (prof) {
    const long long T5 = stamp();
    if (t == 0) {
      long long* d = a.dbg + (size_t)blockIdx.x * 8;
      d[0] = T1 - T0; d[1] = T2 - T1; d[2] = t_issue; d[3] = t_comp; d[4] = t_wait; d[5] = T4 - T3; d[6] = T5 - T4; d[7] = T5 - T0;
    }
  }
}

// MINW = minimum waves per SIMD the register allocator must allow (launch bound).  The
// 128x128 / BK 32 tile needs 149 registers by default (3 waves/SIMD) and fits 127 under
// MINW = 4 without scratch; BK 64 tiles are LDS-limited to 2 blocks per CU anyway.
template <int BN, int BM, int WGN, int WGM, int BK>
void launch_cfg(const ConvArgs& a, bool dst_f32, hipStream_t st) {
  const int ntn = a.Npad / BN;
  const int ntm = (a.M + BM - 1) / BM;
  dim3 grid(ntn * ntm * a.nphase, 1, 1);
  constexpr int HI = BK == 32 ? 4 : 2;
  constexpr int LO = BK == 32 ? 3 : 2;
  (void)LO;
  if (dst_f32) {
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, true, HI, 1>), grid, dim3(256), 0, st, a);
#ifdef CTD_AB_VARIANTS                // selftest build only: the A/B variants that lost, and the instrumented instantiations
  } else if (g_igemm_occ_lo == 1) {   // register-staged loads (global -> VGPR -> ds_write)
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, false, HI, 1>), grid, dim3(256), 0, st, a);
  } else if ((a.k_rot & 16) && a.dbg) {   // cycle-stamped instantiation
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, false, HI, 2, true, true>), grid, dim3(256), 0, st, a);
  } else if (a.k_rot) {                   // ablation instantiation
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, false, HI, 2, false, true>), grid, dim3(256), 0, st, a);
#endif
  } else {                            // LDS-DMA (global_load_lds, 16 B per lane), +5..10 % measured
    hipLaunchKernelGGL((conv_igemm_kernel<BN, BM, WGN, WGM, BK, false, HI, 2>), grid, dim3(256), 0, st, a);
  }
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

int igemm_pick_bk(int c0, int c1, int K, int N, int log2_down) {
  const bool can64 = (c0 % 64 == 0) && (c1 % 64 == 0);
  if (g_igemm_force_bk == 32 || !can64) return 32;
  if (g_igemm_force_bk == 64) return 64;
  // measured (selftest, B=8): BK=32 at 4 waves/SIMD wins wherever the grid fills the chip;
  // BK=64 (half the barriers) only pays on the low-resolution maps (<= 1/32 scale) whose
  // grids are a few hundred blocks, and only for deep reductions with wide outputs
  return (log2_down >= 5 && K >= 512 && N >= 128) ? 64 : 32;
}

static int pick_bk(const ConvArgs& a) { return a.bk ? a.bk : igemm_pick_bk(a.s0.c, a.s1.c, a.K, a.N, 0); }

// logical weights: float [nphase][N][K] (K index = tap * Ctot + c)  ->  packed halves.
//   tiled = false : [nphase][Npad][K]
//   tiled = true  : [nphase][Npad/bn][K/bk][bn][bk]   (one K step of one N tile is contiguous)
void igemm_pack_weights(const float* logical, int nphase, int N, int K, int bn, int bk, bool tiled,
                        std::vector<half_t>& out) {
  const int Npad = (N + bn - 1) / bn * bn;
  const int nk = K / bk;
  out.assign((size_t)nphase * Npad * K, (half_t)0.f);
  for (int ph = 0; ph < nphase; ++ph)
    for (int n = 0; n < N; ++n) {
      const float* src = logical + ((size_t)ph * N + n) * K;
      for (int k = 0; k < K; ++k) {
        size_t dst;
        if (tiled)
          dst = (size_t)ph * Npad * K + ((((size_t)(n / bn) * nk + k / bk) * bn + n % bn) * bk + k % bk);
        else
          dst = ((size_t)ph * Npad + n) * K + k;
        out[dst] = (half_t)src[k];
      }
    }
}

bool igemm_supported(const ConvArgs& a) {
  constexpr int BK = 32;
  if (a.s0.c % BK || a.s1.c % BK) return false;
  if (a.s0.pitch % 8 || (a.s1.c && a.s1.pitch % 8)) return false;
  if (a.pitchD % 4 || (a.res && a.pitchR % 4)) return false;
  if (a.K % BK) return false;
  return true;
}

// n / d == (uint64(n) * mul) >> sh for every n < 2^31 (Granlund-Montgomery, 31-bit dividend)
static void magic_div(int d, unsigned& mul, unsigned& sh) {
  int L = 0;
  while ((1ll << L) < d) ++L;
  sh = 31 + L;
  mul = (unsigned)(((1ull << sh) / (unsigned)d) + 1);
}

void launch_conv_igemm(const ConvArgs& a_in, bool dst_f32, hipStream_t st) {
  ConvArgs a = a_in;
  magic_div(a.Mw, a.mw_mul, a.mw_sh);
  magic_div(a.Mh, a.mh_mul, a.mh_sh);
  a.bk = pick_bk(a);
  if (conv_halo3_supported(a, dst_f32)) {
    launch_conv_halo3(a, st);
    return;
  }
  if (conv_halo2_supported(a, dst_f32)) {
    launch_conv_halo2(a, st);
    return;
  }
  if (conv_halo_supported(a, dst_f32)) {
    launch_conv_halo(a, st);
    return;
  }
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
