// Device kernels of the detector tail (reference inference.py:148-178 after the network):
//
//  * DB text-line stage (reference utils/db_utils.py:123-211): after the two labelling passes
//    (`launch_ccl`: 8-connected foreground, 4-connected background) the per-contour quantities the
//    host geometry needs are compacted here, so no label image or probability map leaves the GPU:
//      - containment links (which hole a component sits in, which component rings a hole),
//      - sum of the probability map per component / per hole / per hole-border ring,
//      - leftmost and rightmost pixel of every component (and ring) per row: the convex hull of a
//        pixel set is the hull of its row extremes.
//  * mask refinement (reference utils/textmask.py:29-131): batched per-window kernels;
//    every text-block window of a page batch is one entry of a window table, candidate masks and
//    merged masks live as bands of packed canvases that `launch_ccl` labels in one launch.
//
// Integer / comparison work on u8 and i32 pixels: HBM- and latency-bound, no MFMA.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "tail.h"

namespace {

inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > (long long)g_tail_max_blocks) g = g_tail_max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

// ======================================================================================================
// DB stage
// ======================================================================================================
__global__ __launch_bounds__(256) void dbc_prep_kernel(DbcTables t) {
  const int b = blockIdx.y, l = blockIdx.x * 256 + threadIdx.x;
  const int hw = t.H * t.W;
  const int nf = min(t.n_f[b], t.cap), nb = min(t.n_b[b], t.cap);
  const size_t r = (size_t)b * t.cap + l;
  if (l < nf) {
    const int p = t.first_f[r];
    t.par_f[r] = (p % t.W) > 0 ? max(-t.lab[(size_t)b * hw + p - 1], 0) : 0;
    t.off_f[r] = t.st_f[r * 5 + 3];
  }
  if (l < nb) {
    const int* s = t.st_b + r * 5;
    const bool hole = s[0] > 0 && s[1] > 0 && s[0] + s[2] < t.W && s[1] + s[3] < t.H;   // does not touch the frame
    t.par_b[r] = hole ? max(t.lab[(size_t)b * hw + t.first_b[r] - 1], 0) : 0;
    t.off_b[r] = hole ? s[3] + 2 : 0;
  }
}

__device__ __forceinline__ int block_excl_scan(int v, int* sh, int* total) {   // 256 threads
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(incl, off);
    if (lane >= off) incl += u;
  }
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += sh[i];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return base + incl - v;
}

// row-table offsets: exclusive scan over [heights of the components | heights + 2 of the holes]
__global__ __launch_bounds__(256) void dbc_scan_kernel(DbcTables t) {
  __shared__ int sh[4];
  const int b = blockIdx.x;
  const int nf = min(t.n_f[b], t.cap), nb = min(t.n_b[b], t.cap);
  int carry = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int* a = (pass ? t.off_b : t.off_f) + (size_t)b * t.cap;
    const int n = pass ? nb : nf;
    for (int base = 0; base < n; base += 256) {
      const int i = base + threadIdx.x;
      const int v = i < n ? a[i] : 0;
      int total;
      const int e = block_excl_scan(v, sh, &total);
      if (i < n) a[i] = carry + e;
      carry += total;
    }
  }
  if (threadIdx.x == 0) {
    int* h = t.hdr + b * 4;
    h[0] = nf, h[1] = nb, h[2] = carry;
    h[3] = (t.n_f[b] > t.cap || t.n_b[b] > t.cap || carry > t.rcap) ? 1 : 0;
  }
}

__global__ void dbc_init_kernel(DbcTables t) {
  const long long total = (long long)t.B * t.rcap;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / t.rcap), r = (int)(i % t.rcap);
    if (r < t.hdr[b * 4 + 2]) t.row_lo[i] = 0x7fffffff, t.row_hi[i] = -1;
  }
}

__global__ __launch_bounds__(256) void dbc_accum_kernel(DbcTables t) {
  const int hw = t.H * t.W;
  const int span = gridDim.x * 256;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y;                            // one page per grid row: 32-bit index arithmetic only
  const size_t cb = (size_t)b * t.cap, rb = (size_t)b * t.rcap;
  // Four steps of the sweep at a time: their labels, then (background pixels) the ring of their component -- two dependent
  // loads that nearly every wave needs only to find out that it has nothing to add.  One step at a time a thread spent
  // its ~44 steps waiting for one load after the other.
  constexpr int DU = 4;
  for (long long pb = (long long)blockIdx.x * 256; pb < hw; pb += (long long)span * DU) {
    int keys[DU], pars[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const long long q = pb + (long long)u * span + (int)threadIdx.x;
      keys[u] = q < hw ? t.lab[(long long)b * hw + q] : 0;   // +foreground id / -background id
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) pars[u] = (keys[u] < 0 && -keys[u] <= t.cap) ? t.par_b[cb + (-keys[u]) - 1] : 0;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      if (pb + (long long)u * span >= hw) break;
      const int p0 = (int)(pb + (long long)u * span);
      const int p = min(p0 + (int)threadIdx.x, hw - 1);
      const bool live = p0 + (int)threadIdx.x < hw;
      const long long i = (long long)b * hw + p;
      const int x = p % t.W, y = p / t.W;
      const int key = keys[u];
      const int lf = max(key, 0), lb = max(-key, 0);
      // Most waves see only page background (a complement component that is no hole): nothing to add, and
      // the f64 scan below (14 cross-lane moves) is what this kernel's time goes into.
      const bool hole = live && lf <= 0 && lb > 0 && lb <= t.cap && pars[u] > 0;
      if (!__ballot(live && (lf > 0 || hole))) continue;
      const double pr = live ? (double)t.prob[(long long)b * t.prob_stride + p] : 0.0;
      // horizontal runs of one label inside the wave: the first lane of a run acts for it
      const int prev = __shfl_up(key, 1);
      const bool head = lane == 0 || prev != key || x == 0;
      const unsigned long long heads = __ballot(head);
      const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
      const int len = later ? __ffsll((long long)later) : 64 - lane;
      double ps = pr;
      for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(ps, off);
        if (lane >= off) ps += up;
      }
      const double s_tail = __shfl(ps, lane + len - 1);
      const double s_prev = __shfl_up(ps, 1);
      const double run = s_tail - (lane > 0 ? s_prev : 0.0);
      if (!live || t.hdr[b * 4 + 3]) continue;          // overflowed page: the host takes the label-image path
      if (lf > 0 && lf <= t.cap) {
        if (head) {
          unsafeAtomicAdd(t.sum_f + cb + lf - 1, run);
          const int row = t.off_f[cb + lf - 1] + (y - t.st_f[(cb + lf - 1) * 5 + 1]);
          atomicMin(t.row_lo + rb + row, x);
          atomicMax(t.row_hi + rb + row, x + len - 1);
        }
        // border ring of a hole: foreground pixels of the ringing component that 4-touch the hole (the four neighbours'
        // labels, then the four rings, are fetched together: one pixel has up to eight dependent loads here otherwise)
        int seen[4];
        int ns = 0;
        const int dq[4] = {-1, 1, -t.W, t.W};
        const bool ok[4] = {x > 0, x + 1 < t.W, y > 0, y + 1 < t.H};
        int hbs[4], prs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) hbs[k] = ok[k] ? -t.lab[i + dq[k]] : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) prs[k] = (hbs[k] > 0 && hbs[k] <= t.cap) ? t.par_b[cb + hbs[k] - 1] : -1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hb = hbs[k];
          if (hb <= 0 || hb > t.cap || prs[k] != lf) continue;
          bool dup = false;
          for (int j = 0; j < ns; ++j) dup |= seen[j] == hb;
          if (dup) continue;
          seen[ns++] = hb;
          unsafeAtomicAdd(t.ring_sum + cb + hb - 1, pr);
          atomicAdd(t.ring_cnt + cb + hb - 1, 1);
          const int row = t.off_b[cb + hb - 1] + (y - (t.st_b[(cb + hb - 1) * 5 + 1] - 1));
          atomicMin(t.row_lo + rb + row, x);
          atomicMax(t.row_hi + rb + row, x);
        }
      } else if (head && hole) {
        unsafeAtomicAdd(t.sum_b + cb + lb - 1, run);
      }
    }
  }
}

// ======================================================================================================
// mask refinement
// ======================================================================================================
__device__ __forceinline__ int gray_of(const uint8_t* p) {
  // OpenCV 4.x RGB2Gray<uchar>: 15-bit coefficients (BY15, GY15, RY15), round to nearest
  return ((int)p[0] * 3735 + (int)p[1] * 19235 + (int)p[2] * 9798 + 16384) >> 15;
}

// One histogram increment per lane.  Page backgrounds are flat, so most of a wave hits ONE bin (a 64-way
// LDS atomic conflict): the two most common values of the wave are added once by a leader lane, the rest
// fall through to plain atomics.  Lanes that already left the caller's loop are simply absent from the ballots.
__device__ __forceinline__ void hist_add(unsigned* h, int bin, bool on) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const unsigned long long m = __ballot(on);
    if (!m) return;
    const int leader = __ffsll((long long)m) - 1;
    const int v = __shfl(bin, leader);
    const unsigned long long same = __ballot(on && bin == v);
    if (lane == leader) atomicAdd(h + v, (unsigned)__popcll(same));
    on = on && bin != v;
  }
  if (on) atomicAdd(h + bin, 1u);
}

// The window kernels walk a window in groups of four horizontally consecutive pixels per thread: one
// 12-B load for the BGR bytes and one 4-/6-B load per mask row instead of a byte load each (the byte-load
// version was bound by the number of vector-memory instructions, not by bytes: rocprofv3 0.11-0.19 ms per
// kernel for ~29 Mpixel of windows per batch).  gfx950 global loads need no alignment, so the window origin
// is free.  Groups cut by the window's right edge take the byte path.
struct Grp {
  int x, y, nv;   // first pixel, row, valid pixels (1..4)
};
__device__ __forceinline__ int win_groups(const TWin& w) { return ((w.w + 3) >> 2) * w.h; }
__device__ __forceinline__ Grp win_group(const TWin& w, int g) {
  const int ngx = (w.w + 3) >> 2;
  Grp r;
  r.y = g / ngx;
  r.x = (g - r.y * ngx) * 4;
  r.nv = min(4, w.w - r.x);
  return r;
}
__device__ __forceinline__ void load_bgr4(const TWin& w, const Grp& g, uint8_t* px) {
  const uint8_t* p = w.img + ((size_t)(w.y1 + g.y) * w.img_w + w.x1 + g.x) * 3;
  if (g.nv == 4) {
    __builtin_memcpy(px, p, 12);
  } else {
    for (int k = 0; k < 12; ++k) px[k] = k < g.nv * 3 ? p[k] : 0;
  }
}
// six bytes of a row: columns x-1 .. x+4 of the window, `fill` outside the window
__device__ __forceinline__ void load_row6(const uint8_t* row, int x, int ww, int fill, uint8_t* mv) {
  if (x >= 1 && x + 5 <= ww) {
    __builtin_memcpy(mv, row + x - 1, 6);
  } else {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int xx = x - 1 + c;
      mv[c] = (xx >= 0 && xx < ww) ? row[xx] : (uint8_t)fill;
    }
  }
}

// ---- windows at least 8 pixels wide (`FAST`; the kernels pick the instantiation per window) ----------------------------
// A group cut by the window's right edge takes the byte paths above: branches whose loads are consumed inside them, a memory
// round trip in the middle of every wave that spans a row end (most do).  Here the last group of a row is the row's LAST
// four pixels instead: it overlaps its left neighbour, and its first `lo` pixels -- the neighbour's -- are skipped (sums,
// counts) or written again with the same values (stores, ORs).  Every group is then four whole pixels inside the window:
// one 12-B / 4-B / 8-B load each, no edge branch, and the loads of TW_U groups are issued before the first is used.
constexpr int TW_U = 2;
template <bool FAST>
__device__ __forceinline__ Grp win_group_lo(const TWin& w, int gi, int& lo) {
  Grp g = win_group(w, gi);
  lo = 0;
  if (FAST) {
    lo = 4 - g.nv;
    g.x -= lo;
    g.nv = 4;
  }
  return g;
}
// columns x-1 .. x+4 of a row as bytes 0..5 of one 8-byte load that stays inside the row (window at least 8 wide, x <= ww - 4);
// `fill255` ? 255 : 0 outside the window
__device__ __forceinline__ unsigned long long row6_wide(const uint8_t* row0, int x, int ww, bool fill255) {
  const int s0 = min(max(x - 1, 0), ww - 8);
  unsigned long long v;
  __builtin_memcpy(&v, row0 + s0, 8);
  const int d = x - 1 - s0;                                    // -1 (the first group of a row) .. 3
  v = d < 0 ? (v << 8) : (v >> (8 * (d & 7)));
  const int nvalid = ww - (x - 1);                             // bytes from byte 0 on that are window columns (>= 5)
  unsigned long long inside = nvalid < 8 ? ~(~0ull << (8 * (nvalid & 7))) : ~0ull;
  inside &= d < 0 ? ~0xffull : ~0ull;                          // column -1
  v &= inside;
  return fill255 ? (v | ~inside) : v;
}
// load_row6 through row6_wide when FAST
template <bool FAST>
__device__ __forceinline__ unsigned long long row6(const uint8_t* row0, int x, int ww, int fill) {
  if (FAST) return row6_wide(row0, x, ww, fill != 0);
  uint8_t mv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  load_row6(row0, x, ww, fill, mv);
  unsigned long long v;
  __builtin_memcpy(&v, mv, 8);
  return v;
}

// grey over pixels whose 3x3-eroded mask > 127 (textmask.py:58-61) and B, G, R of the whole window (Otsu, :44-47)
template <bool FAST>
__device__ __forceinline__ void tw_hist_body(const TWin& w, unsigned* h) {
  constexpr int TW_U = 1;                                        // one group: the erosion's three rows and the pixel are 44 B in flight already
  const int ng = win_groups(w);
  for (int g0 = blockIdx.x * 256 * TW_U; g0 < ng; g0 += gridDim.x * 256 * TW_U) {
    Grp g[TW_U];
    int lo[TW_U];
    uint8_t px[TW_U][12];
    unsigned long long rows[TW_U][3];
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      const int gi = g0 + u * 256 + threadIdx.x;
      g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo[u]);
      if (gi >= ng) lo[u] = 4;                                   // beyond the window: nothing counted
      load_bgr4(w, g[u], px[u]);
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {                         // a row outside the window is ignored: 255 is neutral
        const bool ok = (unsigned)(g[u].y + dy) < (unsigned)w.h;
        const unsigned long long r = row6<FAST>(w.mask + (size_t)(w.y1 + g[u].y + (ok ? dy : 0)) * w.mask_w + w.x1, g[u].x, w.w, 255);
        rows[u][dy + 1] = ok ? r : ~0ull;
      }
    }
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      int er[4] = {255, 255, 255, 255};                          // 3x3 erosion, pixels outside the window ignored
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
          er[k] = min(er[k], min((int)((rows[u][r] >> (8 * k)) & 0xff), min((int)((rows[u][r] >> (8 * k + 8)) & 0xff), (int)((rows[u][r] >> (8 * k + 16)) & 0xff))));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool on = k >= lo[u] && k < g[u].nv;
        const uint8_t* q = px[u] + 3 * k;
        hist_add(h + 256, q[0], on);
        hist_add(h + 512, q[1], on);
        hist_add(h + 768, q[2], on);
        hist_add(h, gray_of(q), on && er[k] > 127);
      }
    }
  }
}
__global__ __launch_bounds__(256) void tw_hist_kernel(const TWin* __restrict__ wins, unsigned* __restrict__ hist) {
  __shared__ unsigned h[4 * 256];
  const TWin w = wins[blockIdx.y];
  for (int i = threadIdx.x; i < 1024; i += 256) h[i] = 0;
  __syncthreads();
  if (w.w >= 8) tw_hist_body<true>(w, h);
  else tw_hist_body<false>(w, h);
  __syncthreads();
  unsigned* out = hist + (size_t)blockIdx.y * 1024;
  for (int i = threadIdx.x; i < 1024; i += 256)
    if (h[i]) atomicAdd(out + i, h[i]);
}

// kind 0: cv2.inRange(grey, lo, hi) with the integer bounds cv2 derives from the scalars (lo > hi: empty);
// kind 1..3: threshold(channel B/G/R, lo, 255, THRESH_BINARY)
__device__ __forceinline__ bool rule_on(int kind, int lo, int hi, int b, int g, int r, int grey) {
  if (kind == 0) return grey >= lo && grey <= hi;
  return (kind == 1 ? b : (kind == 2 ? g : r)) > lo;   // selects, not an indexed read: the pixel stays in registers
}

// xor distance sum(cand ? 255 - m : m) of the 6 candidate rules of every window (textmask.py:36-37)
template <bool FAST>
__device__ __forceinline__ void tw_xor_body(const TWin& w, const TRule* rs, unsigned long long* acc) {
  const int ng = win_groups(w);
  for (int g0 = blockIdx.x * 256 * TW_U; g0 < ng; g0 += gridDim.x * 256 * TW_U) {
    Grp g[TW_U];
    int lo[TW_U];
    uint8_t px[TW_U][12], mv[TW_U][4];
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      const int gi = g0 + u * 256 + threadIdx.x;
      g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo[u]);
      if (gi >= ng) lo[u] = 4;
      load_bgr4(w, g[u], px[u]);
      const uint8_t* mrow = w.mask + (size_t)(w.y1 + g[u].y) * w.mask_w + w.x1 + g[u].x;
      if (g[u].nv == 4) {
        __builtin_memcpy(mv[u], mrow, 4);
      } else {
        for (int j = 0; j < 4; ++j) mv[u][j] = j < g[u].nv ? mrow[j] : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < TW_U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < lo[u] || j >= g[u].nv) continue;
        const int m = mv[u][j];
        const int cb = px[u][3 * j], cg = px[u][3 * j + 1], cr = px[u][3 * j + 2], grey = gray_of(px[u] + 3 * j);
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (rs[k].kind >= 0) acc[k] += rule_on(rs[k].kind, rs[k].lo, rs[k].hi, cb, cg, cr, grey) ? (255 - m) : m;
      }
  }
}
__global__ __launch_bounds__(256) void tw_xor_kernel(const TWin* __restrict__ wins, const TRule* __restrict__ rules,
                                                     unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long red[4];
  const TWin w = wins[blockIdx.y];
  TRule rs[6];
  for (int k = 0; k < 6; ++k) rs[k] = rules[(size_t)blockIdx.y * 6 + k];
  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
  if (w.w >= 8) tw_xor_body<true>(w, rs, acc);
  else tw_xor_body<false>(w, rs, acc);
  for (int k = 0; k < 6; ++k) {
    unsigned long long v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long s = red[0] + red[1] + red[2] + red[3];
      if (s) atomicAdd(sums + (size_t)blockIdx.y * 6 + k, s);
    }
  }
}

template <bool FAST>
__device__ __forceinline__ void tw_render_body(const TWin& w, const TBand& bd, uint8_t* __restrict__ canvas, int canvas_w) {
  const int ng = win_groups(w);
  for (int g0 = blockIdx.x * 256 * TW_U; g0 < ng; g0 += gridDim.x * 256 * TW_U) {
    Grp g[TW_U];
    uint8_t px[TW_U][12];
    bool live[TW_U];
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      const int gi = g0 + u * 256 + threadIdx.x;
      int lo;
      live[u] = gi < ng;
      g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo);      // (an overlapping group stores its neighbour's pixels again: same values)
      load_bgr4(w, g[u], px[u]);
    }
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      if (!live[u]) continue;
      uint8_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = (rule_on(bd.kind, bd.lo, bd.hi, px[u][3 * j], px[u][3 * j + 1], px[u][3 * j + 2], gray_of(px[u] + 3 * j)) != (bd.invert != 0)) ? 255 : 0;
      uint8_t* dst = canvas + (size_t)(bd.cy + g[u].y) * canvas_w + bd.cx + g[u].x;
      if (g[u].nv == 4) {
        __builtin_memcpy(dst, o, 4);
      } else {
        for (int j = 0; j < g[u].nv; ++j) dst[j] = o[j];
      }
    }
  }
}
__global__ __launch_bounds__(256) void tw_render_kernel(const TWin* __restrict__ wins, const TBand* __restrict__ bands,
                                                        uint8_t* __restrict__ canvas, int canvas_w) {
  const TBand bd = bands[blockIdx.y];
  const TWin w = wins[bd.win];
  if (w.w >= 8) tw_render_body<true>(w, bd, canvas, canvas_w);
  else tw_render_body<false>(w, bd, canvas, canvas_w);
}

// pred_bin of merge_mask_list (:85-89): 3x3 cross erosion of the window's mask, > 60 -> 255
__device__ __forceinline__ bool pred_on(const TWin& w, int x, int y) {
  int m = w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x];
  if (y > 0) m = min(m, (int)w.mask[(size_t)(w.y1 + y - 1) * w.mask_w + w.x1 + x]);
  if (y + 1 < w.h) m = min(m, (int)w.mask[(size_t)(w.y1 + y + 1) * w.mask_w + w.x1 + x]);
  if (x > 0) m = min(m, (int)w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x - 1]);
  if (x + 1 < w.w) m = min(m, (int)w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x + 1]);
  return m > 60;
}

// four labels of a group (0 beyond the window's right edge)
__device__ __forceinline__ void load_lab4(const int* row, const Grp& g, int* l) {
  if (g.nv == 4) {
    __builtin_memcpy(l, row, 16);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = k < g.nv ? row[k] : 0;
  }
}

// pred_on for the four pixels of a group: bit k set = predicted text
__device__ __forceinline__ unsigned pred_on4(const TWin& w, const Grp& g) {
  const uint8_t* base = w.mask + (size_t)(w.y1 + g.y) * w.mask_w + w.x1;
  uint8_t c[6];
  load_row6(base, g.x, w.w, 255, c);
  int m[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = min((int)c[k + 1], min((int)c[k], (int)c[k + 2]));
#pragma unroll
  for (int dy = -1; dy <= 1; dy += 2) {
    const int yy = g.y + dy;
    if (yy < 0 || yy >= w.h) continue;
    const uint8_t* r = base + (long long)dy * w.mask_w + g.x;
    uint8_t v[4];
    if (g.nv == 4) {
      __builtin_memcpy(v, r, 4);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = k < g.nv ? r[k] : 255;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) m[k] = min(m[k], (int)v[k]);
  }
  unsigned bits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) bits |= (m[k] > 60 ? 1u : 0u) << k;
  return bits;
}

// counters[2l] / [2l+1]: pixels of component l not merged yet that are predicted text / background.
// A thread owns four consecutive pixels, consecutive lanes consecutive groups.  Threads whose pixels all
// carry one (label, prediction) key -- the inside of a component -- are merged into wave-level runs that
// their first lane adds once (a candidate's big components would otherwise serialise on two words);
// threads on a component's edge add their pixels one by one.  Either way the additions go to a per-block LDS table
// first (open addressing on the key, a few probes, global atomic as the fallback) and reach HBM once per block and key:
// stroke-like candidates are mostly edges, i.e. up to four L2 atomics per thread (rocprofv3: 81 us per launch of this
// kernel alone against 9 us for the apply kernel that walks the same pixels).
constexpr int TWA_HN = 1024;
__global__ __launch_bounds__(256) void tw_accept_count_kernel(const TWin* __restrict__ wins, const TBand* __restrict__ bands,
                                                              int round, const int* __restrict__ labels, int canvas_w,
                                                              int max_labels, const uint8_t* __restrict__ merged,
                                                              int merged_w, unsigned* __restrict__ counters) {
  const TBand bd = bands[blockIdx.y];
  if (round >= 0 && bd.round != round) return;
  const TWin w = wins[bd.win];
  const int ng = win_groups(w);
  if ((int)blockIdx.x * 256 >= ng) return;          // block-uniform: nothing to count, nothing to flush
  const int lane = threadIdx.x & 63;
  __shared__ int hkey[TWA_HN];
  __shared__ unsigned hcnt[TWA_HN];
  for (int i = threadIdx.x; i < TWA_HN; i += 256) hkey[i] = 0, hcnt[i] = 0;
  __syncthreads();
  auto add = [&](int key, unsigned n) {
    unsigned h = ((unsigned)key * 2654435761u) >> 22;          // 10 bits
#pragma unroll
    for (int probe = 0; probe < 4; ++probe, h = (h + 1) & (TWA_HN - 1)) {
      const int prev = atomicCAS(&hkey[h], 0, key);
      if (prev == 0 || prev == key) {
        atomicAdd(&hcnt[h], n);
        return;
      }
    }
    atomicAdd(counters + (size_t)key, n);
  };
  for (int g0 = blockIdx.x * 256; g0 < ng; g0 += gridDim.x * 256) {
    const int gi = g0 + threadIdx.x;
    int ukey = 0, ulen = 0;                         // this thread's contribution to a wave-level run
    if (gi < ng) {
      const Grp g = win_group(w, gi);
      int l[4];
      load_lab4(labels + (size_t)(bd.cy + g.y) * canvas_w + bd.cx + g.x, g, l);
      const uint8_t* mr = merged + (size_t)(w.my + g.y) * merged_w + w.mx + g.x;
      uint8_t mg[4];
      if (g.nv == 4) {
        __builtin_memcpy(mg, mr, 4);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) mg[k] = k < g.nv ? mr[k] : 255;
      }
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!(k < g.nv && l[k] > 0 && l[k] <= max_labels && mg[k] == 0)) l[k] = 0;
        any |= l[k] != 0;
      }
      if (any) {
        const unsigned pred = pred_on4(w, g);
        int key[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) key[k] = l[k] ? 2 * l[k] + (((pred >> k) & 1u) ? 0 : 1) : 0;
        bool same = true;
#pragma unroll
        for (int k = 1; k < 4; ++k) same &= k >= g.nv || key[k] == key[0];
        if (same) {
          ukey = key[0], ulen = g.nv;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (key[k]) add(key[k], 1u);
        }
      }
    }
    if (!__ballot(ukey != 0)) continue;
    const int prev = __shfl_up(ukey, 1);
    const bool head = lane == 0 || prev != ukey;
    const unsigned long long heads = __ballot(head);
    int ps = ulen;
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(ps, off);
      if (lane >= off) ps += u;
    }
    const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int lanes = later ? __ffsll((long long)later) : 64 - lane;
    const int tail = __shfl(ps, lane + lanes - 1);
    if (head && ukey) add(ukey, (unsigned)(tail - ps + ulen));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TWA_HN; i += 256)
    if (hkey[i]) atomicAdd(counters + (size_t)hkey[i], hcnt[i]);
}

// OR a component into the merged mask iff its bbox has >= min_box pixels (:98-99) and it lowers the
// xor distance to pred_bin (on > off, the reference's `xor_merged < xor_origin`)
__global__ __launch_bounds__(256) void tw_accept_apply_kernel(const TWin* __restrict__ wins, const TBand* __restrict__ bands,
                                                              int round, const int* __restrict__ labels, int canvas_w,
                                                              const int* __restrict__ stats, int max_labels, int min_box,
                                                              uint8_t* __restrict__ merged, int merged_w,
                                                              const unsigned* __restrict__ counters) {
  const TBand bd = bands[blockIdx.y];
  if (round >= 0 && bd.round != round) return;
  const TWin w = wins[bd.win];
  const int ng = win_groups(w);
  for (int gi = blockIdx.x * 256 + threadIdx.x; gi < ng; gi += gridDim.x * 256) {
    const Grp g = win_group(w, gi);
    int l[4];
    load_lab4(labels + (size_t)(bd.cy + g.y) * canvas_w + bd.cx + g.x, g, l);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lk = l[k];
      if (lk <= 0 || lk > max_labels) continue;
      const bool ok = stats[(size_t)(lk - 1) * 5 + 2] * stats[(size_t)(lk - 1) * 5 + 3] >= min_box;
      if (ok && counters[2 * (size_t)lk] > counters[2 * (size_t)lk + 1])
        merged[(size_t)(w.my + g.y) * merged_w + w.mx + g.x + k] = 255;
    }
  }
}

// ---- one block per window: every merge round of `merge_mask_list` (reference utils/textmask.py:73-104) -------------------
// The per-round launches above walk every window of the batch with a grid sized for the largest one, twice per round
// (count, apply), 2-4 rounds: 5-9 launches per work item whose time is the largest window's at three blocks (rocprofv3,
// round 4: 3.7 ms of kernel time per step at 29 windows per page, the biggest entry of the tail).  A window's rounds
// depend on each other (a round counts the pixels the earlier rounds have NOT merged), different windows do not: one block
// of 512 threads takes a window through all of its bands.  Counters live in an LDS table per band (2048 slots, open
// addressing; keys that do not fit go to the global counters, which stay zero otherwise), so the decision of a component
// = LDS entry + global entry.  Only this block touches the window's pixels of `merged`: __syncthreads orders its rounds.
constexpr int TWB_THREADS = 512, TWB_HN = 2048;
// Groups a thread has in flight per sweep step.  A window is one block's work, so its time is a chain of dependent loads
// (labels -> prediction rows -> table, labels -> stats -> counters) times the steps of the sweep: one group per thread and
// step made the largest window of a work item take 0.6 ms (rocprofv3, round 5: the biggest entry of the tail's GPU time,
// and a block that holds 16 KB of LDS on its CU for all of it).  Every sweep below first issues the loads of TWB_U groups,
// then uses them.  Sums, maxima and ORs: the order of the additions does not change a counter.
constexpr int TWB_U = 4;

// A group cut by the window's right edge takes a byte path in `load_lab4` / `pred_on4`: a branch whose loads are consumed
// inside it, i.e. a memory round trip in the middle of every wave that spans a row end (most do) -- with it the groups of a
// step wait for each other again.  In windows at least 8 pixels wide (`FAST`: the kernels pick the instantiation per
// window) the last group of a row is the row's LAST four pixels instead: it overlaps its left neighbour, and its first `lo`
// pixels -- the neighbour's -- are skipped.  Every group is then four whole pixels inside the window: one 16-B / 4-B / 8-B
// load each, no edge branch.  The sweeps only add, take maxima and set bytes per pixel, and every pixel is still taken
// exactly once.
// four bytes of a row from column g.x on, little-endian in one word; 255 beyond the window's right edge
__device__ __forceinline__ unsigned bytes4(const uint8_t* row0, const Grp& g) {
  uint8_t b[4];
  if (g.nv == 4) {
    __builtin_memcpy(b, row0 + g.x, 4);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = k < g.nv ? row0[g.x + k] : (uint8_t)255;
  }
  unsigned v;
  __builtin_memcpy(&v, b, 4);
  return v;
}
// pred_on4 of a whole group in a window at least 8 wide, without a byte path and in two halves: the loads of several
// groups are issued before the first is used
struct PredRaw {
  unsigned long long mid;   // 8 bytes of the group's row from column clamp(x - 1, 0, w - 8) on
  unsigned up, dn;          // columns x .. x+3 of the rows above / below (the group's own row where there is none)
};
__device__ __forceinline__ PredRaw pred_load_wide(const TWin& w, const Grp& g) {
  const uint8_t* base = w.mask + (size_t)(w.y1 + g.y) * w.mask_w + w.x1;
  PredRaw r;
  __builtin_memcpy(&r.mid, base + min(max(g.x - 1, 0), w.w - 8), 8);
  __builtin_memcpy(&r.up, base + (g.y > 0 ? -(long long)w.mask_w : 0ll) + g.x, 4);
  __builtin_memcpy(&r.dn, base + (g.y + 1 < w.h ? (long long)w.mask_w : 0ll) + g.x, 4);
  return r;
}
__device__ __forceinline__ unsigned pred_bits_wide(const TWin& w, const Grp& g, const PredRaw& r) {
  // columns x-1 .. x+4 as bytes 0..5; 255 outside the window
  const int d = g.x - 1 - min(max(g.x - 1, 0), w.w - 8);       // -1 (the first group of a row) .. 3
  unsigned long long v = d < 0 ? ((r.mid << 8) | 0xffull) : (r.mid >> (8 * (d & 7)));
  const int nvalid = w.w - (g.x - 1);                          // bytes from byte 0 on that are window columns (>= 5)
  v |= nvalid < 8 ? ~0ull << (8 * (nvalid & 7)) : 0ull;
  int c[6], m[4];
#pragma unroll
  for (int j = 0; j < 6; ++j) c[j] = (int)((v >> (8 * j)) & 0xffull);
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = min(c[k + 1], min(c[k], c[k + 2]));
  // a row outside the window is ignored: its stand-in is the group's own row, already in the minimum
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = min(m[k], min((int)((r.up >> (8 * k)) & 0xffu), (int)((r.dn >> (8 * k)) & 0xffu)));
  unsigned bits = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) bits |= (m[k] > 60 ? 1u : 0u) << k;
  return bits;
}

template <bool FAST>
__device__ __forceinline__ void tw_accept_all_body(const TWin& w, const TBand* __restrict__ bands, const int* __restrict__ labels,
                                                   int canvas_w, const int* __restrict__ stats, int max_labels, int min_box,
                                                   uint8_t* __restrict__ merged, int merged_w, unsigned* __restrict__ counters,
                                                   int* hkey, unsigned* hcnt) {
  const int ng = win_groups(w);
  const int lane = threadIdx.x & 63;
  auto add = [&](int key, unsigned n) {
    unsigned h = ((unsigned)key * 2654435761u) >> 21;          // 11 bits
#pragma unroll
    for (int probe = 0; probe < 4; ++probe, h = (h + 1) & (TWB_HN - 1)) {
      const int prev = atomicCAS(&hkey[h], 0, key);
      if (prev == 0 || prev == key) {
        atomicAdd(&hcnt[h], n);
        return;
      }
    }
    atomicAdd(counters + (size_t)key, n);
  };
  auto in_lds = [&](int key) -> unsigned {                    // the LDS entry of a key (0: none)
    unsigned h = ((unsigned)key * 2654435761u) >> 21;
#pragma unroll
    for (int probe = 0; probe < 4; ++probe, h = (h + 1) & (TWB_HN - 1)) {
      const int k = hkey[h];
      if (k == key) return hcnt[h];
      if (k == 0) break;
    }
    return 0u;
  };
  auto in_hbm = [&](int key) -> unsigned {                    // what overflowed to the global table
    return __hip_atomic_load(counters + (size_t)key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int r = 0; r < w.nband; ++r) {
    const TBand bd = bands[w.band0 + r];
    for (int i = threadIdx.x; i < TWB_HN; i += TWB_THREADS) hkey[i] = 0, hcnt[i] = 0;
    __syncthreads();                                           // also: the previous round's writes to `merged` are visible
    // ---- count: pixels of every component of this band not merged yet, split by the prediction (as tw_accept_count)
    for (int g0 = 0; g0 < ng; g0 += TWB_THREADS * TWB_U) {
      Grp g[TWB_U];
      int lo[TWB_U];                                           // pixels lo .. nv-1 of the group are this thread's
      int l[TWB_U][4];
      unsigned mg[TWB_U];                                      // four bytes of `merged`, 255 beyond the window's right edge
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        const int gi = g0 + u * TWB_THREADS + threadIdx.x;
        g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo[u]);
        load_lab4(labels + (size_t)(bd.cy + g[u].y) * canvas_w + bd.cx + g[u].x, g[u], l[u]);
        mg[u] = bytes4(merged + (size_t)(w.my + g[u].y) * merged_w + w.mx, g[u]);
        if (gi >= ng) lo[u] = 4;                               // a group beyond the window counts nothing
      }
      unsigned pred[TWB_U];
      PredRaw raw[TWB_U];
      bool any[TWB_U];
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        any[u] = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // (`&`, not `&&`: one mask per pixel instead of a chain of exec-mask branches; k >= nv: the byte is 255)
          const bool keep = (k >= lo[u]) & (l[u][k] > 0) & (l[u][k] <= max_labels) & (((mg[u] >> (8 * k)) & 0xffu) == 0);
          l[u][k] = keep ? l[u][k] : 0;
          any[u] |= keep;
        }
      }
      // the prediction around the groups that have a pixel to count: again every group's loads before the first use
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        pred[u] = 0;
        raw[u].mid = 0, raw[u].up = raw[u].dn = 0;
        if (any[u]) {
          if (FAST) raw[u] = pred_load_wide(w, g[u]);
          else pred[u] = pred_on4(w, g[u]);
        }
      }
      if (FAST) {
#pragma unroll
        for (int u = 0; u < TWB_U; ++u) pred[u] = pred_bits_wide(w, g[u], raw[u]);   // unused where the group has no key
      }
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        int ukey = 0, ulen = 0;                                // this thread's contribution to a wave-level run
        if (l[u][0] | l[u][1] | l[u][2] | l[u][3]) {
          int key[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) key[k] = l[u][k] ? 2 * l[u][k] + (((pred[u] >> k) & 1u) ? 0 : 1) : 0;
          const int last = g[u].nv - 1;                        // the group's last pixel is always this thread's
          const int kl = last == 3 ? key[3] : (last == 2 ? key[2] : (last == 1 ? key[1] : key[0]));
          bool same = true;
#pragma unroll
          for (int k = 0; k < 4; ++k) same &= k < lo[u] || k >= g[u].nv || key[k] == kl;
          if (same) {
            ukey = kl, ulen = g[u].nv - lo[u];
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (key[k]) add(key[k], 1u);
          }
        }
        if (!__ballot(ukey != 0)) continue;
        const int prev = __shfl_up(ukey, 1);
        const bool head = lane == 0 || prev != ukey;
        const unsigned long long heads = __ballot(head);
        int ps = ulen;
        for (int off = 1; off < 64; off <<= 1) {
          const int v = __shfl_up(ps, off);
          if (lane >= off) ps += v;
        }
        const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int lanes = later ? __ffsll((long long)later) : 64 - lane;
        const int tail = __shfl(ps, lane + lanes - 1);
        if (head && ukey) add(ukey, (unsigned)(tail - ps + ulen));
      }
    }
    __syncthreads();
    // ---- apply: OR a component in iff its bbox has >= min_box pixels and it lowers the xor distance (as tw_accept_apply)
    for (int g0 = 0; g0 < ng; g0 += TWB_THREADS * TWB_U) {
      Grp g[TWB_U];
      int lo[TWB_U];
      int l[TWB_U][4];
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        const int gi = g0 + u * TWB_THREADS + threadIdx.x;
        g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo[u]);
        load_lab4(labels + (size_t)(bd.cy + g[u].y) * canvas_w + bd.cx + g[u].x, g[u], l[u]);
        if (gi >= ng) lo[u] = 4;
      }
#pragma unroll
      for (int u = 0; u < TWB_U; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) l[u][k] = ((k >= lo[u]) & (l[u][k] > 0) & (l[u][k] <= max_labels)) ? l[u][k] : 0;
      // the first component of every group (neighbours mostly share theirs) and what decides it, fetched for all groups
      // before any is used (component 1's entries stand in where a group has none)
      int l0[TWB_U], bw[TWB_U], bh[TWB_U];
      unsigned on[TWB_U], off[TWB_U];
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        l0[u] = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k)
          if (l[u][k]) l0[u] = l[u][k];
        const int li = max(l0[u], 1);
        bw[u] = stats[(size_t)(li - 1) * 5 + 2], bh[u] = stats[(size_t)(li - 1) * 5 + 3];
        on[u] = in_hbm(2 * li), off[u] = in_hbm(2 * li + 1);
      }
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        if (!l0[u]) continue;
        int lprev = l0[u];
        bool okprev = bw[u] * bh[u] >= min_box && on[u] + in_lds(2 * lprev) > off[u] + in_lds(2 * lprev + 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int lk = l[u][k];
          if (!lk) continue;
          if (lk != lprev) {                                   // a second component in one group: decide it here
            lprev = lk;
            okprev = stats[(size_t)(lk - 1) * 5 + 2] * stats[(size_t)(lk - 1) * 5 + 3] >= min_box &&
                     in_hbm(2 * lk) + in_lds(2 * lk) > in_hbm(2 * lk + 1) + in_lds(2 * lk + 1);
          }
          if (okprev) merged[(size_t)(w.my + g[u].y) * merged_w + w.mx + g[u].x + k] = 255;
        }
      }
    }
    __syncthreads();                                           // before the table is cleared for the next band
  }
}

__global__ __launch_bounds__(TWB_THREADS) void tw_accept_all_kernel(const TWin* __restrict__ wins, const TBand* __restrict__ bands,
                                                                    const int* __restrict__ labels, int canvas_w,
                                                                    const int* __restrict__ stats, int max_labels, int min_box,
                                                                    uint8_t* __restrict__ merged, int merged_w,
                                                                    unsigned* __restrict__ counters) {
  __shared__ int hkey[TWB_HN];
  __shared__ unsigned hcnt[TWB_HN];
  const TWin w = wins[blockIdx.x];
  if (w.w >= 8) tw_accept_all_body<true>(w, bands, labels, canvas_w, stats, max_labels, min_box, merged, merged_w, counters, hkey, hcnt);
  else tw_accept_all_body<false>(w, bands, labels, canvas_w, stats, max_labels, min_box, merged, merged_w, counters, hkey, hcnt);
}

// ---- one block per window: the four passes of the hole filling (reference utils/textmask.py:113-131) ---------------------
template <bool FAST>
__device__ __forceinline__ void tw_holes_all_body(const TWin& w, const int* __restrict__ labels2, const int* __restrict__ stats2,
                                                  const int* __restrict__ first2, int max_labels,
                                                  const unsigned* __restrict__ count255, uint8_t* __restrict__ merged,
                                                  int merged_w, unsigned* __restrict__ counters2, int* tp) {
  const int ng = win_groups(w);
  if (threadIdx.x < 3) tp[threadIdx.x] = -1;
  __syncthreads();
  const int a0 = (int)count255[blockIdx.x];                    // the background entry: pixels already set
  for (int pass = 0; pass < 4; ++pass) {
    if (pass <= 1 && threadIdx.x == 0) {
      if (pass == 0) atomicMax(&tp[0], a0);
      else if (a0 == tp[0]) atomicAdd(&tp[1], 1);
      else atomicMax(&tp[2], a0);
    }
    const int m1 = pass >= 1 ? tp[0] : 0;
    const int thr = pass >= 2 ? (tp[1] >= 1 ? tp[0] : tp[2]) : 0;
    if (pass >= 2 && thr < 0) break;                           // block-uniform: nothing can be filled
    for (int g0 = 0; g0 < ng; g0 += TWB_THREADS * TWB_U) {
      Grp g[TWB_U];
      int lo[TWB_U];
      int lab[TWB_U][4];
      size_t c0[TWB_U];
#pragma unroll
      for (int u = 0; u < TWB_U; ++u) {
        const int gi = g0 + u * TWB_THREADS + threadIdx.x;
        g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo[u]);
        c0[u] = (size_t)(w.my + g[u].y) * merged_w + w.mx + g[u].x;
        load_lab4(labels2 + c0[u], g[u], lab[u]);
        if (gi >= ng) lo[u] = 4;
      }
#pragma unroll
      for (int u = 0; u < TWB_U; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) lab[u][k] = ((k >= lo[u]) & (lab[u][k] > 0) & (lab[u][k] <= max_labels)) ? lab[u][k] : 0;
      // per pixel: the component's area and (passes 0, 1) its representative pixel -- all of them fetched before the first use
      int area[TWB_U][4], rep[TWB_U][4];
#pragma unroll
      for (int u = 0; u < TWB_U; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int l = lab[u][k];
          area[u][k] = l ? stats2[(size_t)(l - 1) * 5 + 4] : 0;
          rep[u][k] = l && pass <= 1 ? first2[l - 1] : -1;
        }
#pragma unroll
      for (int u = 0; u < TWB_U; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int l = lab[u][k];
          if (!l) continue;
          const size_t ci = c0[u] + k;
          const int ar = area[u][k];
          if (pass <= 1) {
            if (rep[u][k] != (int)ci) continue;                // one representative pixel per component
            if (pass == 0) atomicMax(&tp[0], ar);
            else if (ar == m1) atomicAdd(&tp[1], 1);
            else atomicMax(&tp[2], ar);
          } else if (pass == 2) {
            if (ar < thr && merged[ci] == 0) atomicAdd(counters2 + 2 * (size_t)l + (pred_on(w, g[u].x + k, g[u].y) ? 0 : 1), 1u);
          } else {
            if (ar < thr &&
                __hip_atomic_load(counters2 + 2 * (size_t)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >
                    __hip_atomic_load(counters2 + 2 * (size_t)l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
              merged[ci] = 255;
          }
        }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(TWB_THREADS) void tw_holes_all_kernel(const TWin* __restrict__ wins, const int* __restrict__ labels2,
                                                                   const int* __restrict__ stats2, const int* __restrict__ first2,
                                                                   int max_labels, const unsigned* __restrict__ count255,
                                                                   uint8_t* __restrict__ merged, int merged_w,
                                                                   unsigned* __restrict__ counters2) {
  __shared__ int tp[3];                                        // [maximum, multiplicity - 1, runner-up]
  const TWin w = wins[blockIdx.x];
  if (w.w >= 8) tw_holes_all_body<true>(w, labels2, stats2, first2, max_labels, count255, merged, merged_w, counters2, tp);
  else tw_holes_all_body<false>(w, labels2, stats2, first2, max_labels, count255, merged, merged_w, counters2, tp);
}

// 3x3 rect dilation inside the window (REFINEMASK_INPAINT, textmask.py:110-111) or a copy; also the
// complement canvas for the hole-filling labelling (:113) and the count of set pixels per window
template <bool FAST>
__device__ __forceinline__ unsigned tw_dilate_body(const TWin& w, const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                   uint8_t* __restrict__ comp, int merged_w, int dilate) {
  constexpr int TW_U = 1;                                        // one group: three 8-byte rows in flight
  unsigned cnt = 0;
  const int ng = win_groups(w);
  for (int g0 = blockIdx.x * 256 * TW_U; g0 < ng; g0 += gridDim.x * 256 * TW_U) {
    Grp g[TW_U];
    int lo[TW_U];
    bool live[TW_U];
    unsigned long long rows[TW_U][3];
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      const int gi = g0 + u * 256 + threadIdx.x;
      live[u] = gi < ng;
      g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo[u]);
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {                         // rows outside the window (and, without dilation, the neighbours) add nothing
        const bool ok = (dy == 0 || dilate) && (unsigned)(g[u].y + dy) < (unsigned)w.h;
        const unsigned long long r = row6<FAST>(in + (size_t)(w.my + g[u].y + (ok ? dy : 0)) * merged_w + w.mx, g[u].x, w.w, 0);
        rows[u][dy + 1] = ok ? r : 0ull;
      }
    }
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      if (!live[u]) continue;
      int m[4] = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int a = (int)((rows[u][r] >> (8 * k)) & 0xff), b = (int)((rows[u][r] >> (8 * k + 8)) & 0xff), c = (int)((rows[u][r] >> (8 * k + 16)) & 0xff);
          m[k] = max(m[k], dilate ? max(a, max(b, c)) : b);
        }
      uint8_t o[4], c[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = (uint8_t)m[k];
        c[k] = (uint8_t)(255 - m[k]);
        cnt += k >= lo[u] && k < g[u].nv && m[k] == 255;
      }
      const size_t at = (size_t)(w.my + g[u].y) * merged_w + w.mx + g[u].x;
      if (g[u].nv == 4) {                                        // (an overlapping group stores its neighbour's pixels again: same values)
        __builtin_memcpy(out + at, o, 4);
        __builtin_memcpy(comp + at, c, 4);
      } else {
        for (int k = 0; k < g[u].nv; ++k) out[at + k] = o[k], comp[at + k] = c[k];
      }
    }
  }
  return cnt;
}
__global__ __launch_bounds__(256) void tw_dilate_kernel(const TWin* __restrict__ wins, const uint8_t* __restrict__ in,
                                                        uint8_t* __restrict__ out, uint8_t* __restrict__ comp, int merged_w,
                                                        unsigned* __restrict__ count255, int dilate) {
  __shared__ unsigned red[4];
  const TWin w = wins[blockIdx.y];
  unsigned cnt = w.w >= 8 ? tw_dilate_body<true>(w, in, out, comp, merged_w, dilate) : tw_dilate_body<false>(w, in, out, comp, merged_w, dilate);
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned s = red[0] + red[1] + red[2] + red[3];
    if (s) atomicAdd(count255 + blockIdx.y, s);
  }
}

// Hole filling (:113-131): components of the complement with area < the second largest entry of
// sorted({set-pixel count} U {component areas}) may be OR-ed in.  pass 0: maximum; pass 1: multiplicity
// of the maximum and the runner-up; pass 2: on / off counters; pass 3: apply.
// top2 (n,3) = [maximum, multiplicity - 1, runner-up], all three initialised to -1 by the caller
// (one memset); a window without any entry besides its background keeps runner-up -1: nothing is filled.
__global__ __launch_bounds__(256) void tw_holes_kernel(const TWin* __restrict__ wins, int pass, const int* __restrict__ labels2,
                                                       const int* __restrict__ stats2, const int* __restrict__ first2,
                                                       int max_labels, const unsigned* __restrict__ count255,
                                                       int* __restrict__ top2, uint8_t* __restrict__ merged, int merged_w,
                                                       unsigned* __restrict__ counters2) {
  const TWin w = wins[blockIdx.y];
  int* tp = top2 + (size_t)blockIdx.y * 3;
  if (pass <= 1 && blockIdx.x == 0 && threadIdx.x == 0) {      // the background entry: pixels already set
    const int a = (int)count255[blockIdx.y];
    if (pass == 0) atomicMax(tp, a);
    else if (a == tp[0]) atomicAdd(tp + 1, 1);
    else atomicMax(tp + 2, a);
  }
  const int m1 = pass >= 1 ? tp[0] : 0;
  const int thr = pass >= 2 ? (tp[1] >= 1 ? tp[0] : tp[2]) : 0;
  const int ng = win_groups(w);
  for (int gi = blockIdx.x * 256 + threadIdx.x; gi < ng; gi += gridDim.x * 256) {
    const Grp g = win_group(w, gi);
    const size_t c0 = (size_t)(w.my + g.y) * merged_w + w.mx + g.x;
    int lab[4];
    load_lab4(labels2 + c0, g, lab);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int l = lab[k];
      if (l <= 0 || l > max_labels) continue;
      const size_t ci = c0 + k;
      const int area = stats2[(size_t)(l - 1) * 5 + 4];
      if (pass <= 1) {
        if (first2[l - 1] != (int)ci) continue;                  // one representative pixel per component
        if (pass == 0) atomicMax(tp, area);
        else if (area == m1) atomicAdd(tp + 1, 1);
        else atomicMax(tp + 2, area);
      } else if (pass == 2) {
        // only components that may be filled are counted: the big ones (the window's real background) would
        // serialise every pixel of the window on two counters
        if (area < thr && merged[ci] == 0) atomicAdd(counters2 + 2 * (size_t)l + (pred_on(w, g.x + k, g.y) ? 0 : 1), 1u);
      } else {
        if (area < thr && counters2[2 * (size_t)l] > counters2[2 * (size_t)l + 1]) merged[ci] = 255;
      }
    }
  }
}

// refined[y1:y2, x1:x2] |= merged (textmask.py:167); windows may overlap -> word-wide atomic OR
template <bool FAST>
__device__ __forceinline__ void tw_commit_body(const TWin& w, const uint8_t* __restrict__ merged, int merged_w) {
  const int ng = win_groups(w);
  for (int g0 = blockIdx.x * 256 * TW_U; g0 < ng; g0 += gridDim.x * 256 * TW_U) {
    Grp g[TW_U];
    unsigned v[TW_U];
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      const int gi = g0 + u * 256 + threadIdx.x;
      int lo;
      g[u] = win_group_lo<FAST>(w, min(gi, ng - 1), lo);      // (an overlapping group ORs its neighbour's pixels again)
      const uint8_t* src = merged + (size_t)(w.my + g[u].y) * merged_w + w.mx + g[u].x;
      v[u] = 0;
      if (g[u].nv == 4) {
        __builtin_memcpy(&v[u], src, 4);
      } else {
        for (int k = 0; k < g[u].nv; ++k) v[u] |= (unsigned)src[k] << (8 * k);
      }
      if (gi >= ng) v[u] = 0;
    }
#pragma unroll
    for (int u = 0; u < TW_U; ++u) {
      if (!v[u]) continue;
      const size_t idx = (size_t)(w.y1 + g[u].y) * w.out_w + w.x1 + g[u].x;
      unsigned* a = (unsigned*)(w.out + (idx & ~(size_t)3));
      // the page buffers are 4-byte aligned and padded to a multiple of 4 bytes: the group covers at most two words
      const unsigned long long v2 = (unsigned long long)v[u] << (8 * (idx & 3));
      if ((unsigned)v2) atomicOr(a, (unsigned)v2);
      if ((unsigned)(v2 >> 32)) atomicOr(a + 1, (unsigned)(v2 >> 32));
    }
  }
}
__global__ __launch_bounds__(256) void tw_commit_kernel(const TWin* __restrict__ wins, const uint8_t* __restrict__ merged,
                                                        int merged_w) {
  const TWin w = wins[blockIdx.y];
  if (w.w >= 8) tw_commit_body<true>(w, merged, merged_w);
  else tw_commit_body<false>(w, merged, merged_w);
}

__global__ void mask_clear_where_kernel(uint8_t* __restrict__ mask, const uint8_t* __restrict__ refined, long long n, int thr) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if ((int)refined[i] > thr) mask[i] = 0;
}

__global__ void copy2d_u8_kernel(const uint8_t* __restrict__ src, int spitch, uint8_t* __restrict__ dst, int dpitch, int rows,
                                 int cols) {
  const long long n = (long long)rows * cols;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(i / cols), x = (int)(i % cols);
    dst[(size_t)y * dpitch + x] = src[(size_t)y * spitch + x];
  }
}

inline int win_gx(int max_pix) { return max(1, min(64, (max_pix + 4095) / 4096)); }
// x extent of a (gx, n) window grid under the block cap (the window kernels stride over their window in x)
inline int win_gx(int max_pix, int n) { return max(1, min(win_gx(max_pix), g_tail_max_blocks / max(1, n))); }

// ---- one launch for many copies / fills -----------------------------------------------------------------------------
// The tail moves dozens of small tables per batch (counts, contour tables, window / rule / band tables, histograms,
// page masks) and zeroes a dozen scratch buffers.  As hipMemcpyAsync / hipMemsetAsync each of them is a blit-kernel
// launch of its own (rocprofv3, round 2: 64 copyBuffer + 52 fillBuffer launches per step, 2.3 ms of kernel time per
// 12.8 ms step).  Here a stage's copies and fills are segments of ONE kernel: device <-> device, device -> page-locked
// host and page-locked host -> device alike (hipHostMalloc'ed memory is device accessible), linear or 2-D.
template <typename V>
__device__ __forceinline__ void mseg_run(const MSeg& g, long long first, long long step) {
  const long long rowv = (long long)(g.row_bytes / sizeof(V));
  const long long total = rowv * g.rows;
  V fillv;
  if (!g.src) __builtin_memset(&fillv, g.fill, sizeof(V));
  for (long long i = first; i < total; i += step) {
    long long r = 0, c = i;
    if (g.rows > 1) r = i / rowv, c = i - r * rowv;
    V* d = (V*)((char*)g.dst + r * g.dpitch) + c;
    if (g.src) *d = *((const V*)((const char*)g.src + r * g.spitch) + c);
    else *d = fillv;
  }
}

__global__ __launch_bounds__(256) void multi_copy_kernel(MSegs m) {
  int s = 0;
  while (s + 1 < m.n && (int)blockIdx.x >= m.blk_off[s + 1]) ++s;     // <= 24 segments: a scalar loop
  const MSeg& g = m.s[s];
  const int nb = m.blk_off[s + 1] - m.blk_off[s];
  const long long first = (long long)(blockIdx.x - m.blk_off[s]) * 256 + threadIdx.x, step = (long long)nb * 256;
  if (g.vec == 16) mseg_run<uint4>(g, first, step);
  else if (g.vec == 4) mseg_run<unsigned>(g, first, step);
  else mseg_run<uint8_t>(g, first, step);
}

// counters[2 l], [2 l + 1] of the labels that exist (1 .. min(*n_labels, cap)) -> 0.  The table is sized for the bound on the
// number of components (a quarter of the canvas' pixels, 8 B each: 0.23 GB of zero-fill per 32 pages when it was cleared
// whole); the labelling has left the actual count on the device.
__global__ void label_counters_zero_kernel(unsigned* __restrict__ counters, const int* __restrict__ n_labels, int cap) {
  const long long total = 2ll * (min(*n_labels, cap) + 1);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) counters[i] = 0;
}

}  // namespace

void launch_multi_copy(MSegs& m, hipStream_t st) {
  if (m.n <= 0) return;
  int off = 0;
  for (int i = 0; i < m.n; ++i) {
    MSeg& g = m.s[i];
    const unsigned long long a = (unsigned long long)(uintptr_t)g.dst | (unsigned long long)(uintptr_t)g.src | g.row_bytes |
                                 (g.rows > 1 ? (unsigned long long)g.dpitch | (unsigned long long)g.spitch : 0ull);
    g.vec = (a & 15) == 0 ? 16 : ((a & 3) == 0 ? 4 : 1);
    const unsigned long long elems = g.row_bytes / g.vec * (unsigned long long)g.rows;
    // 8 elements per thread, at most 128 blocks per segment: a PCIe-bound segment needs few waves in flight, and the
    // forward running next to the tail should keep its CUs
    const int nb = (int)std::min<unsigned long long>(128, std::max<unsigned long long>(1, (elems + 2047) / 2048));
    m.blk_off[i] = off;
    off += nb;
  }
  m.blk_off[m.n] = off;
  hipLaunchKernelGGL(multi_copy_kernel, dim3(off), dim3(256), 0, st, m);
  m.n = 0;
}

void launch_dbc(const DbcTables& t, hipStream_t st) {
  hipLaunchKernelGGL(dbc_prep_kernel, dim3((t.cap + 255) / 256, t.B), dim3(256), 0, st, t);
  hipLaunchKernelGGL(dbc_scan_kernel, dim3(t.B), dim3(256), 0, st, t);
  hipLaunchKernelGGL(dbc_init_kernel, dim3(grid_for((long long)t.B * t.rcap)), dim3(256), 0, st, t);
  const int per_page = std::max(1, std::min((t.H * t.W + 255) / 256, g_tail_max_blocks / std::max(1, t.B)));
  hipLaunchKernelGGL(dbc_accum_kernel, dim3(per_page, t.B), dim3(256), 0, st, t);
}

void launch_tw_hist(const TWin* wins, int n, int max_pix, unsigned* hist, hipStream_t st) {
  hipLaunchKernelGGL(tw_hist_kernel, dim3(win_gx(max_pix, n), n), dim3(256), 0, st, wins, hist);
}

void launch_tw_xor(const TWin* wins, const TRule* rules, int n, int max_pix, unsigned long long* sums, hipStream_t st) {
  hipLaunchKernelGGL(tw_xor_kernel, dim3(win_gx(max_pix, n), n), dim3(256), 0, st, wins, rules, sums);
}

void launch_tw_render(const TWin* wins, const TBand* bands, int nbands, int max_pix, uint8_t* canvas, int canvas_w,
                      hipStream_t st) {
  hipLaunchKernelGGL(tw_render_kernel, dim3(win_gx(max_pix, nbands), nbands), dim3(256), 0, st, wins, bands, canvas, canvas_w);
}

void launch_tw_accept(const TWin* wins, const TBand* bands, int nbands, int max_pix, int round, const int* labels,
                      int canvas_w, const int* stats, int max_labels, int min_box, uint8_t* merged, int merged_w,
                      unsigned* counters, hipStream_t st) {
  const dim3 g(win_gx(max_pix, nbands), nbands);
  hipLaunchKernelGGL(tw_accept_count_kernel, g, dim3(256), 0, st, wins, bands, round, labels, canvas_w, max_labels, merged,
                     merged_w, counters);
  hipLaunchKernelGGL(tw_accept_apply_kernel, g, dim3(256), 0, st, wins, bands, round, labels, canvas_w, stats, max_labels,
                     min_box, merged, merged_w, counters);
}

void launch_label_counters_zero(unsigned* counters, const int* n_labels, int cap, hipStream_t st) {
  hipLaunchKernelGGL(label_counters_zero_kernel, dim3(64), dim3(256), 0, st, counters, n_labels, cap);
}

void launch_tw_accept_all(const TWin* wins, const TBand* bands, int n, const int* labels, int canvas_w, const int* stats,
                          int max_labels, int min_box, uint8_t* merged, int merged_w, unsigned* counters, hipStream_t st) {
  hipLaunchKernelGGL(tw_accept_all_kernel, dim3(n), dim3(TWB_THREADS), 0, st, wins, bands, labels, canvas_w, stats, max_labels,
                     min_box, merged, merged_w, counters);
}

void launch_tw_holes_all(const TWin* wins, int n, const int* labels2, const int* stats2, const int* first2, int max_labels,
                         const unsigned* count255, uint8_t* merged, int merged_w, unsigned* counters2, hipStream_t st) {
  hipLaunchKernelGGL(tw_holes_all_kernel, dim3(n), dim3(TWB_THREADS), 0, st, wins, labels2, stats2, first2, max_labels, count255,
                     merged, merged_w, counters2);
}

void launch_tw_dilate(const TWin* wins, int n, int max_pix, const uint8_t* in, uint8_t* out, uint8_t* comp, int merged_w,
                      unsigned* count255, int dilate, hipStream_t st) {
  hipLaunchKernelGGL(tw_dilate_kernel, dim3(win_gx(max_pix, n), n), dim3(256), 0, st, wins, in, out, comp, merged_w, count255,
                     dilate);
}

void launch_tw_holes(const TWin* wins, int n, int max_pix, const int* labels2, const int* stats2, const int* first2,
                     int max_labels, const unsigned* count255, int* top2, uint8_t* merged, int merged_w,
                     unsigned* counters2, hipStream_t st) {
  const dim3 g(win_gx(max_pix, n), n);
  for (int pass = 0; pass < 4; ++pass)
    hipLaunchKernelGGL(tw_holes_kernel, g, dim3(256), 0, st, wins, pass, labels2, stats2, first2, max_labels, count255, top2,
                       merged, merged_w, counters2);
}

void launch_tw_commit(const TWin* wins, int n, int max_pix, const uint8_t* merged, int merged_w, hipStream_t st) {
  hipLaunchKernelGGL(tw_commit_kernel, dim3(win_gx(max_pix, n), n), dim3(256), 0, st, wins, merged, merged_w);
}

void launch_mask_clear_where(uint8_t* mask, const uint8_t* refined, long long n, int thr, hipStream_t st) {
  hipLaunchKernelGGL(mask_clear_where_kernel, dim3(grid_for(n)), dim3(256), 0, st, mask, refined, n, thr);
}

void launch_copy2d_u8(const uint8_t* src, int spitch, uint8_t* dst, int dpitch, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(copy2d_u8_kernel, dim3(grid_for((long long)rows * cols)), dim3(256), 0, st, src, spitch, dst, dpitch,
                     rows, cols);
}
