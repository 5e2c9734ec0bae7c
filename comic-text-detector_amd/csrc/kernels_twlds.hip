// merge_mask_list for ONE window in ONE block, on bit planes in LDS (gfx950) -- reference utils/textmask.py:73-132 with
// refine_mask's loop body around it (:159-169): render a candidate, label its 8-connected components, OR in every
// component that lowers the xor distance to the eroded prediction, next candidate; dilate; fill holes; OR the window into
// the page's refined mask.
//
// Why: as separate passes over packed canvases (kernels_tail.hip `tw_render / tw_accept_all / tw_dilate / tw_holes_all /
// tw_commit` around two launches of the page-scale labelling, kernels_post.hip) this stage wrote and re-read int32 label
// planes of every candidate of every window: 4.6 of the tail's 8.7 ms of kernel time and 2.3 of its 3.9 GB per 32 pages
// (DESIGN 4.12).  Nothing in it needs a label PLANE.  What a merge round needs is, per component, one number:
//
//     diff(C) = |C & ~merged & pred| - |C & ~merged & ~pred|        accepted iff diff > 0
//
// (the reference compares xor sums over the component's bounding box before / after OR-ing it in: they differ on the
// component's not-yet-merged pixels only; components of one candidate are disjoint, so a round's decisions are
// independent), plus `w * h >= 3` for its bounding box -- which only excludes components of ONE pixel or of TWO pixels side
// by side / on top of each other, a local bit pattern that is removed from the candidate before it is labelled.
//
// Data: a window is H rows of wp = ceil(W / 32) words, bit x & 31 of word x >> 5, bits beyond W zero.  Planes `pred`
// (3x3 cross erosion of the prediction > 60), `merged`, `cand`.  A RUN is a maximal horizontal stretch of set pixels; its
// id is the number of run starts before it in raster order, so the run that covers any set pixel is
//     base[word] + popcount(starts[word] & bits_up_to(pixel)) - 1
// with base = exclusive prefix sum of popcount(starts) -- no run table, no scan along a run.  Everything else is
// word-local: a word's groups of set bits against the word above it and ONE halo bit either side give the 8-connected
// unions (union-find over run ids, LDS compare-and-swap, smaller id wins), the weights, and the bits to OR in.
//
// LDS: 3 planes + base (u16 per word) + parent / acc (one u32 each per run, `rcap` runs).  A candidate (or the complement)
// with more runs than `rcap` raises the window's overflow flag and commits nothing: the host sends that window through the
// canvas path (tail.hip), as it does up front for windows whose planes do not fit.
//
// tests/twlds_emul.py is this file word for word in Python, checked against the oracle's pixel-level merge_mask_list
// (tests/test_twlds_emul.py, CPU); tests/test_gpu_e2e.py compares the kernel itself with the oracle and with the canvas
// path byte for byte.
#include <algorithm>

#include "kernels.h"
#include "tail.h"

namespace {

constexpr int TL_MAXT = 1024;   // threads per block: 256 / 512 / 1024 ("tail_lds_threads"), a launch parameter

struct Geo {
  int W, H, wp, words;
  unsigned last;   // valid bits of a row's last word
};

__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// three consecutive words of a row (zeros outside the window) and the row shifted by K pixels: result[x] = row[x + K]
struct Row3 {
  unsigned prev, cur, nxt;
};
__device__ __forceinline__ Row3 row3(const unsigned* p, const Geo& g, int y, int wi) {
  Row3 r{0u, 0u, 0u};
  if ((unsigned)y < (unsigned)g.H) {
    const unsigned* q = p + y * g.wp + wi;
    r.cur = q[0];
    if (wi > 0) r.prev = q[-1];
    if (wi + 1 < g.wp) r.nxt = q[1];
  }
  return r;
}
template <int K> __device__ __forceinline__ unsigned shk(const Row3& r) {
  if constexpr (K == 0) return r.cur;
  else if constexpr (K > 0) return (r.cur >> K) | (r.nxt << (32 - K));
  else return (r.cur << -K) | (r.prev >> (32 + K));
}
__device__ __forceinline__ unsigned ring3(const Row3& r) { return shk<-1>(r) | r.cur | shk<1>(r); }

// bits 0 .. p of a word
__device__ __forceinline__ unsigned upto(int p) { return ~(~1u << p); }

// n valid bytes (n <= 32) from p as eight words, zero padded -- never reads beyond p + n
__device__ __forceinline__ void load32(const uint8_t* p, int n, unsigned* v) {
  if (n >= 32) {
    __builtin_memcpy(v, p, 16);
    __builtin_memcpy(v + 4, p + 16, 16);
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned x = 0;
    if (4 * j + 4 <= n) {
      __builtin_memcpy(&x, p + 4 * j, 4);
    } else {
      for (int k = 0; k < 3; ++k)
        if (4 * j + k < n) x |= (unsigned)p[4 * j + k] << (8 * k);
    }
    v[j] = x;
  }
}

// bit k of the result: byte k of the four bytes of x is > 60   (SWAR: bit 7 of (b & 127) + 67 or of b itself)
__device__ __forceinline__ unsigned gt60_nibble(unsigned x) {
  const unsigned t = (((x & 0x7f7f7f7fu) + 0x43434343u) | x) & 0x80808080u;
  const unsigned g = t >> 7;
  return (g | (g >> 7) | (g >> 14) | (g >> 21)) & 0xfu;
}

__device__ __forceinline__ int gray3(unsigned b, unsigned g, unsigned r) {
  // OpenCV 4.x RGB2Gray<uchar>: 15-bit coefficients, round to nearest (kernels_tail.hip gray_of)
  return (int)((b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15);
}

// union-find over run ids in LDS: the smaller id is the root
__device__ __forceinline__ unsigned uf_find(unsigned* parent, unsigned a) {
  for (;;) {
    const unsigned p = lds_ld(parent + a);
    if (p == a) return a;
    const unsigned gp = lds_ld(parent + p);
    if (gp != p) lds_st(parent + a, gp);      // path halving: any ancestor is a valid parent
    a = gp;
  }
}
__device__ __forceinline__ void uf_union(unsigned* parent, unsigned a, unsigned b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const unsigned t = a;
      a = b;
      b = t;
    }
    if (atomicCAS(parent + a, a, b) == a) return;
  }
}

struct TLdsArgs {
  const TWin* wins;
  const TBand* bands;
  const int* order;     // window indices of this launch
  int dilate;           // REFINEMASK_INPAINT: 3x3 dilation before the hole filling
  int max_words, rlay;  // LDS layout of the launch: plane words, run-table entries (2 * rlay >= max_words: the scratch plane)
  int rcap;             // runs a labelling may have (<= rlay)
  int* ovf;             // per window: 1 = a run table overflowed, nothing committed
};

__global__ __launch_bounds__(TL_MAXT) void tw_lds_kernel(TLdsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int TL_NT = blockDim.x, NWV = TL_NT >> 6;
  const int widx = a.order[blockIdx.x];
  const TWin w = a.wins[widx];
  Geo g;
  g.W = w.w, g.H = w.h, g.wp = (w.w + 31) >> 5, g.words = g.wp * g.H;
  g.last = (w.w & 31) ? ((1u << (w.w & 31)) - 1u) : 0xffffffffu;
  const int MW = a.max_words, RL = a.rlay, RC = a.rcap;
  unsigned* const pred = lds;
  unsigned* const merged = lds + MW;
  unsigned* const cand = lds + 2 * MW;
  unsigned short* const base = (unsigned short*)(lds + 3 * MW);
  unsigned* const parent = lds + 3 * MW + ((MW + 1) >> 1);
  int* const acc = (int*)(parent + RL);
  unsigned* const red = parent + 2 * RL;           // [0..15] wave partials, [16..18] top-2 table
  unsigned* const tmp = parent;                    // a scratch plane where no run table is live (2 * rlay >= max_words)
  auto validw = [&](int wi) { return wi == g.wp - 1 ? g.last : 0xffffffffu; };
  // every word i = t, t + NT, ... of the window with its (row, word in row): one division per thread for the whole kernel
  // (a pass has 2-4 words per thread and there are ~45 passes: `i / wp` per word was a tenth of the kernel's instructions)
  const int y_t = t / g.wp, wi_t = t - y_t * g.wp;
  const int dy_t = TL_NT / g.wp, dwi_t = TL_NT - dy_t * g.wp;
  auto for_words = [&](auto&& fn) {
    int y = y_t, wi = wi_t;
    for (int i = t; i < g.words; i += TL_NT) {
      fn(i, y, wi);
      wi += dwi_t, y += dy_t;
      if (wi >= g.wp) wi -= g.wp, ++y;
    }
  };

  // ---- the prediction: b = mask > 60, then the 3x3 cross erosion as AND of the five neighbours (outside the window: ones)
  for_words([&](int i, int y, int wi) {
    unsigned v[8];
    load32(w.mask + (size_t)(w.y1 + y) * w.mask_w + w.x1 + 32 * wi, g.W - 32 * wi, v);
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) bits |= gt60_nibble(v[j]) << (4 * j);
    tmp[i] = bits & validw(wi);
    merged[i] = 0;
  });
  __syncthreads();
  for_words([&](int i, int y, int wi) {
    auto ext = [&](int yy, int ww) -> unsigned {
      if ((unsigned)yy >= (unsigned)g.H || (unsigned)ww >= (unsigned)g.wp) return 0xffffffffu;
      return tmp[yy * g.wp + ww] | ~validw(ww);
    };
    const unsigned cur = ext(y, wi), prev = ext(y, wi - 1), nxt = ext(y, wi + 1);
    pred[i] = cur & ((cur << 1) | (prev >> 31)) & ((cur >> 1) | (nxt << 31)) & ext(y - 1, wi) & ext(y + 1, wi) & validw(wi);
  });
  __syncthreads();

  // ---- block-wide pieces shared by the candidate rounds and the hole filling -------------------------------------------
  // run starts of `cand` -> base[] (exclusive prefix of their counts, raster order); returns the number of runs
  const int seg = (g.words + TL_NT - 1) / TL_NT;
  auto starts_of = [&](int i, int wi) -> unsigned {
    const unsigned c = cand[i], cl = wi > 0 ? cand[i - 1] : 0u;
    return c & ~((c << 1) | (cl >> 31));
  };
  auto scan_runs = [&]() -> int {
    const int i0 = min(t * seg, g.words), i1 = min(i0 + seg, g.words);
    int cnt = 0;
    {
      int wi = i0 % g.wp;
      for (int i = i0; i < i1; ++i) {
        cnt += __popc(starts_of(i, wi));
        if (++wi == g.wp) wi = 0;
      }
    }
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    if (lane == 63) red[wave] = (unsigned)inc;
    __syncthreads();
    int before = inc - cnt;
    int total = 0;
    for (int k = 0; k < NWV; ++k) {
      const int v = (int)red[k];
      before += k < wave ? v : 0;
      total += v;
    }
    {
      int wi = i0 % g.wp, run = before;
      for (int i = i0; i < i1; ++i) {
        base[i] = (unsigned short)run;
        const int n = __popc(starts_of(i, wi));
        for (int k = run; k < run + n && k < RL; ++k) parent[k] = (unsigned)k, acc[k] = 0;   // the runs that start in this word
        run += n;
        if (++wi == g.wp) wi = 0;
      }
    }
    __syncthreads();
    return total;
  };
  // the run covering set pixel p of a word with run starts `s` and prefix `b`
  auto rid_from = [](unsigned b, unsigned s, int p) -> unsigned { return b + (unsigned)__popc(s & upto(p)) - 1u; };
  // every group of set bits of word i: fn(lowest bit, group mask)
  auto for_groups = [&](unsigned v, auto&& fn) {
    while (v) {
      const unsigned lb = v & (0u - v);
      const unsigned grp = v & ~(v + lb);
      fn(__ffs((int)v) - 1, grp);
      v &= ~grp;
    }
  };
  // 8-connected unions of every word-local group with the row above (one halo bit either side) + fn(i, grp, rid)
  auto link_and = [&](auto&& fn) {
    for_words([&](int i, int y, int wi) {
      const unsigned c = cand[i];
      if (!c) return;
      const unsigned cl = wi > 0 ? cand[i - 1] : 0u;
      const unsigned sc = c & ~((c << 1) | (cl >> 31)), bc = base[i];
      unsigned up = 0, ul = 0, ur = 0, su = 0, bu = 0;
      if (y > 0) {
        up = cand[i - g.wp];
        if (wi > 0) ul = cand[i - g.wp - 1];
        if (wi + 1 < g.wp) ur = cand[i - g.wp + 1];
        su = up & ~((up << 1) | (ul >> 31));
        bu = base[i - g.wp];
      }
      for_groups(c, [&](int p, unsigned grp) {
        const unsigned me = rid_from(bc, sc, p);
        fn(i, grp, me);
        if (y == 0) return;
        const unsigned an = up & (grp | (grp << 1) | (grp >> 1));
        unsigned reps = an & ~(an << 1);
        while (reps) {
          const int q = __ffs((int)reps) - 1;
          reps &= reps - 1;
          uf_union(parent, me, rid_from(bu, su, q));
        }
        if ((grp & 1u) && (ul >> 31)) uf_union(parent, me, bu - 1u);
        // the run covering bit 0 of the word above-right: it starts there unless the word above ends set
        if ((grp >> 31) && (ur & 1u)) uf_union(parent, me, (unsigned)base[i - g.wp + 1] + ((up >> 31) ? 0u : 1u) - 1u);
      });
    });
  };
  auto fold_to_roots = [&](int n) {                              // acc of every run into its root's
    for (int i = t; i < n; i += TL_NT) {
      const unsigned r = uf_find(parent, (unsigned)i);
      if (r != (unsigned)i) {
        const int v = acc[i];
        if (v) atomicAdd(acc + r, v);
      }
    }
    __syncthreads();
  };
  auto or_accepted = [&]() {                                     // merged |= groups whose root's acc > 0
    for_words([&](int i, int, int wi) {
      const unsigned c = cand[i];
      if (!c) return;
      const unsigned cl = wi > 0 ? cand[i - 1] : 0u;
      const unsigned sc = c & ~((c << 1) | (cl >> 31)), bc = base[i];
      unsigned add = 0;
      for_groups(c, [&](int p, unsigned grp) {
        if (acc[uf_find(parent, rid_from(bc, sc, p))] > 0) add |= grp;
      });
      if (add) merged[i] |= add;
    });
    __syncthreads();
  };
  bool overflow = false;

  // ================= the candidates, in merge order (reference :92-108) ==================================================
  for (int r = 0; r < w.nband && !overflow; ++r) {
    const TBand bd = a.bands[w.band0 + r];
    // ---- render: cv2.inRange(grey, lo, hi) / threshold(channel, lo), the polarity minxor_thresh picked
    for_words([&](int i, int y, int wi) {
      const int npx = min(32, g.W - 32 * wi);
      const uint8_t* src = w.img + ((size_t)(w.y1 + y) * w.img_w + w.x1 + 32 * wi) * 3;
      unsigned v[24];
      load32(src, min(32, 3 * npx), v);
      load32(src + 32, min(32, max(0, 3 * npx - 32)), v + 8);
      load32(src + 64, min(32, max(0, 3 * npx - 64)), v + 16);
      unsigned bits = 0;
      auto byte_at = [&](int o) { return (v[o >> 2] >> (8 * (o & 3))) & 0xffu; };
      if (bd.kind == 0) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const int gr = gray3(byte_at(3 * k), byte_at(3 * k + 1), byte_at(3 * k + 2));
          bits |= ((gr >= bd.lo && gr <= bd.hi) ? 1u : 0u) << k;
        }
      } else {
        const int ch = bd.kind - 1;                            // B, G, R
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          const unsigned c0 = byte_at(3 * k), c1 = byte_at(3 * k + 1), c2 = byte_at(3 * k + 2);
          bits |= (((int)(ch == 0 ? c0 : (ch == 1 ? c1 : c2)) > bd.lo) ? 1u : 0u) << k;
        }
      }
      if (bd.invert) bits = ~bits;
      tmp[i] = bits & validw(wi);
    });
    __syncthreads();
    // ---- minus the components whose bounding box has fewer than 3 pixels (:97): 1x1, 2x1, 1x2
    for_words([&](int i, int y, int wi) {
      const Row3 c0 = row3(tmp, g, y, wi);
      unsigned out = c0.cur;
      if (out) {
        const Row3 m1 = row3(tmp, g, y - 1, wi), p1 = row3(tmp, g, y + 1, wi), m2 = row3(tmp, g, y - 2, wi), p2 = row3(tmp, g, y + 2, wi);
        const unsigned up = ring3(m1), dn = ring3(p1), cur = c0.cur;
        const unsigned single = cur & ~(shk<-1>(c0) | shk<1>(c0) | up | dn);
        const unsigned hl = cur & shk<1>(c0) & ~(shk<-1>(c0) | shk<2>(c0) | up | dn | shk<2>(m1) | shk<2>(p1));
        const unsigned hr = cur & shk<-1>(c0) & ~(shk<-2>(c0) | shk<1>(c0) | up | dn | shk<-2>(m1) | shk<-2>(p1));
        const unsigned vt = cur & p1.cur & ~(shk<-1>(c0) | shk<1>(c0) | shk<-1>(p1) | shk<1>(p1) | up | ring3(p2));
        const unsigned vb = cur & m1.cur & ~(shk<-1>(c0) | shk<1>(c0) | shk<-1>(m1) | shk<1>(m1) | dn | ring3(m2));
        out = cur & ~(single | hl | hr | vt | vb);
      }
      cand[i] = out;
    });
    __syncthreads();
    const int nr = scan_runs();
    if (nr > RC) {
      overflow = true;
      break;
    }
    // ---- unions + every group's weight: not-yet-merged pixels that are predicted text minus those that are not
    link_and([&](int i, unsigned grp, unsigned me) {
      const unsigned nm = grp & ~merged[i];
      const int wgt = __popc(nm & pred[i]) - __popc(nm & ~pred[i]);
      if (wgt) atomicAdd(acc + me, wgt);
    });
    __syncthreads();
    fold_to_roots(nr);
    or_accepted();
  }

  // ================= dilation (:110-111), hole filling (:113-131) =========================================================
  if (!overflow) {
    if (a.dilate) {
      for_words([&](int i, int y, int wi) {
        unsigned v = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
          const Row3 r = row3(merged, g, y + dy, wi);
          v |= ring3(r);
        }
        tmp[i] = v & validw(wi);
      });
      __syncthreads();
    }
    unsigned cnt = 0;
    for_words([&](int i, int, int wi) {
      const unsigned m = a.dilate ? tmp[i] : merged[i];
      merged[i] = m;
      cand[i] = ~m & validw(wi);
      cnt += (unsigned)__popc(m);
    });
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    __syncthreads();                                           // tmp (= parent) is free; red[] was last read before the previous barrier
    if (lane == 0) red[wave] = cnt;
    if (t == 0) red[16] = 0u, red[17] = 0u, red[18] = 0u;      // [max + 1, multiplicity, runner-up + 1] (0 = none)
    __syncthreads();
    unsigned a0 = 0;                                           // the background entry of the sorted areas: pixels already set
    for (int k = 0; k < NWV; ++k) a0 += red[k];
    __syncthreads();                                           // (scan_runs writes red[0..3] again)
    const int nr = scan_runs();
    if (nr > RC) {
      overflow = true;
    } else {
      link_and([&](int, unsigned grp, unsigned me) { atomicAdd(acc + me, __popc(grp)); });
      __syncthreads();
      fold_to_roots(nr);
      // area threshold = second largest of {a0} U {areas of the components} (sorted_area[-2])
      for (int i = t; i < nr + 1; i += TL_NT) {
        const bool is = i == nr || parent[i] == (unsigned)i;
        if (is) atomicMax(red + 16, (i == nr ? a0 : (unsigned)acc[i]) + 1u);
      }
      __syncthreads();
      const unsigned mx = red[16] - 1u;
      for (int i = t; i < nr + 1; i += TL_NT) {
        const bool is = i == nr || parent[i] == (unsigned)i;
        if (!is) continue;
        const unsigned ar = i == nr ? a0 : (unsigned)acc[i];
        if (ar == mx) atomicAdd(red + 17, 1u);
        else atomicMax(red + 18, ar + 1u);
      }
      __syncthreads();
      const long long thr = red[17] >= 2u ? (long long)mx : (long long)red[18] - 1;   // -1: nothing can be filled
      if (thr >= 0) {
        for (int i = t; i < nr; i += TL_NT)
          if (parent[i] == (unsigned)i) acc[i] = (long long)acc[i] < thr ? 0 : -(1 << 30);
        __syncthreads();
        for_words([&](int i, int, int wi) {
          const unsigned c = cand[i];
          if (!c) return;
          const unsigned cl = wi > 0 ? cand[i - 1] : 0u;
          const unsigned sc = c & ~((c << 1) | (cl >> 31)), bc = base[i];
          for_groups(c, [&](int p, unsigned grp) {
            const int wgt = __popc(grp & pred[i]) - __popc(grp & ~pred[i]);
            if (wgt) atomicAdd(acc + uf_find(parent, rid_from(bc, sc, p)), wgt);
          });
        });
        __syncthreads();
        or_accepted();
      }
    }
  }
  if (overflow) {
    if (t == 0) a.ovf[widx] = 1;
    return;
  }
  // ================= refined[y1:y2, x1:x2] |= merged (:167); windows may overlap -> word-wide atomic OR =================
  for_words([&](int i, int y, int wi) {
    const unsigned m = merged[i];
    if (!m) return;
    const size_t idx0 = (size_t)(w.y1 + y) * w.out_w + w.x1 + 32 * wi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned nib = (m >> (4 * j)) & 0xfu;
      if (!nib) continue;
      const unsigned v = ((nib & 1u) ? 0xffu : 0u) | ((nib & 2u) ? 0xff00u : 0u) | ((nib & 4u) ? 0xff0000u : 0u) |
                         ((nib & 8u) ? 0xff000000u : 0u);
      const size_t idx = idx0 + 4 * j;
      unsigned* ap = (unsigned*)(w.out + (idx & ~(size_t)3));
      // the page buffers are 4-byte aligned and padded to a multiple of 4 bytes: the four pixels cover at most two words
      const unsigned long long v2 = (unsigned long long)v << (8 * (idx & 3));
      if ((unsigned)v2) atomicOr(ap, (unsigned)v2);
      if ((unsigned)(v2 >> 32)) atomicOr(ap + 1, (unsigned)(v2 >> 32));
    }
  });
}

}  // namespace

// bytes of dynamic LDS for windows of at most `max_words` plane words and `rcap` runs
size_t tw_lds_bytes(int max_words, int rcap) {
  const int rlay = std::max(rcap, (max_words + 1) / 2);
  return ((size_t)3 * max_words + (max_words + 1) / 2 + (size_t)2 * rlay + 20) * 4;
}

// the run capacity a launch gets for its largest window: `g_tw_lds_runs_x10` / 10 runs per word (default 2.5: text strokes on
// the reference's real page reach 1.7), at least 1024, at most what a u16 prefix holds
int g_tw_lds_runs_x10 = 25;     // "tail_lds_runs_x10"
int g_tw_lds_threads = 512;     // "tail_lds_threads": 256 / 512 / 1024 threads per window (512: a third off the merge wait of a serial tail; end to end the three are equal, profiles/r06_twlds_knobs.txt)
int tw_lds_rcap(int max_words) { return std::min(65000, std::max(1024, (int)((long long)max_words * g_tw_lds_runs_x10 / 10))); }

bool launch_tw_lds(const TWin* wins, const TBand* bands, const int* order, int n, int max_words, int rcap, int dilate, int* ovf,
                   hipStream_t st) {
  if (n <= 0) return true;
  TLdsArgs a;
  a.wins = wins, a.bands = bands, a.order = order, a.dilate = dilate, a.max_words = max_words;
  a.rcap = std::min(rcap, 65000), a.rlay = std::max(a.rcap, (max_words + 1) / 2);
  a.ovf = ovf;
  const size_t bytes = tw_lds_bytes(max_words, a.rcap);
  if (bytes > 48 * 1024 &&                                     // more dynamic LDS than the default limit: the kernel has to ask
      hipFuncSetAttribute((const void*)tw_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    (void)hipGetLastError();
    return false;                                              // the caller sends these windows through the canvas path
  }
  const int nt = g_tw_lds_threads >= 1024 ? 1024 : (g_tw_lds_threads >= 512 ? 512 : 256);
  hipLaunchKernelGGL(tw_lds_kernel, dim3(n), dim3(nt), bytes, st, a);
  return true;
}
