// Post-processing kernels: class-aware greedy NMS and connected-component
// labelling with statistics.  Integer / comparison work, HBM- and latency-bound.
#include <algorithm>

#include "kernels.h"

// The labelling kernels of the tail run NEXT TO the network's forward of the following batch.  A neighbour that fills every
// wave slot (8 waves per SIMD of short blocks) stretches the forward's VALU-bound kernels 2.5-4x; at <= 4 waves per SIMD
// they keep their speed (selftest ST_CORUN).  So the big-grid tail kernels are launched with at most this many blocks of
// 256 threads (measured: no effect on the end-to-end rate between 768 and no cap, 7 % lower at 384, 30 % at 192 where the tail becomes the bottleneck) and walk their tiles; "tail_max_blocks".
int g_tail_max_blocks = 1024;

namespace {

// ===========================================================================
// NMS  (reference utils/yolov5_utils.py:124-218 `non_max_suppression`,
//       multi_label=False, agnostic=False; torchvision.ops.nms at :202)
// ===========================================================================
struct Cand {
  float x1, y1, x2, y2;  // un-offset xyxy
  float score;
  int cls;
  int idx;               // original row (tie-break: lower row first)
  int alive;
};

// stage 1: obj > conf (:136,155), conf = obj*cls (:171), best class (:181),
// conf > conf_thres (:182), xywh -> xyxy (:174, :220-227); compact per page.
__global__ void nms_filter_kernel(const float* __restrict__ blks, int B, int rows, int no, float conf_thres,
                                  Cand* __restrict__ cands, int* __restrict__ counts) {
  const long long total = (long long)B * rows;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / rows), r = (int)(i % rows);
    const float* x = blks + i * no;
    const float obj = x[4];
    if (!(obj > conf_thres)) continue;
    float best = x[5] * obj;
    int bj = 0;
    for (int j = 1; j < no - 5; ++j) {
      const float c = x[5 + j] * obj;
      if (c > best) { best = c; bj = j; }
    }
    if (!(best > conf_thres)) continue;
    const int pos = atomicAdd(&counts[b], 1);
    Cand c;
    c.x1 = x[0] - x[2] / 2;
    c.y1 = x[1] - x[3] / 2;
    c.x2 = x[0] + x[2] / 2;
    c.y2 = x[1] + x[3] / 2;
    c.score = best;
    c.cls = bj;
    c.idx = r;
    c.alive = 1;
    cands[(size_t)b * rows + pos] = c;
  }
}

struct Best { float s; int idx; int pos; };
constexpr int NMS_THREADS = 256;   // 4 waves: a round's two barriers and its 4-entry argmax cost a quarter of the 16-wave block's
constexpr int NMS_REG = 8;         // candidates a thread of nms_greedy_kernel keeps in registers (2048 per page)

__device__ __forceinline__ bool better(const Best& a, const Best& b) {
  return a.s > b.s || (a.s == b.s && a.idx < b.idx);
}

__device__ __forceinline__ Best block_argmax(Best v, Best* sh) {
  for (int off = 32; off > 0; off >>= 1) {
    Best o;
    o.s = __shfl_down(v.s, off);
    o.idx = __shfl_down(v.idx, off);
    o.pos = __shfl_down(v.pos, off);
    if (better(o, v)) v = o;
  }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  Best r = sh[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i)
    if (better(sh[i], r)) r = sh[i];
  return r;
}

// stage 2: one block per page.  Greedy: repeatedly take the best alive candidate
// (score desc, row asc), suppress every alive candidate of IoU > thr computed on
// the class-offset boxes in fp32 exactly as torchvision's nms kernel does
// (ovr = inter / (iarea + area_j - inter), strict >), stop at max_det (:203-204).
__global__ __launch_bounds__(NMS_THREADS) void nms_greedy_kernel(Cand* __restrict__ cands_all, const int* __restrict__ counts,
                                                          int rows, float iou_thres, int max_det, int max_nms,
                                                          float max_wh, float* __restrict__ dets,
                                                          int* __restrict__ out_counts) {
  __shared__ Best sh[16];
  __shared__ unsigned cnt_sh;
  const int b = blockIdx.x;
  Cand* c = cands_all + (size_t)b * rows;
  const int n = counts[b];
  float* out = dets + (size_t)b * max_det * 6;

  // max_nms cap (:196-197): keep the max_nms highest scores (ties at the cut are all kept;
  // the reference's argsort leaves their order unspecified)
  if (n > max_nms) {
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned trial = prefix | (1u << bit);
      if (threadIdx.x == 0) cnt_sh = 0;
      __syncthreads();
      unsigned local = 0;
      for (int j = threadIdx.x; j < n; j += blockDim.x) local += __float_as_uint(c[j].score) >= trial;
      atomicAdd(&cnt_sh, local);
      __syncthreads();
      if (cnt_sh >= (unsigned)max_nms) prefix = trial;
      __syncthreads();
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x)
      if (__float_as_uint(c[j].score) < prefix) c[j].alive = 0;
    __syncthreads();
  }

  // Up to NMS_REG candidates per thread live in registers and the round's winner is handed round through LDS: a round
  // is then two barriers.  With the candidates in HBM every round was three dependent memory round trips (the winner, the
  // flags, the boxes) -- ~6 us a round, 0.19 ms for a page's ~30 kept boxes, on a block that fills a CU's wave slots.
  // Same comparisons and the same fp32 expressions on the same values, in the same order per pair.
  if (n <= NMS_REG * (int)blockDim.x) {
    __shared__ Cand ksh;
    Cand r[NMS_REG];
    Best mine{-1.f, 0x7fffffff, -1};
#pragma unroll
    for (int q = 0; q < NMS_REG; ++q) {
      const int j = threadIdx.x + q * blockDim.x;
      r[q].alive = 0;
      if (j < n) r[q] = c[j];
      if (r[q].alive) {
        Best t{r[q].score, r[q].idx, j};
        if (better(t, mine)) mine = t;
      }
    }
    int kept = 0;
    while (kept < max_det) {
      const Best top = block_argmax(mine, sh);
      if (top.pos < 0) break;
#pragma unroll
      for (int q = 0; q < NMS_REG; ++q)
        if (top.pos == (int)(threadIdx.x + q * blockDim.x)) ksh = r[q];
      __syncthreads();
      const Cand k = ksh;
      const float off = (float)k.cls * max_wh;
      const float ix1 = k.x1 + off, iy1 = k.y1 + off, ix2 = k.x2 + off, iy2 = k.y2 + off;
      const float iarea = (ix2 - ix1) * (iy2 - iy1);
      if (threadIdx.x == 0) {
        float* o = out + (size_t)kept * 6;
        o[0] = k.x1; o[1] = k.y1; o[2] = k.x2; o[3] = k.y2; o[4] = k.score; o[5] = (float)k.cls;
      }
      ++kept;
      mine = Best{-1.f, 0x7fffffff, -1};
#pragma unroll
      for (int q = 0; q < NMS_REG; ++q) {
        const int j = threadIdx.x + q * blockDim.x;
        if (!r[q].alive) continue;
        if (j == top.pos) { r[q].alive = 0; continue; }
        const float o2 = (float)r[q].cls * max_wh;
        const float jx1 = r[q].x1 + o2, jy1 = r[q].y1 + o2, jx2 = r[q].x2 + o2, jy2 = r[q].y2 + o2;
        const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
        const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (iarea + (jx2 - jx1) * (jy2 - jy1) - inter);
        if (ovr > iou_thres) { r[q].alive = 0; continue; }
        Best t{r[q].score, r[q].idx, j};
        if (better(t, mine)) mine = t;
      }
    }
    if (threadIdx.x == 0) out_counts[b] = kept;
    return;
  }

  Best mine{-1.f, 0x7fffffff, -1};
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    if (c[j].alive) {
      Best t{c[j].score, c[j].idx, j};
      if (better(t, mine)) mine = t;
    }
  int kept = 0;
  while (kept < max_det) {
    const Best top = block_argmax(mine, sh);
    if (top.pos < 0) break;
    const Cand k = c[top.pos];
    const float off = (float)k.cls * max_wh;
    const float ix1 = k.x1 + off, iy1 = k.y1 + off, ix2 = k.x2 + off, iy2 = k.y2 + off;
    const float iarea = (ix2 - ix1) * (iy2 - iy1);
    if (threadIdx.x == 0) {
      float* o = out + (size_t)kept * 6;
      o[0] = k.x1; o[1] = k.y1; o[2] = k.x2; o[3] = k.y2; o[4] = k.score; o[5] = (float)k.cls;
    }
    ++kept;
    __syncthreads();
    mine = Best{-1.f, 0x7fffffff, -1};
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      if (!c[j].alive) continue;
      if (j == top.pos) { c[j].alive = 0; continue; }
      const float o2 = (float)c[j].cls * max_wh;
      const float jx1 = c[j].x1 + o2, jy1 = c[j].y1 + o2, jx2 = c[j].x2 + o2, jy2 = c[j].y2 + o2;
      const float xx1 = fmaxf(ix1, jx1), yy1 = fmaxf(iy1, jy1);
      const float xx2 = fminf(ix2, jx2), yy2 = fminf(iy2, jy2);
      const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
      const float inter = w * h;
      const float ovr = inter / (iarea + (jx2 - jx1) * (jy2 - jy1) - inter);
      if (ovr > iou_thres) { c[j].alive = 0; continue; }
      Best t{c[j].score, c[j].idx, j};
      if (better(t, mine)) mine = t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out_counts[b] = kept;
}

// ===========================================================================
// CCL with stats (replaces cv2.connectedComponentsWithStats, reference
// utils/textmask.py:93,113,138).  Union-find over pixel indices with atomicMin
// links (root = smallest linear index of the component = its first pixel in
// raster order), then roots are ranked in raster order so label ids follow the
// first-pixel order (OpenCV SAUF numbering for 4-connectivity).
// ===========================================================================
// (agent scope: every thread that links or flattens a parent array runs on this GPU; the default system scope made each
// hop a load that bypasses the caches)
__device__ __forceinline__ int uf_load(const int* parent, int x) {
  return __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(int* parent, int x) {
  int p = uf_load(parent, x);
  while (p != x) {
    x = p;
    p = uf_load(parent, x);
  }
  return x;
}

__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
  while (true) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(parent + b, a);
    if (old == b) return;
    b = old;
  }
}

// Phase 1: tile-local labelling in LDS.  Each 32x32 tile is resolved with a
// union-find over LDS atomics (no HBM atomics, no cross-CU traffic); the result
// written to HBM is parent[p] = global index of p's tile-local root (the first
// pixel of its local component in raster order).
constexpr int CT = 32;  // tile edge

__device__ __forceinline__ int lds_find(int* lp, int x) {
  int p = __atomic_load_n(lp + x, __ATOMIC_RELAXED);
  while (p != x) {
    x = p;
    p = __atomic_load_n(lp + x, __ATOMIC_RELAXED);
  }
  return x;
}

__device__ __forceinline__ void lds_union(int* lp, int a, int b) {
  while (true) {
    a = lds_find(lp, a);
    b = lds_find(lp, b);
    if (a == b) return;
    if (a > b) { const int t = a; a = b; b = t; }
    const int old = atomicMin(lp + b, a);
    if (old == b) return;
    b = old;
  }
}

constexpr int RK_CHUNK = 4096;  // pixels ranked per block
constexpr int RK_PER_T = RK_CHUNK / 256;

// Run-based: a row of a tile is 32 pixels = half a wavefront, so the horizontal runs of a row come from
// one ballot -- every pixel is pre-labelled with the first pixel of its run without any atomic, and only run
// HEADS take part in the union-find, each with the (few) runs of the row above it overlaps.  The per-pixel
// version issued up to four LDS union chains per foreground pixel; on page backgrounds and window
// complements (runs of 32) that was 30x the work (rocprofv3: 0.36 ms per 32 Mpixel launch before).
template <int CONN>
__device__ __forceinline__ void ccl_local_body(int vb, const uint8_t* __restrict__ img, int* __restrict__ parent_all,
                                                        unsigned* __restrict__ colmask, int H, int W, int tiles_x, int tiles_y,
                                                        int thresh, int invert) {
  __shared__ int lp[CT * CT];
  __shared__ unsigned rowmask[CT];
  int bid = vb;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const size_t base = (size_t)b * H * W;
  const int x0 = tx * CT, y0 = ty * CT;
  const int lane = threadIdx.x & 63;
  const int lx = threadIdx.x & 31;
  // the tile's four rows of this thread first, then the ballots: a ballot consumes its load, and interleaved the four
  // were four memory round trips in a row (pixels outside the image read the image's first byte and are masked)
  int pix[CT * CT / 256];
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int gx = x0 + lx, gy = y0 + ((threadIdx.x + 256 * k) >> 5);
    pix[k] = img[base + (gx < W && gy < H ? (size_t)gy * W + gx : (size_t)0)];
  }
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int li = threadIdx.x + 256 * k;
    const int ly = li >> 5;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool fg = gx < W && gy < H && ((pix[k] > thresh) != (invert != 0));
    const unsigned long long bal = __ballot(fg);
    const unsigned m = (unsigned)(lane < 32 ? bal : bal >> 32);          // this row's 32 pixels
    // first pixel of my run: one past the highest clear bit below me
    const unsigned below = ~m & ((1u << lx) - 1u);
    const int start = below ? 32 - __clz(below) : 0;
    lp[li] = fg ? (ly << 5) + start : -1;
    if (lx == 0) rowmask[ly] = m;
  }
  __syncthreads();
  // the tile's first and last column as bit masks (bit ly = foreground): all the vertical-boundary kernel needs to know
  // where a link is due -- read from the parent plane, a column is one 64-B line per PIXEL (0.5 GB per 32 pages)
  if (threadIdx.x < 64) {
    const unsigned rm = rowmask[threadIdx.x & 31];
    const unsigned long long bl = __ballot(threadIdx.x < 32 && (rm & 1u)), br = __ballot(threadIdx.x < 32 && (rm >> 31));
    if (threadIdx.x == 0) {
      unsigned* cm = colmask + ((size_t)b * tiles_y * tiles_x + (size_t)ty * tiles_x + tx) * 2;
      cm[0] = (unsigned)bl, cm[1] = (unsigned)br;
    }
  }
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int li = threadIdx.x + 256 * k;
    const int ly = li >> 5;
    if (ly == 0) continue;
    const unsigned m = rowmask[ly], ma = rowmask[ly - 1];
    // run heads only -- decided from the row mask: lp[] is already being rewritten by other heads' unions
    if (!((m >> lx) & 1u) || (lx > 0 && ((m >> (lx - 1)) & 1u))) continue;
    // my run [lx, end]: the set bits of m from lx up to the next clear bit
    const unsigned from = m >> lx;
    const int len = (~from) ? __ffs(~from) - 1 : 32 - lx;
    unsigned run = (len >= 32 ? 0xffffffffu : ((1u << len) - 1u)) << lx;
    if (CONN == 8) run |= (run << 1) | (run >> 1);                       // diagonal neighbours in the row above
    unsigned ov = ma & run;
    while (ov) {
      const int bpos = __ffs(ov) - 1;
      const unsigned belowa = ~ma & ((1u << bpos) - 1u);
      const int sa = belowa ? 32 - __clz(belowa) : 0;                    // head of that run in the row above
      lds_union(lp, li, ((ly - 1) << 5) + sa);
      const unsigned froma = ma >> bpos;                                 // clear the rest of that run
      const int lena = (~froma) ? __ffs(~froma) - 1 : 32 - bpos;
      ov &= ~((lena >= 32 ? 0xffffffffu : ((1u << lena) - 1u)) << bpos);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int li = threadIdx.x + 256 * k;
    const int gx = x0 + lx, gy = y0 + (li >> 5);
    if (gx >= W || gy >= H) continue;
    int v = -1;
    if (lp[li] >= 0) {
      const int r = lds_find(lp, lp[li]);
      v = (y0 + (r >> 5)) * W + x0 + (r & 31);
    }
    parent_all[base + (size_t)gy * W + gx] = v;
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
template <int CONN>
__global__ __launch_bounds__(256) void ccl_local_kernel(const uint8_t* __restrict__ img, int* __restrict__ parent_all,
                                                        unsigned* __restrict__ colmask, int H, int W, int tiles_x, int tiles_y,
                                                        int thresh, int invert, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl_local_body<CONN>(vb, img, parent_all, colmask, H, W, tiles_x, tiles_y, thresh, invert);
    __syncthreads();
  }
}

// Phase 2: merge across tile borders with HBM atomics.  Only border pixels take part (~3/32 of the image)
// and every chain starts at a tile-local root.  Threads are enumerated over the border lines themselves
// (blockIdx.y = which tile boundary, blockIdx.z = image): a flat grid-stride loop over all pixels spent its
// time on 64-bit divisions to find the 3 border pixels in 32 (rocprofv3: 0.45-1.0 ms per launch at 32 x 1024^2).
//   horizontal boundary y = CT*k : pixel (x, y) with the row above: (x, y-1) and, 8-connected, (x-1, y-1), (x+1, y-1)
//   vertical boundary   x = CT*k : pixel (x, y) with (x-1, y) and, 8-connected, (x-1, y-1); and (x-1, y) with (x, y-1)
template <int CONN>
__global__ __launch_bounds__(256) void ccl_border_h_kernel(int* __restrict__ parent_all, int H, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = CT * (blockIdx.y + 1);
  if (x >= W || y >= H) return;
  int* parent = parent_all + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  if (parent[p] < 0) return;
  // Only one link per pair of overlapping runs: horizontally adjacent foreground pixels are already one
  // set (tile-local labelling, or the vertical-boundary kernel, whose pruning never relies on this one),
  // so when the left neighbour is foreground it has made -- or inherited -- the links to the row above.
  const bool q = x > 0 && parent[p - 1] >= 0;
  const bool up = parent[p - W] >= 0;
  if (CONN == 4) {
    if (up && !(q && parent[p - W - 1] >= 0)) uf_union(parent, p, p - W);
  } else {
    const bool ur = x + 1 < W && parent[p - W + 1] >= 0;
    if (q) {
      if (ur && !up) uf_union(parent, p, p - W + 1);
    } else if (up) {
      uf_union(parent, p, p - W);
    } else {
      if (x > 0 && parent[p - W - 1] >= 0) uf_union(parent, p, p - W - 1);
      if (ur) uf_union(parent, p, p - W + 1);
    }
  }
}

template <int CONN>
__global__ __launch_bounds__(256) void ccl_border_v_kernel(int* __restrict__ parent_all, const unsigned* __restrict__ colmask,
                                                           int H, int W, int tiles_x, int tiles_y) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y;                              // boundary between tile columns k and k + 1
  const int x = CT * (k + 1);
  if (y >= H || x >= W) return;
  int* parent = parent_all + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  // which of the four pixels around (x, y) are foreground: from the column masks the tile-local kernel left (bit = row of
  // the tile; [0] first column, [1] last column), not from the parent plane
  const unsigned* cm = colmask + (size_t)blockIdx.z * tiles_y * tiles_x * 2;
  auto fg = [&](int yy, int tile_x, int side) { return ((cm[((size_t)(yy >> 5) * tiles_x + tile_x) * 2 + side] >> (yy & 31)) & 1u) != 0; };
  const bool me = fg(y, k + 1, 0), left = fg(y, k, 1);
  if (!me && !left) return;
  // Rows y-1 and y of one tile: vertical neighbours are already one set, so the 2x2 block needs a link only
  // where the row above does not provide the connection.  On a horizontal tile boundary (y % CT == 0) nothing
  // is assumed about the row above (that is the other kernel's job) and every adjacent pair is linked.
  const bool ua = y > 0, same_tile = (y % CT) != 0;
  const bool mu = ua && fg(y - 1, k + 1, 0), lu = ua && fg(y - 1, k, 1);
  if (!same_tile) {
    if (me && left) uf_union(parent, p, p - 1);
    if (CONN == 8) {
      if (me && lu) uf_union(parent, p, p - W - 1);
      if (left && mu) uf_union(parent, p - 1, p - W);
    }
    return;
  }
  if (me && left && !(mu && lu)) uf_union(parent, p, p - 1);
  if (CONN == 8) {
    if (me && lu && !left && !mu) uf_union(parent, p, p - W - 1);
    if (left && mu && !me && !lu) uf_union(parent, p - 1, p - W);
  }
}


__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int* total);

// Path flattening fused with pass 1 of the ranking: a block owns one rank chunk, replaces every parent by
// its root and counts the roots of the chunk (a pixel is a root iff it is its own parent).
// ... and with the root bitmap of the chunk: bit (p % 64) of word (p / 64) of the image = pixel p is a root.  The ranking
// kernel numbers the roots from these words (8 B per 64 pixels) instead of reading the parent plane again (256 B).
__device__ __forceinline__ void ccl_flatten_count_body(int vb, int* __restrict__ parent_all, int hw, int nchunks,
                                                                int* __restrict__ chunk_cnt, unsigned long long* __restrict__ rootmask) {
  __shared__ int sh[4];
  const int b = vb / nchunks, ch = vb % nchunks;
  int* parent = parent_all + (size_t)b * hw;
  const int p0 = ch * RK_CHUNK + threadIdx.x;
  int local = 0;
  // Four pixels per thread at a time, hop by hop: the parents of all four, then the parents' parents of all four -- after
  // the tile-local labelling nearly every pixel is a root or points at one, i.e. done after these two loads -- and only
  // what is left walks its chain alone.  One pixel at a time was a chain of 2-3 dependent loads, 16 times in a row.
  constexpr int FU = 4;
  static_assert(RK_PER_T % FU == 0, "whole batches");
  for (int j0 = 0; j0 < RK_PER_T; j0 += FU) {
    int v[FU], g[FU];
#pragma unroll
    for (int j = 0; j < FU; ++j) {
      const int p = p0 + 256 * (j0 + j);        // coalesced: consecutive threads, consecutive pixels
      v[j] = p < hw ? uf_load(parent, p) : -1;
    }
#pragma unroll
    for (int j = 0; j < FU; ++j) g[j] = v[j] >= 0 ? uf_load(parent, v[j]) : -1;
#pragma unroll
    for (int j = 0; j < FU; ++j) {
      const int p = p0 + 256 * (j0 + j);
      bool root = false;
      if (v[j] >= 0) {
        const int r = g[j] == v[j] ? v[j] : uf_find(parent, g[j]);
        if (r != v[j]) parent[p] = r;           // most pixels already point at their root (tile-local labelling)
        root = r == p;
        local += root;
      }
      const unsigned long long m = __ballot(root);                 // this wave's 64 consecutive pixels
      if ((threadIdx.x & 63) == 0) rootmask[((size_t)b * nchunks + ch) * 64 + 4 * (j0 + j) + (threadIdx.x >> 6)] = m;
    }
  }
  int total;
  block_exclusive_scan(local, sh, &total);
  if (threadIdx.x == 0) chunk_cnt[vb] = total;
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl_flatten_count_kernel(int* __restrict__ parent_all, int hw, int nchunks,
                                                                int* __restrict__ chunk_cnt, unsigned long long* __restrict__ rootmask,
                                                                int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl_flatten_count_body(vb, parent_all, hw, nchunks, chunk_cnt, rootmask);
    __syncthreads();
  }
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int* total) {
  // 256 threads
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += sh[i];
  *total = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return base + incl - v;
}

// pass 3: assign raster-order ids to the roots (pass 1, the per-chunk count, is fused into the flattening).
__device__ __forceinline__ void ccl_rank_body(int vb, const unsigned long long* __restrict__ rootmask, int hw, int nchunks,
                                                       const int* __restrict__ chunk_cnt, int* __restrict__ ids_all,
                                                       int* __restrict__ first, int max_labels) {
  // A wave owns 1024 consecutive pixels of the chunk, 64 at a time (coalesced); the root masks of the 16
  // groups stay in scalar registers between the counting and the numbering sweep.
  static_assert(RK_CHUNK == 4 * 16 * 64, "4 waves x 16 groups x 64 lanes");
  __shared__ int sh[4];
  const int b = vb / nchunks, ch = vb % nchunks;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int base = ch * RK_CHUNK + w * 1024 + lane;
  unsigned long long m[16];
  int cnt = 0;
  // the root bitmap the flattening left: this wave's 16 words (pixels beyond the image are no roots there either)
  const unsigned long long* words = rootmask + ((size_t)b * nchunks + ch) * 64 + w * 16;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    m[j] = words[j];
    cnt += __popcll(m[j]);
  }
  if (lane == 0) sh[w] = cnt;
  __syncthreads();
  int id0 = chunk_cnt[vb];   // exclusive offset of the chunk (after ccl_scan_chunks_kernel)
  for (int i = 0; i < w; ++i) id0 += sh[i];
  int* ids = ids_all + (size_t)b * hw;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    if ((m[j] >> lane) & 1ull) {
      const int p = base + 64 * j;
      const int id = id0 + __popcll(m[j] & below) + 1;
      ids[p] = id;
      if (first && id <= max_labels) first[(size_t)b * max_labels + id - 1] = p;   // root = first pixel in raster order
    }
    id0 += __popcll(m[j]);
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl_rank_kernel(const unsigned long long* __restrict__ rootmask, int hw, int nchunks,
                                                       const int* __restrict__ chunk_cnt, int* __restrict__ ids_all,
                                                       int* __restrict__ first, int max_labels, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl_rank_body(vb, rootmask, hw, nchunks, chunk_cnt, ids_all, first, max_labels);
    __syncthreads();
  }
}

// pass 2: per image exclusive scan of the chunk counts (one block per image)
__global__ __launch_bounds__(256) void ccl_scan_chunks_kernel(int* __restrict__ chunk_cnt, int nchunks,
                                                              int* __restrict__ n_out) {
  __shared__ int sh[4];
  int* c = chunk_cnt + (size_t)blockIdx.x * nchunks;
  int carry = 0;
  for (int base = 0; base < nchunks; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nchunks ? c[i] : 0;
    int total;
    const int excl = block_exclusive_scan(v, sh, &total);
    if (i < nchunks) c[i] = carry + excl;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) n_out[blockIdx.x] = carry;
}

// rows of labels that exist (1..n, capped at max_labels) only: n is on the device by now
__global__ void ccl_stats_init_kernel(int* __restrict__ stats, const int* __restrict__ n_out, int max_labels, int H, int W) {
  const int b = blockIdx.y;
  const int n = min(n_out[b], max_labels);
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) {
    int* s = stats + ((size_t)b * max_labels + l) * 5;
    s[0] = W; s[1] = H; s[2] = -1; s[3] = -1; s[4] = 0;
  }
}

// Final labels + statistics.  A block owns LB_CHUNK consecutive pixels of one image.  Statistics go
// through two levels of aggregation before they reach HBM: (1) a wave covers 64 consecutive pixels, each
// horizontal run of one label is handled by its first lane; (2) the runs of a block are merged per label in
// a small LDS hash table that is flushed once at the end.  Big components (a page's background, a window's
// complement) otherwise serialise tens of thousands of atomics on the same five words of `stats`
// (rocprofv3: 1.2-2.7 ms per launch before the block-level table).
constexpr int LB_CHUNK = 8192;
constexpr int LB_SLOTS = 128;

__device__ __forceinline__ void ccl_label_body(int vb, int* __restrict__ labels_all, const int* __restrict__ ids_all,
                                                        int B, int H, int W, int chunks, int* __restrict__ stats,
                                                        int max_labels, int bg_negative) {
  __shared__ int hkey[LB_SLOTS];
  __shared__ int hst[LB_SLOTS * 5];
  const int hw = H * W;
  const int b = vb / chunks, ch = vb % chunks;
  const int p_begin = ch * LB_CHUNK, p_end = min(hw, p_begin + LB_CHUNK);
  const size_t base = (size_t)b * hw;
  if (stats) {
    for (int s = threadIdx.x; s < LB_SLOTS; s += 256) {
      hkey[s] = -1;
      hst[5 * s] = W, hst[5 * s + 1] = H, hst[5 * s + 2] = -1, hst[5 * s + 3] = -1, hst[5 * s + 4] = 0;
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  // Four 256-pixel steps at a time: their roots, then their ids (a dependent gather), then the statistics step by step.
  // One step at a time was two memory round trips per step, 32 steps in a row.
  constexpr int LU = 4;
  for (int pb = p_begin; pb < p_end; pb += 256 * LU) {
    int root[LU], idv[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int p = pb + 256 * u + threadIdx.x;
      root[u] = p < p_end ? labels_all[base + p] : -1;
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) idv[u] = root[u] >= 0 ? ids_all[base + root[u]] : 0;
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int p = pb + 256 * u + threadIdx.x;
      if (p < p_end && !(bg_negative && root[u] < 0)) labels_all[base + p] = idv[u];   // (a background pixel holds -1 already)
    }
    if (!stats) continue;
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int p0 = pb + 256 * u;
      if (p0 >= p_end) break;
      const int p = p0 + threadIdx.x;
      const bool live = p < p_end;
      const int id = idv[u];
      const int x = live ? p % W : 0, y = live ? p / W : 0;
      const int key = live ? id : -1;                         // dead lanes end a run
      const int prev = __shfl_up(key, 1);
      const bool head = lane == 0 || prev != key || x == 0;
      const unsigned long long heads = __ballot(head);
      if (head && id > 0 && id <= max_labels) {
        const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int len = later ? __ffsll((long long)later) : 64 - lane;
        int slot = -1;
        unsigned hsh = ((unsigned)id * 2654435761u) >> 25;    // 7 bits
        for (int probe = 0; probe < 4; ++probe) {
          const int sidx = (hsh + probe) & (LB_SLOTS - 1);
          const int old = atomicCAS(&hkey[sidx], -1, id);
          if (old == -1 || old == id) {
            slot = sidx;
            break;
          }
        }
        int* s = slot >= 0 ? hst + 5 * slot : stats + ((size_t)b * max_labels + (id - 1)) * 5;
        atomicMin(s + 0, x);
        atomicMin(s + 1, y);
        atomicMax(s + 2, x + len - 1);
        atomicMax(s + 3, y);
        atomicAdd(s + 4, len);
      }
    }
  }
  if (stats) {
    __syncthreads();
    for (int sl = threadIdx.x; sl < LB_SLOTS; sl += 256) {
      const int id = hkey[sl];
      if (id <= 0) continue;
      int* s = stats + ((size_t)b * max_labels + (id - 1)) * 5;
      atomicMin(s + 0, hst[5 * sl]);
      atomicMin(s + 1, hst[5 * sl + 1]);
      atomicMax(s + 2, hst[5 * sl + 2]);
      atomicMax(s + 3, hst[5 * sl + 3]);
      atomicAdd(s + 4, hst[5 * sl + 4]);
    }
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl_label_kernel(int* __restrict__ labels_all, const int* __restrict__ ids_all,
                                                        int B, int H, int W, int chunks, int* __restrict__ stats,
                                                        int max_labels, int bg_negative, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl_label_body(vb, labels_all, ids_all, B, H, W, chunks, stats, max_labels, bg_negative);
    __syncthreads();
  }
}

__global__ void ccl_stats_final_kernel(int* __restrict__ stats, const int* __restrict__ n_out, int max_labels) {
  const int b = blockIdx.y;
  const int n = min(n_out[b], max_labels);
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < n; l += gridDim.x * blockDim.x) {
    int* s = stats + ((size_t)b * max_labels + l) * 5;
    s[2] = s[2] - s[0] + 1;
    s[3] = s[3] - s[1] + 1;
  }
}

// ======================================================================================================
// Dual labelling: the 8-connected components of the FOREGROUND (img > thresh) and the 4-connected components
// of the BACKGROUND of one image in ONE union-find.  The DB stage needs both (cv2.findContours(RETR_LIST)
// yields a contour per foreground component and per enclosed background region); the two pixel sets are
// disjoint, so one parent array, one border merge, one flattening, one ranking sweep and one label pass serve
// both -- the two separate launches read and wrote every int32 plane twice.  Output: ONE signed label image
// (+id foreground, -id background, ids per class in raster order of the first pixel) and per-class stats.
// ======================================================================================================
__device__ __forceinline__ void ccl2_local_body(int vb, const uint8_t* __restrict__ img, int* __restrict__ parent_all,
                                                         int H, int W, int tiles_x, int tiles_y, int thresh) {
  __shared__ int lp[CT * CT];
  __shared__ unsigned mrow[2][CT];                 // row masks: [0] foreground, [1] background (pixels inside the image)
  int bid = vb;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y;
  const int b = bid / tiles_y;
  const size_t base = (size_t)b * H * W;
  const int x0 = tx * CT, y0 = ty * CT;
  const int lane = threadIdx.x & 63;
  const int lx = threadIdx.x & 31;
  int pix[CT * CT / 256];                                   // loads first, ballots second (see ccl_local_body)
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int gx = x0 + lx, gy = y0 + ((threadIdx.x + 256 * k) >> 5);
    pix[k] = img[base + (gx < W && gy < H ? (size_t)gy * W + gx : (size_t)0)];
  }
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int li = threadIdx.x + 256 * k;
    const int ly = li >> 5;
    const int gx = x0 + lx, gy = y0 + ly;
    const bool valid = gx < W && gy < H;
    const bool fg = valid && pix[k] > thresh;
    const unsigned long long balf = __ballot(fg), balv = __ballot(valid);
    const unsigned mf = (unsigned)(lane < 32 ? balf : balf >> 32);
    const unsigned mv = (unsigned)(lane < 32 ? balv : balv >> 32);
    const unsigned mb = mv & ~mf;
    const unsigned mine = fg ? mf : mb;
    const unsigned below = ~mine & ((1u << lx) - 1u);
    const int start = below ? 32 - __clz(below) : 0;
    lp[li] = valid ? (ly << 5) + start : -1;
    if (lx == 0) mrow[0][ly] = mf, mrow[1][ly] = mb;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int li = threadIdx.x + 256 * k;
    const int ly = li >> 5;
    if (ly == 0) continue;
    const int cls = ((mrow[0][ly] >> lx) & 1u) ? 0 : 1;
    const unsigned m = mrow[cls][ly], ma = mrow[cls][ly - 1];
    if (!((m >> lx) & 1u) || (lx > 0 && ((m >> (lx - 1)) & 1u))) continue;   // heads of this class's runs only
    const unsigned from = m >> lx;
    const int len = (~from) ? __ffs(~from) - 1 : 32 - lx;
    unsigned run = (len >= 32 ? 0xffffffffu : ((1u << len) - 1u)) << lx;
    if (cls == 0) run |= (run << 1) | (run >> 1);                            // foreground: 8-connected
    unsigned ov = ma & run;
    while (ov) {
      const int bpos = __ffs(ov) - 1;
      const unsigned belowa = ~ma & ((1u << bpos) - 1u);
      const int sa = belowa ? 32 - __clz(belowa) : 0;
      lds_union(lp, li, ((ly - 1) << 5) + sa);
      const unsigned froma = ma >> bpos;
      const int lena = (~froma) ? __ffs(~froma) - 1 : 32 - bpos;
      ov &= ~((lena >= 32 ? 0xffffffffu : ((1u << lena) - 1u)) << bpos);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < CT * CT / 256; ++k) {
    const int li = threadIdx.x + 256 * k;
    const int gx = x0 + lx, gy = y0 + (li >> 5);
    if (gx >= W || gy >= H) continue;
    const int r = lds_find(lp, lp[li]);
    parent_all[base + (size_t)gy * W + gx] = (y0 + (r >> 5)) * W + x0 + (r & 31);
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl2_local_kernel(const uint8_t* __restrict__ img, int* __restrict__ parent_all,
                                                         int H, int W, int tiles_x, int tiles_y, int thresh, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl2_local_body(vb, img, parent_all, H, W, tiles_x, tiles_y, thresh);
    __syncthreads();
  }
}

// Border links of both classes (pruned to one link per pair of overlapping runs, see ccl_border_*_kernel):
// foreground pixels follow the 8-connected rules, background pixels the 4-connected ones.
__global__ __launch_bounds__(256) void ccl2_border_h_kernel(int* __restrict__ parent_all, const uint8_t* __restrict__ img_all,
                                                            int thresh, int H, int W) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  const int y = CT * (blockIdx.y + 1);
  if (x >= W || y >= H) return;
  int* parent = parent_all + (size_t)blockIdx.z * H * W;
  const uint8_t* img = img_all + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  const bool me = (int)img[p] > thresh;
  auto same = [&](int q) { return ((int)img[q] > thresh) == me; };
  const bool q = x > 0 && same(p - 1);
  const bool up = same(p - W);
  const bool ul = x > 0 && same(p - W - 1);
  if (!me) {                                      // background: 4-connected
    if (up && !(q && ul)) uf_union(parent, p, p - W);
    return;
  }
  const bool ur = x + 1 < W && same(p - W + 1);
  if (q) {
    if (ur && !up) uf_union(parent, p, p - W + 1);
  } else if (up) {
    uf_union(parent, p, p - W);
  } else {
    if (ul) uf_union(parent, p, p - W - 1);
    if (ur) uf_union(parent, p, p - W + 1);
  }
}

__global__ __launch_bounds__(256) void ccl2_border_v_kernel(int* __restrict__ parent_all, const uint8_t* __restrict__ img_all,
                                                            int thresh, int H, int W) {
  const int y = blockIdx.x * 256 + threadIdx.x;
  const int x = CT * (blockIdx.y + 1);
  if (y >= H || x >= W) return;
  int* parent = parent_all + (size_t)blockIdx.z * H * W;
  const uint8_t* img = img_all + (size_t)blockIdx.z * H * W;
  const int p = y * W + x;
  const bool ua = y > 0, same_tile = (y % CT) != 0;
  const bool fme = (int)img[p] > thresh, fleft = (int)img[p - 1] > thresh;
  const bool fmu = ua && (int)img[p - W] > thresh, flu = ua && (int)img[p - W - 1] > thresh;
  // background, 4-connected: only the horizontal pair
  {
    const bool me = !fme, left = !fleft, mu = ua && !fmu, lu = ua && !flu;
    if (me && left && !(same_tile && mu && lu)) uf_union(parent, p, p - 1);
  }
  // foreground, 8-connected
  {
    const bool me = fme, left = fleft, mu = fmu, lu = flu;
    if (!me && !left) return;
    if (!same_tile) {
      if (me && left) uf_union(parent, p, p - 1);
      if (me && lu) uf_union(parent, p, p - W - 1);
      if (left && mu) uf_union(parent, p - 1, p - W);
      return;
    }
    if (me && left && !(mu && lu)) uf_union(parent, p, p - 1);
    if (me && lu && !left && !mu) uf_union(parent, p, p - W - 1);
    if (left && mu && !me && !lu) uf_union(parent, p - 1, p - W);
  }
}

// chunk_cnt: (B, 2, nchunks) -- class 0 = foreground roots, 1 = background roots
__device__ __forceinline__ void ccl2_flatten_count_body(int vb, int* __restrict__ parent_all, const uint8_t* __restrict__ img_all,
                                                                 int thresh, int hw, int nchunks, int* __restrict__ chunk_cnt,
                                                                 unsigned long long* __restrict__ rootmask) {
  __shared__ int sh[4];
  const int b = vb / nchunks, ch = vb % nchunks;
  int* parent = parent_all + (size_t)b * hw;
  const uint8_t* img = img_all + (size_t)b * hw;
  const int p0 = ch * RK_CHUNK + threadIdx.x;
  int lf = 0, lb = 0;
  constexpr int FU = 4;                         // hop by hop over four pixels (see ccl_flatten_count_body)
  static_assert(RK_PER_T % FU == 0, "whole batches");
  for (int j0 = 0; j0 < RK_PER_T; j0 += FU) {
    int v[FU], g[FU], px[FU];
#pragma unroll
    for (int j = 0; j < FU; ++j) {
      const int p = p0 + 256 * (j0 + j);
      v[j] = p < hw ? uf_load(parent, p) : -1;  // every pixel of the image has a class: parents are never negative
      px[j] = img[min(p, hw - 1)];
    }
#pragma unroll
    for (int j = 0; j < FU; ++j) g[j] = v[j] >= 0 ? uf_load(parent, v[j]) : -1;
#pragma unroll
    for (int j = 0; j < FU; ++j) {
      const int p = p0 + 256 * (j0 + j);
      bool rf = false, rb = false;
      if (v[j] >= 0) {
        const int r = g[j] == v[j] ? v[j] : uf_find(parent, g[j]);
        if (r != v[j]) parent[p] = r;
        if (r == p) {
          if (px[j] > thresh) ++lf, rf = true; else ++lb, rb = true;
        }
      }
      // root bitmaps per class (see ccl_flatten_count_body): (B, 2, nchunks, 64) words
      const unsigned long long mf = __ballot(rf), mb = __ballot(rb);
      if ((threadIdx.x & 63) == 0) {
        const size_t wi = (size_t)ch * 64 + 4 * (j0 + j) + (threadIdx.x >> 6);
        rootmask[((size_t)b * 2 + 0) * nchunks * 64 + wi] = mf;
        rootmask[((size_t)b * 2 + 1) * nchunks * 64 + wi] = mb;
      }
    }
  }
  int tf, tb;
  block_exclusive_scan(lf, sh, &tf);
  block_exclusive_scan(lb, sh, &tb);
  if (threadIdx.x == 0) {
    chunk_cnt[((size_t)b * 2 + 0) * nchunks + ch] = tf;
    chunk_cnt[((size_t)b * 2 + 1) * nchunks + ch] = tb;
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl2_flatten_count_kernel(int* __restrict__ parent_all, const uint8_t* __restrict__ img_all,
                                                                 int thresh, int hw, int nchunks, int* __restrict__ chunk_cnt,
                                                                 unsigned long long* __restrict__ rootmask, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl2_flatten_count_body(vb, parent_all, img_all, thresh, hw, nchunks, chunk_cnt, rootmask);
    __syncthreads();
  }
}

// one block per (image, class): exclusive scan of that class's chunk counts; the totals go to n_f / n_b
__global__ __launch_bounds__(256) void ccl2_scan_chunks_kernel(int* __restrict__ chunk_cnt, int nchunks, int* __restrict__ n_f,
                                                               int* __restrict__ n_b) {
  __shared__ int sh[4];
  int* c = chunk_cnt + (size_t)blockIdx.x * nchunks;
  int carry = 0;
  for (int base = 0; base < nchunks; base += 256) {
    const int i = base + threadIdx.x;
    const int v = i < nchunks ? c[i] : 0;
    int total;
    const int excl = block_exclusive_scan(v, sh, &total);
    if (i < nchunks) c[i] = carry + excl;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) ((blockIdx.x & 1) ? n_b : n_f)[blockIdx.x >> 1] = carry;
}

// ids: +rank for a foreground root, -rank for a background root (ranks per class, raster order)
__device__ __forceinline__ void ccl2_rank_body(int vb, const unsigned long long* __restrict__ rootmask,
                                                        int hw, int nchunks, const int* __restrict__ chunk_cnt,
                                                        int* __restrict__ ids_all, int* __restrict__ first_f,
                                                        int* __restrict__ first_b, int max_labels) {
  __shared__ int sh[2][4];
  const int b = vb / nchunks, ch = vb % nchunks;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int base = ch * RK_CHUNK + w * 1024 + lane;
  unsigned long long mf[16], mb[16];
  int cf = 0, cb = 0;
  const unsigned long long* wf = rootmask + ((size_t)b * 2 + 0) * nchunks * 64 + (size_t)ch * 64 + w * 16;   // the flattening's root bitmaps
  const unsigned long long* wb = rootmask + ((size_t)b * 2 + 1) * nchunks * 64 + (size_t)ch * 64 + w * 16;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    mf[j] = wf[j];
    mb[j] = wb[j];
    cf += __popcll(mf[j]);
    cb += __popcll(mb[j]);
  }
  if (lane == 0) sh[0][w] = cf, sh[1][w] = cb;
  __syncthreads();
  int idf = chunk_cnt[((size_t)b * 2 + 0) * nchunks + ch], idb = chunk_cnt[((size_t)b * 2 + 1) * nchunks + ch];
  for (int i = 0; i < w; ++i) idf += sh[0][i], idb += sh[1][i];
  int* ids = ids_all + (size_t)b * hw;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int p = base + 64 * j;
    if ((mf[j] >> lane) & 1ull) {
      const int id = idf + __popcll(mf[j] & below) + 1;
      ids[p] = id;
      if (id <= max_labels) first_f[(size_t)b * max_labels + id - 1] = p;
    } else if ((mb[j] >> lane) & 1ull) {
      const int id = idb + __popcll(mb[j] & below) + 1;
      ids[p] = -id;
      if (id <= max_labels) first_b[(size_t)b * max_labels + id - 1] = p;
    }
    idf += __popcll(mf[j]);
    idb += __popcll(mb[j]);
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl2_rank_kernel(const unsigned long long* __restrict__ rootmask,
                                                        int hw, int nchunks, const int* __restrict__ chunk_cnt,
                                                        int* __restrict__ ids_all, int* __restrict__ first_f,
                                                        int* __restrict__ first_b, int max_labels, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl2_rank_body(vb, rootmask, hw, nchunks, chunk_cnt, ids_all, first_f, first_b, max_labels);
    __syncthreads();
  }
}

// signed final labels + the statistics of both classes (aggregation as in ccl_label_kernel)
__device__ __forceinline__ void ccl2_label_body(int vb, int* __restrict__ labels_all, const int* __restrict__ ids_all, int B, int H,
                                                         int W, int chunks, int* __restrict__ st_f, int* __restrict__ st_b,
                                                         int max_labels) {
  constexpr int EMPTY = (int)0x80000000;
  __shared__ int hkey[LB_SLOTS];
  __shared__ int hst[LB_SLOTS * 5];
  const int hw = H * W;
  const int b = vb / chunks, ch = vb % chunks;
  const int p_begin = ch * LB_CHUNK, p_end = min(hw, p_begin + LB_CHUNK);
  const size_t base = (size_t)b * hw;
  for (int s = threadIdx.x; s < LB_SLOTS; s += 256) {
    hkey[s] = EMPTY;
    hst[5 * s] = W, hst[5 * s + 1] = H, hst[5 * s + 2] = -1, hst[5 * s + 3] = -1, hst[5 * s + 4] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  auto row_of = [&](int id) { return (id > 0 ? st_f : st_b) + ((size_t)b * max_labels + ((id > 0 ? id : -id) - 1)) * 5; };
  constexpr int LU = 4;                                     // four steps' roots, then their ids (see ccl_label_body)
  for (int pb = p_begin; pb < p_end; pb += 256 * LU) {
    int root[LU], idv[LU];
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int p = pb + 256 * u + threadIdx.x;
      root[u] = p < p_end ? labels_all[base + p] : -1;
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) idv[u] = root[u] >= 0 ? ids_all[base + root[u]] : 0;
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int p = pb + 256 * u + threadIdx.x;
      if (p < p_end) labels_all[base + p] = idv[u];
    }
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      const int p0 = pb + 256 * u;
      if (p0 >= p_end) break;
      const int p = p0 + threadIdx.x;
      const bool live = p < p_end;
      const int id = idv[u];                                  // 0 = dead lane (every pixel has a class, so no real id is 0)
      const int x = live ? p % W : 0, y = live ? p / W : 0;
      const int prev = __shfl_up(id, 1);
      const bool head = lane == 0 || prev != id || x == 0;
      const unsigned long long heads = __ballot(head);
      const int mag = id > 0 ? id : -id;
      if (head && id != 0 && mag <= max_labels) {
        const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
        const int len = later ? __ffsll((long long)later) : 64 - lane;
        int slot = -1;
        const unsigned hsh = ((unsigned)id * 2654435761u) >> 25;    // 7 bits
        for (int probe = 0; probe < 4; ++probe) {
          const int sidx = (hsh + probe) & (LB_SLOTS - 1);
          const int old = atomicCAS(&hkey[sidx], EMPTY, id);
          if (old == EMPTY || old == id) {
            slot = sidx;
            break;
          }
        }
        int* s = slot >= 0 ? hst + 5 * slot : row_of(id);
        atomicMin(s + 0, x);
        atomicMin(s + 1, y);
        atomicMax(s + 2, x + len - 1);
        atomicMax(s + 3, y);
        atomicAdd(s + 4, len);
      }
    }
  }
  __syncthreads();
  for (int sl = threadIdx.x; sl < LB_SLOTS; sl += 256) {
    const int id = hkey[sl];
    if (id == EMPTY) continue;
    int* s = row_of(id);
    atomicMin(s + 0, hst[5 * sl]);
    atomicMin(s + 1, hst[5 * sl + 1]);
    atomicMax(s + 2, hst[5 * sl + 2]);
    atomicMax(s + 3, hst[5 * sl + 3]);
    atomicAdd(s + 4, hst[5 * sl + 4]);
  }
}
// a block walks several tiles / chunks: the launcher caps the grid (tail_max_blocks) so that these kernels hold a few
// waves per SIMD next to the network, not all of them
__global__ __launch_bounds__(256) void ccl2_label_kernel(int* __restrict__ labels_all, const int* __restrict__ ids_all, int B, int H,
                                                         int W, int chunks, int* __restrict__ st_f, int* __restrict__ st_b,
                                                         int max_labels, int nvb) {
  for (int vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
    ccl2_label_body(vb, labels_all, ids_all, B, H, W, chunks, st_f, st_b, max_labels);
    __syncthreads();
  }
}


// grid of a kernel whose blocks walk `nvb` tiles / chunks: at most g_tail_max_blocks blocks
inline int capped(int nvb) { return std::max(1, std::min(nvb, g_tail_max_blocks)); }

inline int grid_for(long long total, int block = 256) {
  long long g = (total + block - 1) / block;
  if (g > (long long)g_tail_max_blocks) g = g_tail_max_blocks;
  if (g < 1) g = 1;
  return (int)g;
}

template <int CONN>
void launch_border(int* labels, const unsigned* colmask, int B, int H, int W, hipStream_t st) {
  const int nh = (H - 1) / CT, nv = (W - 1) / CT;      // boundaries strictly inside the image
  if (nh > 0) hipLaunchKernelGGL((ccl_border_h_kernel<CONN>), dim3((W + 255) / 256, nh, B), dim3(256), 0, st, labels, H, W);
  if (nv > 0)
    hipLaunchKernelGGL((ccl_border_v_kernel<CONN>), dim3((H + 255) / 256, nv, B), dim3(256), 0, st, labels, colmask, H, W,
                       (W + CT - 1) / CT, (H + CT - 1) / CT);
}

}  // namespace

size_t nms_workspace_bytes(int B, int rows) { return (size_t)B * rows * sizeof(Cand) + (size_t)B * sizeof(int) + 256; }

void launch_nms(const float* blks, int B, int rows, int no, float conf, float iou, int max_det, int max_nms,
                float max_wh, float* dets, int* counts, void* ws, hipStream_t st) {
  int* cnt = (int*)ws;
  Cand* cands = (Cand*)((char*)ws + ((size_t)B * sizeof(int) + 255) / 256 * 256);
  (void)hipMemsetAsync(cnt, 0, (size_t)B * sizeof(int), st);
  (void)hipMemsetAsync(dets, 0, (size_t)B * max_det * 6 * sizeof(float), st);
  hipLaunchKernelGGL(nms_filter_kernel, dim3(grid_for((long long)B * rows)), dim3(256), 0, st, blks, B, rows, no, conf,
                     cands, cnt);
  hipLaunchKernelGGL(nms_greedy_kernel, dim3(B), dim3(NMS_THREADS), 0, st, cands, cnt, rows, iou, max_det, max_nms, max_wh,
                     dets, counts);
}

size_t ccl_workspace_bytes(int B, int H, int W) {
  const size_t hw = (size_t)H * W;
  const size_t nchunks = (hw + RK_CHUNK - 1) / RK_CHUNK;
  // ids + chunk counts + root bitmaps (64 words per chunk; two planes of each for launch_ccl_dual) + the tiles' column masks
  // (two words per 32 x 32 tile; callers size for a pixel count and label any H x W within it: tiles <= hw / 1024 + (H + W) / 32 + 1)
  const size_t tiles_max = hw / (CT * CT) + (hw + 1) / CT + 2;
  return (size_t)B * hw * sizeof(int) + 2 * (size_t)B * nchunks * sizeof(int) + 2 * (size_t)B * nchunks * 64 * 8 +
         (size_t)B * tiles_max * 2 * sizeof(unsigned) + 1280;
}

void launch_ccl(const uint8_t* img, int B, int H, int W, int thresh, int conn, int* labels, int* n_out, int* stats,
                int max_labels, void* ws, hipStream_t st, int invert, int* first, int bg_negative) {
  const int hw = H * W;
  const long long total = (long long)B * hw;
  const int nchunks = (hw + RK_CHUNK - 1) / RK_CHUNK;
  int* ids = (int*)ws;
  int* chunk_cnt = (int*)((char*)ws + ((size_t)total * sizeof(int) + 255) / 256 * 256);
  unsigned long long* rootmask = (unsigned long long*)((char*)chunk_cnt + ((size_t)B * nchunks * sizeof(int) + 255) / 256 * 256);
  unsigned* colmask = (unsigned*)((char*)rootmask + ((size_t)B * nchunks * 64 * 8 + 255) / 256 * 256);
  const int tiles_x = (W + CT - 1) / CT, tiles_y = (H + CT - 1) / CT;
  if (conn == 8) {
    hipLaunchKernelGGL((ccl_local_kernel<8>), dim3(capped(B * tiles_x * tiles_y)), dim3(256), 0, st, img, labels, colmask, H, W,
                       tiles_x, tiles_y, thresh, invert, B * tiles_x * tiles_y);
    launch_border<8>(labels, colmask, B, H, W, st);
  } else {
    hipLaunchKernelGGL((ccl_local_kernel<4>), dim3(capped(B * tiles_x * tiles_y)), dim3(256), 0, st, img, labels, colmask, H, W,
                       tiles_x, tiles_y, thresh, invert, B * tiles_x * tiles_y);
    launch_border<4>(labels, colmask, B, H, W, st);
  }
  hipLaunchKernelGGL(ccl_flatten_count_kernel, dim3(capped(B * nchunks)), dim3(256), 0, st, labels, hw, nchunks, chunk_cnt, rootmask,
                     B * nchunks);
  hipLaunchKernelGGL(ccl_scan_chunks_kernel, dim3(B), dim3(256), 0, st, chunk_cnt, nchunks, n_out);
  hipLaunchKernelGGL(ccl_rank_kernel, dim3(capped(B * nchunks)), dim3(256), 0, st, rootmask, hw, nchunks, chunk_cnt, ids, first,
                     max_labels, B * nchunks);
  const int sgrid = std::max(1, std::min(64, (max_labels + 255) / 256));
  if (stats) hipLaunchKernelGGL(ccl_stats_init_kernel, dim3(sgrid, B), dim3(256), 0, st, stats, n_out, max_labels, H, W);
  const int lchunks = (hw + LB_CHUNK - 1) / LB_CHUNK;
  hipLaunchKernelGGL(ccl_label_kernel, dim3(capped(B * lchunks)), dim3(256), 0, st, labels, ids, B, H, W, lchunks, stats, max_labels,
                     bg_negative, B * lchunks);
  if (stats) hipLaunchKernelGGL(ccl_stats_final_kernel, dim3(sgrid, B), dim3(256), 0, st, stats, n_out, max_labels);
}

void launch_ccl_dual(const uint8_t* img, int B, int H, int W, int thresh, int* labels, int* n_f, int* n_b, int* st_f,
                     int* st_b, int* first_f, int* first_b, int max_labels, void* ws, hipStream_t st) {
  const int hw = H * W;
  const long long total = (long long)B * hw;
  const int nchunks = (hw + RK_CHUNK - 1) / RK_CHUNK;
  int* ids = (int*)ws;
  int* chunk_cnt = (int*)((char*)ws + ((size_t)total * sizeof(int) + 255) / 256 * 256);   // (B, 2, nchunks)
  unsigned long long* rootmask = (unsigned long long*)((char*)chunk_cnt + (2 * (size_t)B * nchunks * sizeof(int) + 255) / 256 * 256);
  const int tiles_x = (W + CT - 1) / CT, tiles_y = (H + CT - 1) / CT;
  hipLaunchKernelGGL(ccl2_local_kernel, dim3(capped(B * tiles_x * tiles_y)), dim3(256), 0, st, img, labels, H, W, tiles_x, tiles_y,
                     thresh, B * tiles_x * tiles_y);
  const int nh = (H - 1) / CT, nv = (W - 1) / CT;
  if (nh > 0) hipLaunchKernelGGL(ccl2_border_h_kernel, dim3((W + 255) / 256, nh, B), dim3(256), 0, st, labels, img, thresh, H, W);
  if (nv > 0) hipLaunchKernelGGL(ccl2_border_v_kernel, dim3((H + 255) / 256, nv, B), dim3(256), 0, st, labels, img, thresh, H, W);
  hipLaunchKernelGGL(ccl2_flatten_count_kernel, dim3(capped(B * nchunks)), dim3(256), 0, st, labels, img, thresh, hw, nchunks, chunk_cnt,
                     rootmask, B * nchunks);
  hipLaunchKernelGGL(ccl2_scan_chunks_kernel, dim3(2 * B), dim3(256), 0, st, chunk_cnt, nchunks, n_f, n_b);
  hipLaunchKernelGGL(ccl2_rank_kernel, dim3(capped(B * nchunks)), dim3(256), 0, st, rootmask, hw, nchunks, chunk_cnt, ids,
                     first_f, first_b, max_labels, B * nchunks);
  const int sgrid = std::max(1, std::min(64, (max_labels + 255) / 256));
  hipLaunchKernelGGL(ccl_stats_init_kernel, dim3(sgrid, B), dim3(256), 0, st, st_f, n_f, max_labels, H, W);
  hipLaunchKernelGGL(ccl_stats_init_kernel, dim3(sgrid, B), dim3(256), 0, st, st_b, n_b, max_labels, H, W);
  const int lchunks = (hw + LB_CHUNK - 1) / LB_CHUNK;
  hipLaunchKernelGGL(ccl2_label_kernel, dim3(capped(B * lchunks)), dim3(256), 0, st, labels, ids, B, H, W, lchunks, st_f, st_b,
                     max_labels, B * lchunks);
  hipLaunchKernelGGL(ccl_stats_final_kernel, dim3(sgrid, B), dim3(256), 0, st, st_f, n_f, max_labels);
  hipLaunchKernelGGL(ccl_stats_final_kernel, dim3(sgrid, B), dim3(256), 0, st, st_b, n_b, max_labels);
}
