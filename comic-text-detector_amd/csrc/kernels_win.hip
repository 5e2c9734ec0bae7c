// Batched per-window kernels of the mask refinement (reference utils/textmask.py:29-71,
// SURVEY K14): every text-block window of a page batch is one entry of a window table; a
// launch covers all windows (blockIdx.y = window).  Integer / comparison work on u8 pixels.
//
//   win_hist   : grey = BGR2GRAY (15-bit fixed point), msk eroded 3x3 (rect) inside the window;
//                histograms: grey over pixels with eroded mask > 127 (textmask.py:60-61), and
//                B, G, R over the whole window (Otsu, textmask.py:44-47)
//   win_xor    : for up to 6 candidate rules per window (3 grey ranges, 3 channel thresholds)
//                the L1 / xor distance sum(cand ? 255-m : m) to the raw mask (textmask.py:36-37);
//                the negative's distance is 255*n minus it
//   win_render : the chosen candidates (rule + polarity) written as {0,255} bands of the
//                labelling canvas, so `ctd_ccl` runs on them without a host round trip
#include "kernels.h"

namespace {

__device__ __forceinline__ int gray_of(const uint8_t* p) {
  // OpenCV 4.x RGB2Gray<uchar>: 15-bit coefficients (BY15, GY15, RY15), round to nearest
  return ((int)p[0] * 3735 + (int)p[1] * 19235 + (int)p[2] * 9798 + 16384) >> 15;
}

__device__ __forceinline__ int erode_rect(const CtdWin& w, int x, int y) {
  int m = 255;
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = y + dy;
    if (yy < 0 || yy >= w.h) continue;
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = x + dx;
      if (xx < 0 || xx >= w.w) continue;
      m = min(m, (int)w.mask[(size_t)(w.y1 + yy) * w.mask_w + w.x1 + xx]);
    }
  }
  return m;
}

__global__ __launch_bounds__(256) void win_hist_kernel(const CtdWin* __restrict__ wins, unsigned* __restrict__ hist) {
  __shared__ unsigned h[4 * 256];
  const CtdWin w = wins[blockIdx.y];
  for (int i = threadIdx.x; i < 1024; i += 256) h[i] = 0;
  __syncthreads();
  const int npix = w.w * w.h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    const uint8_t* p = w.img + ((size_t)(w.y1 + y) * w.img_w + w.x1 + x) * 3;
    atomicAdd(&h[256 + p[0]], 1u);
    atomicAdd(&h[512 + p[1]], 1u);
    atomicAdd(&h[768 + p[2]], 1u);
    if (erode_rect(w, x, y) > 127) atomicAdd(&h[gray_of(p)], 1u);
  }
  __syncthreads();
  unsigned* out = hist + (size_t)blockIdx.y * 1024;
  for (int i = threadIdx.x; i < 1024; i += 256)
    if (h[i]) atomicAdd(out + i, h[i]);
}

__device__ __forceinline__ bool rule_on(const CtdRule& r, const uint8_t* p) {
  if (r.kind == 0) {                      // cv2.inRange(grey, lo, hi): lo / hi are the integer bounds
    const float g = (float)gray_of(p);   // cv2 derives from the scalars (cvRound, saturated; lo > hi = empty)
    return g >= r.lo && g <= r.hi;
  }
  return (float)p[r.kind - 1] > r.lo;      // threshold(channel, t, 255, BINARY): kind 1..3 = B,G,R
}

__global__ __launch_bounds__(256) void win_xor_kernel(const CtdWin* __restrict__ wins, const CtdRule* __restrict__ rules,
                                                      int nrules, unsigned long long* __restrict__ sums) {
  __shared__ unsigned long long red[4];
  const CtdWin w = wins[blockIdx.y];
  const CtdRule* rs = rules + (size_t)blockIdx.y * nrules;
  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
  const int npix = w.w * w.h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    const uint8_t* p = w.img + ((size_t)(w.y1 + y) * w.img_w + w.x1 + x) * 3;
    const int m = w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x];
    for (int k = 0; k < nrules && k < 6; ++k)
      if (rs[k].kind >= 0) acc[k] += rule_on(rs[k], p) ? (255 - m) : m;
  }
  for (int k = 0; k < nrules && k < 6; ++k) {
    unsigned long long v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t = red[0] + red[1] + red[2] + red[3];
      if (t) atomicAdd(sums + (size_t)blockIdx.y * nrules + k, t);
    }
  }
}

__global__ __launch_bounds__(256) void win_render_kernel(const CtdWin* __restrict__ wins, const CtdRule* __restrict__ rules,
                                                         const int* __restrict__ tops, int nbands,
                                                         uint8_t* __restrict__ canvas, int canvas_w) {
  // blockIdx.y = band (window, candidate); rules[band].aux = window index, .invert = polarity
  const CtdRule r = rules[blockIdx.y];
  const CtdWin w = wins[r.aux];
  const int top = tops[blockIdx.y];
  const int npix = w.w * w.h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    const uint8_t* p = w.img + ((size_t)(w.y1 + y) * w.img_w + w.x1 + x) * 3;
    const bool on = rule_on(r, p) != (r.invert != 0);
    canvas[(size_t)(top + y) * canvas_w + x] = on ? 255 : 0;
  }
  (void)nbands;
}

// ---- merge stage (reference utils/textmask.py:74-131) on the labelled canvas ----------------
// pred_bin of merge_mask_list (:85-89): 3x3 cross erosion of the window's mask, > 60 -> 255
__device__ __forceinline__ bool pred_on(const CtdWin& w, int x, int y) {
  int m = w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x];
  if (y > 0) m = min(m, (int)w.mask[(size_t)(w.y1 + y - 1) * w.mask_w + w.x1 + x]);
  if (y + 1 < w.h) m = min(m, (int)w.mask[(size_t)(w.y1 + y + 1) * w.mask_w + w.x1 + x]);
  if (x > 0) m = min(m, (int)w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x - 1]);
  if (x + 1 < w.w) m = min(m, (int)w.mask[(size_t)(w.y1 + y) * w.mask_w + w.x1 + x + 1]);
  return m > 60;
}

// counters[2l] / [2l+1]: pixels of component l not merged yet that are predicted text / background
__global__ __launch_bounds__(256) void win_accept_count_kernel(const CtdWin* __restrict__ wins, const ctd_band* __restrict__ bands,
                                                               const int* __restrict__ labels, int canvas_w,
                                                               const uint8_t* __restrict__ merged, int merged_w,
                                                               unsigned* __restrict__ counters) {
  const ctd_band bd = bands[blockIdx.y];
  const CtdWin w = wins[bd.win];
  const int npix = w.w * w.h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    const int l = labels[(size_t)(bd.top + y) * canvas_w + x];
    if (l > 0 && merged[(size_t)(bd.mtop + y) * merged_w + x] == 0) atomicAdd(counters + 2 * (size_t)l + (pred_on(w, x, y) ? 0 : 1), 1u);
  }
}

// OR a component into the merged mask iff it is allowed and lowers the xor distance (on > off)
__global__ __launch_bounds__(256) void win_accept_apply_kernel(const CtdWin* __restrict__ wins, const ctd_band* __restrict__ bands,
                                                               const int* __restrict__ labels, int canvas_w,
                                                               const int* __restrict__ stats, const uint8_t* __restrict__ allowed,
                                                               int min_box, uint8_t* __restrict__ merged, int merged_w,
                                                               const unsigned* __restrict__ counters) {
  const ctd_band bd = bands[blockIdx.y];
  const CtdWin w = wins[bd.win];
  const int npix = w.w * w.h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    const int l = labels[(size_t)(bd.top + y) * canvas_w + x];
    if (l <= 0) continue;
    const bool ok = allowed ? allowed[l - 1] != 0 : stats[(size_t)(l - 1) * 5 + 2] * stats[(size_t)(l - 1) * 5 + 3] >= min_box;
    if (ok && counters[2 * (size_t)l] > counters[2 * (size_t)l + 1]) merged[(size_t)(bd.mtop + y) * merged_w + x] = 255;
  }
}

// 3x3 rect dilation inside the window (REFINEMASK_INPAINT, textmask.py:110-111) or a copy; also the
// complement canvas for the hole-filling labelling (:113) and the count of set pixels per window
__global__ __launch_bounds__(256) void win_dilate_kernel(const CtdWin* __restrict__ wins, const int* __restrict__ mtops,
                                                         const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                         uint8_t* __restrict__ comp, int merged_w,
                                                         unsigned* __restrict__ count255, int dilate) {
  __shared__ unsigned red[4];
  const CtdWin w = wins[blockIdx.y];
  const int top = mtops[blockIdx.y];
  const int npix = w.w * w.h;
  unsigned cnt = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    int m = in[(size_t)(top + y) * merged_w + x];
    if (dilate) {
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= w.h) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int xx = x + dx;
          if (xx < 0 || xx >= w.w) continue;
          m = max(m, (int)in[(size_t)(top + yy) * merged_w + xx]);
        }
      }
    }
    out[(size_t)(top + y) * merged_w + x] = (uint8_t)m;
    comp[(size_t)(top + y) * merged_w + x] = (uint8_t)(255 - m);
    cnt += m == 255;
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(count255 + blockIdx.y, t);
  }
}

// refined[y1:y2, x1:x2] |= merged (textmask.py:167); windows may overlap -> word-wide atomic OR
__global__ __launch_bounds__(256) void win_commit_kernel(const CtdWin* __restrict__ wins, const int* __restrict__ mtops,
                                                         const uint8_t* __restrict__ merged, int merged_w,
                                                         uint8_t* __restrict__ page, int page_w) {
  const CtdWin w = wins[blockIdx.y];
  const int top = mtops[blockIdx.y];
  const int npix = w.w * w.h;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
    const int x = i % w.w, y = i / w.w;
    const unsigned v = merged[(size_t)(top + y) * merged_w + x];
    if (!v) continue;
    const size_t idx = (size_t)(w.y1 + y) * page_w + w.x1 + x;
    atomicOr((unsigned*)(page + (idx & ~(size_t)3)), v << (8 * (idx & 3)));
  }
}

}  // namespace

static int win_gx(int max_pix) { return max(1, min(64, (max_pix + 4095) / 4096)); }

void launch_win_accept(const CtdWin* wins_dev, const ctd_band* bands_dev, int nbands, int max_pix, const int* labels,
                       int canvas_w, const int* stats, const uint8_t* allowed, int min_box, uint8_t* merged, int merged_w,
                       unsigned* counters, hipStream_t st) {
  const dim3 g(win_gx(max_pix), nbands);
  hipLaunchKernelGGL(win_accept_count_kernel, g, dim3(256), 0, st, wins_dev, bands_dev, labels, canvas_w, merged, merged_w, counters);
  hipLaunchKernelGGL(win_accept_apply_kernel, g, dim3(256), 0, st, wins_dev, bands_dev, labels, canvas_w, stats, allowed,
                     min_box, merged, merged_w, counters);
}

void launch_win_dilate(const CtdWin* wins_dev, const int* mtops_dev, int n, int max_pix, const uint8_t* in, uint8_t* out,
                       uint8_t* comp, int merged_w, unsigned* count255, int dilate, hipStream_t st) {
  hipLaunchKernelGGL(win_dilate_kernel, dim3(win_gx(max_pix), n), dim3(256), 0, st, wins_dev, mtops_dev, in, out, comp,
                     merged_w, count255, dilate);
}

void launch_win_commit(const CtdWin* wins_dev, const int* mtops_dev, int n, int max_pix, const uint8_t* merged, int merged_w,
                       uint8_t* page, int page_w, hipStream_t st) {
  hipLaunchKernelGGL(win_commit_kernel, dim3(win_gx(max_pix), n), dim3(256), 0, st, wins_dev, mtops_dev, merged, merged_w,
                     page, page_w);
}

void launch_win_hist(const CtdWin* wins_dev, int n, int max_pix, unsigned* hist_dev, hipStream_t st) {
  const int gx = max(1, min(64, (max_pix + 4095) / 4096));
  hipLaunchKernelGGL(win_hist_kernel, dim3(gx, n), dim3(256), 0, st, wins_dev, hist_dev);
}

void launch_win_xor(const CtdWin* wins_dev, const CtdRule* rules_dev, int n, int nrules, int max_pix,
                    unsigned long long* sums_dev, hipStream_t st) {
  const int gx = max(1, min(64, (max_pix + 4095) / 4096));
  hipLaunchKernelGGL(win_xor_kernel, dim3(gx, n), dim3(256), 0, st, wins_dev, rules_dev, nrules, sums_dev);
}

void launch_win_render(const CtdWin* wins_dev, const CtdRule* rules_dev, const int* tops_dev, int nbands, int max_pix,
                       uint8_t* canvas_dev, int canvas_w, hipStream_t st) {
  const int gx = max(1, min(64, (max_pix + 4095) / 4096));
  hipLaunchKernelGGL(win_render_kernel, dim3(gx, nbands), dim3(256), 0, st, wins_dev, rules_dev, tops_dev, nbands,
                     canvas_dev, canvas_w);
}
