// Native host decisions of the mask refinement (reference utils/textmask.py:16-71): everything that is
// O(256) per text-block window -- the top-k grey colours picked from the histogram of the selected grey
// pixels (np.histogram(bins=255) + get_topk_color), the Otsu threshold of a channel histogram
// (cv2.threshold(THRESH_OTSU)), the integer bounds cv2.inRange derives from double scalars, and the
// polarity / ordering of the candidate masks (minxor_thresh, the sorts of :49 and :74).
// The O(pixels) work (histograms, xor distances, candidate masks, labelling, merging) is on the GPU
// (csrc/kernels_tail.hip); csrc/tail.hip drives both.  Built with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/ctd_hip.h"
#include "host_refine.h"
#include "np_dispatch.h"

// np.histogram(px, bins=255) of the multiset given by `hist` (count per grey level 0..255) followed by
// get_topk_color(edges, counts, k=3, color_var=10, bin_tol=0.001) (textmask.py:16-27, 61-62).
// numpy: range = (min, max) of the data ((0, 1) for no data; +-0.5 if min == max), edges =
// linspace(first, last, 256) = i * step + first with edges[255] = last, a value falls into the bin
// [edges[i], edges[i+1]) (the last bin is closed) -- numpy's index estimate is corrected against the
// edges, so only the edges decide.  The descending order of the counts is numpy's own default-kind argsort where the
// process has it (np.argsort's default is unstable for ties, SURVEY / DESIGN section 5), a stable sort elsewhere.
extern "C" int ctd_topk_colors(const int64_t* hist, double* colors) {
  int lo = -1, hi = -1;
  for (int v = 0; v < 256; ++v)
    if (hist[v] > 0) {
      if (lo < 0) lo = v;
      hi = v;
    }
  double first, last;
  if (lo < 0) first = 0.0, last = 1.0;
  else first = (double)lo, last = (double)hi;
  if (first == last) first -= 0.5, last += 0.5;
  const int nb = 255;
  double edges[256];
  const double step = (last - first) / (double)nb;
  for (int i = 0; i <= nb; ++i) edges[i] = (double)i * step + first;
  edges[nb] = last;
  int64_t counts[255];
  std::memset(counts, 0, sizeof(counts));
  int64_t total = 0;
  if (lo >= 0) {
    const double denom = last - first;
    for (int v = lo; v <= hi; ++v) {
      if (!hist[v]) continue;
      const double a = (double)v;
      int idx = (int)(((a - first) / denom) * (double)nb);
      if (idx == nb) idx -= 1;
      if (a < edges[idx]) idx -= 1;
      if (a >= edges[idx + 1] && idx != nb - 1) idx += 1;
      counts[idx] += hist[v];
      total += hist[v];
    }
  }
  // np.argsort(counts * -1), numpy's default kind: its own function where the process has it (np_dispatch.h: bins of EQUAL count
  // beyond 16 elements come out in x86-simd-sort's order there), a stable sort elsewhere
  int order[255];
  long keys[255], idx[255];
  for (int i = 0; i < nb; ++i) keys[i] = -(long)counts[i], idx[i] = i;
  if (npd::argsort_i64(keys, idx, nb)) {
    for (int i = 0; i < nb; ++i) order[i] = (int)idx[i];
  } else {
    for (int i = 0; i < nb; ++i) order[i] = i;
    std::stable_sort(order, order + nb, [&](int a, int b) { return counts[a] > counts[b]; });
  }
  int n = 0;
  colors[n++] = edges[order[0]];
  const double tol = (double)total * 0.001;
  for (int k = 1; k < nb; ++k) {
    const double c = edges[order[k]];
    double dmin = std::fabs(colors[0] - c);
    for (int j = 1; j < n; ++j) dmin = std::min(dmin, std::fabs(colors[j] - c));
    if (dmin > 10) colors[n++] = c;
    if (n >= 3 || (double)counts[order[k]] < tol) break;
  }
  return n;
}

// OpenCV getThreshVal_Otsu_8u (imgproc/src/thresh.cpp): sequential class statistics in double,
// first strict maximum of the between-class variance.
extern "C" int ctd_otsu_from_hist(const int64_t* h) {
  int64_t n = 0;
  for (int i = 0; i < 256; ++i) n += h[i];
  if (n <= 0) return 0;
  const double scale = 1. / (double)n;
  double mu = 0;
  for (int i = 0; i < 256; ++i) mu += i * (double)h[i];
  mu *= scale;
  double mu1 = 0, q1 = 0, max_sigma = 0;
  int max_val = 0;
  const double eps = 1.1920928955078125e-07;   // FLT_EPSILON
  for (int i = 0; i < 256; ++i) {
    const double p_i = (double)h[i] * scale;
    mu1 *= q1;
    q1 += p_i;
    const double q2 = 1. - q1;
    if (std::min(q1, q2) < eps || std::max(q1, q2) > 1. - eps) continue;
    mu1 = (mu1 + i * p_i) / q1;
    const double mu2 = (mu - q1 * mu1) / q2;
    const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
    if (sigma > max_sigma) max_sigma = sigma, max_val = i;
  }
  return max_val;
}

// cv2.inRange(u8 image, lo, hi) with double scalars (OpenCV core/src/arithm.cpp, scalar branch): both
// bounds go to int32 through cvRound (half to even); lb > ub, lb > 255 or ub < 0 matches nothing;
// otherwise the bounds saturate to [0, 255].
extern "C" void ctd_inrange_bounds(double lo, double hi, int32_t* lb, int32_t* ub) {
  const long long ilo = std::llrint(lo), ihi = std::llrint(hi);   // default rounding mode: half to even
  if (ilo > ihi || ilo > 255 || ihi < 0) {
    *lb = 1, *ub = 0;
    return;
  }
  *lb = (int32_t)std::max<long long>(ilo, 0);
  *ub = (int32_t)std::min<long long>(ihi, 255);
}

// Rules of one window from its 4 histograms (grey of the selected pixels, B, G, R of the window):
// rules[0..2] grey ranges around the top-k colours (textmask.py:63-69), rules[3..5] Otsu thresholds.
void refine_rules(const uint32_t* hist4, RRule rules[6]) {
  int64_t h[256];
  for (int v = 0; v < 256; ++v) h[v] = hist4[v];
  double top[3];
  const int nt = ctd_topk_colors(h, top);
  for (int k = 0; k < 3; ++k) {
    rules[k].kind = -1, rules[k].lo = rules[k].hi = 0;
    if (k < nt) {
      const double c_top = std::min(top[k] + 30, 255.0);
      const double c_bottom = c_top - 60;
      rules[k].kind = 0;
      ctd_inrange_bounds(c_bottom, c_top, &rules[k].lo, &rules[k].hi);
    }
  }
  for (int ch = 0; ch < 3; ++ch) {
    for (int v = 0; v < 256; ++v) h[v] = hist4[256 * (1 + ch) + v];
    rules[3 + ch].kind = 1 + ch;
    rules[3 + ch].lo = ctd_otsu_from_hist(h);
    rules[3 + ch].hi = 0;
  }
}

// Candidates of one window in merge order from the xor sums of its 6 rules: polarity by
// `minxor_thresh` (:29-41, the negative wins only if strictly closer), the best Otsu channel (first
// minimum in B, G, R order = the stable sort of :49), then the stable sort by distance of :74.
int refine_candidates(const RRule rules[6], const uint64_t sums[6], long long npix, RCand out[4]) {
  int n = 0;
  auto pick = [&](int k) {
    RCand c;
    const unsigned long long d_pos = sums[k], d_neg = 255ull * (unsigned long long)npix - d_pos;
    c.rule = k;
    c.invert = d_neg < d_pos;
    c.dist = c.invert ? d_neg : d_pos;
    return c;
  };
  for (int k = 0; k < 3; ++k)
    if (rules[k].kind >= 0) out[n++] = pick(k);
  RCand best = pick(3);
  for (int k = 4; k < 6; ++k) {
    const RCand c = pick(k);
    if (c.dist < best.dist) best = c;
  }
  out[n++] = best;
  std::stable_sort(out, out + n, [](const RCand& a, const RCand& b) { return a.dist < b.dist; });
  return n;
}
