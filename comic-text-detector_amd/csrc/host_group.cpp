// Native host code of the block / line grouping: `group_output` (reference utils/textblock.py:421-508)
// with its helpers `examine_textblk` (:302-342), `try_merge_textline` (:344-373), `merge_textlines`
// (:375-388), `split_textblk` (:390-419), `sort_textblk_list` (:267-300), `TextBlock.adjust_bbox`
// (:87-98), `TextBlock.sort_lines` (:100-105) and `union_area` (utils/imgproc_utils.py:13-20).
//
// N is tiny (<= 300 blocks, <= 1000 lines per page) and the arithmetic is scalar float64 with the
// reference's truncation points, so this is host code -- native, so that a batch of pages
// can be grouped from worker threads without the interpreter.
//
// Every float64 expression is evaluated in the reference's operation order (g++ -ffp-contract=off: no
// FMA contraction).  Line coordinates are integers, so the sums, dot products and squared norms the
// reference forms with numpy are exact here as there; square roots and divisions are correctly rounded
// on both sides.  The one step whose last bit is library dependent is `abs(sin(arccos(c)) * d)` of
// `examine_textblk` (:327-328).  numpy's float64 `sin` is libm's; its float64 `arccos` on an AVX-512 host (the
// AVX512_SKX dispatch target) is Intel SVML's `__svml_acos8_ha`, which differs from glibc's `acos` in the last bit of
// ~9 % of arguments (measured, 200 000 random values) -- and lines of one text row tie mathematically, so that last bit
// ORDERS them in `TextBlock.sort_lines` and, through the order, decides splits (round 6: 1 page in ~100 of a seed sweep
// differed in the order of tied lines, 1 in ~350 in its blocks).  `npd::arccos` (np_dispatch.h) therefore calls the very function numpy
// calls when the process has numpy loaded and the CPU takes that dispatch path: the symbol is exported by numpy's
// `_multiarray_umath` module, found among the loaded objects, called on a broadcast vector (SVML is lane-wise).  Without it
// (no numpy in the process, no AVX-512, a numpy built without SVML) numpy itself computes `arccos` with libm, and so does this
// file.  The two operands (c, d) of every line are still handed back so that the Python record carries numpy's own value
// (comic-text-detector_amd/textblock.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/ctd_hip.h"
#include "np_dispatch.h"

namespace {

const double kPi = 3.141592653589793;   // math.pi

struct Line {
  int32_t p[8];   // 4 points (x, y)
  double dist;    // |sin(arccos(dcos)) * dlen|, the line's entry of TextBlock.distance
  double dcos, dlen;
};

struct Dist {
  double dist, dcos, dlen;
};

struct Blk {
  int xyxy[4] = {0, 0, 0, 0};
  int lang = 2;
  bool vertical = false;
  double font = -1;
  bool font_float = false;   // Python type of font_size: int until a merge makes it a float
  int angle = 0;
  double vec[2] = {0, 0};
  double norm = -1;
  bool merged = false;
  double weight = -1;
  std::vector<Line> lines;
  std::vector<Dist> dist;    // TextBlock.distance (its length is NOT always len(lines), see split)
};

// ---- shapely Polygon.intersects for two integer quads (textblock.py:355-356, 400-402) -------------
inline int orient(double ax, double ay, double bx, double by, double cx, double cy) {
  const double v = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
  return (v > 0) - (v < 0);
}
inline bool on_segment(double ax, double ay, double bx, double by, double cx, double cy) {
  return std::min(ax, bx) <= cx && cx <= std::max(ax, bx) && std::min(ay, by) <= cy && cy <= std::max(ay, by);
}
bool inside(double x, double y, const int32_t* q) {
  bool c = false;
  for (int k = 0; k < 4; ++k) {
    const double x0 = q[2 * k], y0 = q[2 * k + 1], x1 = q[2 * ((k + 1) & 3)], y1 = q[2 * ((k + 1) & 3) + 1];
    if ((y0 > y) != (y1 > y) && x < x0 + (y - y0) * (x1 - x0) / (y1 - y0)) c = !c;
  }
  return c;
}
bool quads_intersect(const int32_t* p, const int32_t* q) {
  int pminx = p[0], pmaxx = p[0], pminy = p[1], pmaxy = p[1], qminx = q[0], qmaxx = q[0], qminy = q[1], qmaxy = q[1];
  for (int k = 1; k < 4; ++k) {
    pminx = std::min(pminx, p[2 * k]), pmaxx = std::max(pmaxx, p[2 * k]);
    pminy = std::min(pminy, p[2 * k + 1]), pmaxy = std::max(pmaxy, p[2 * k + 1]);
    qminx = std::min(qminx, q[2 * k]), qmaxx = std::max(qmaxx, q[2 * k]);
    qminy = std::min(qminy, q[2 * k + 1]), qmaxy = std::max(qmaxy, q[2 * k + 1]);
  }
  if (pmaxx < qminx || qmaxx < pminx || pmaxy < qminy || qmaxy < pminy) return false;
  for (int i = 0; i < 4; ++i) {
    const double ax = p[2 * i], ay = p[2 * i + 1], bx = p[2 * ((i + 1) & 3)], by = p[2 * ((i + 1) & 3) + 1];
    for (int j = 0; j < 4; ++j) {
      const double cx = q[2 * j], cy = q[2 * j + 1], dx = q[2 * ((j + 1) & 3)], dy = q[2 * ((j + 1) & 3) + 1];
      const int o1 = orient(ax, ay, bx, by, cx, cy), o2 = orient(ax, ay, bx, by, dx, dy);
      const int o3 = orient(cx, cy, dx, dy, ax, ay), o4 = orient(cx, cy, dx, dy, bx, by);
      if (o1 != o2 && o3 != o4) return true;
      if ((o1 == 0 && on_segment(ax, ay, bx, by, cx, cy)) || (o2 == 0 && on_segment(ax, ay, bx, by, dx, dy)) ||
          (o3 == 0 && on_segment(cx, cy, dx, dy, ax, ay)) || (o4 == 0 && on_segment(cx, cy, dx, dy, bx, by)))
        return true;
    }
  }
  return inside(p[0], p[1], q) || inside(q[0], q[1], p);
}

// ---- mask[y1:y2, x1:x2].mean() / 255 with Python's slice semantics (negative = from the end) -------
inline void py_slice(int a, int b, int n, int& lo, int& hi) {
  if (a < 0) a = std::max(a + n, 0);
  if (b < 0) b = std::max(b + n, 0);
  lo = std::min(a, n);
  hi = std::min(b, n);
}
double mask_score(const uint8_t* mask, int pitch, int im_w, int im_h, int x1, int y1, int x2, int y2) {
  int xa, xb, ya, yb;
  py_slice(x1, x2, im_w, xa, xb);
  py_slice(y1, y2, im_h, ya, yb);
  if (xb <= xa || yb <= ya) return std::numeric_limits<double>::quiet_NaN();   // mean of an empty slice
  unsigned long long s = 0;
  for (int y = ya; y < yb; ++y) {
    const uint8_t* r = mask + (size_t)y * pitch;
    for (int x = xa; x < xb; ++x) s += r[x];
  }
  return (double)s / (double)((long long)(xb - xa) * (yb - ya)) / 255;
}

// ---- TextBlock helpers --------------------------------------------------------------------------
void adjust_bbox(Blk& b, bool with_bbox) {   // textblock.py:87-98
  int lo[2] = {b.lines[0].p[0], b.lines[0].p[1]}, hi[2] = {lo[0], lo[1]};
  for (const Line& l : b.lines)
    for (int k = 0; k < 4; ++k) {
      lo[0] = std::min(lo[0], l.p[2 * k]), hi[0] = std::max(hi[0], l.p[2 * k]);
      lo[1] = std::min(lo[1], l.p[2 * k + 1]), hi[1] = std::max(hi[1], l.p[2 * k + 1]);
    }
  if (with_bbox) {
    lo[0] = std::min(lo[0], b.xyxy[0]), lo[1] = std::min(lo[1], b.xyxy[1]);
    hi[0] = std::max(hi[0], b.xyxy[2]), hi[1] = std::max(hi[1], b.xyxy[3]);
  }
  b.xyxy[0] = lo[0], b.xyxy[1] = lo[1], b.xyxy[2] = hi[0], b.xyxy[3] = hi[1];
}

inline bool dist_less(double a, double b) {   // numpy sort order: NaN last
  if (std::isnan(a)) return false;
  if (std::isnan(b)) return true;
  return a < b;
}

// examine_textblk (textblock.py:302-342)
void examine(Blk& b, int im_w, int im_h, bool sort) {
  const int n = (int)b.lines.size();
  double v[2] = {0, 0}, h[2] = {0, 0};
  for (const Line& l : b.lines) {
    double mid[4][2];
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 2; ++c) mid[k][c] = ((double)l.p[2 * ((k + 1) & 3) + c] + (double)l.p[2 * k + c]) / 2;
    for (int c = 0; c < 2; ++c) {
      v[c] += mid[2][c] - mid[0][c];
      h[c] += mid[1][c] - mid[3][c];
    }
  }
  const double nv = std::sqrt(v[0] * v[0] + v[1] * v[1]), nh = std::sqrt(h[0] * h[0] + h[1] * h[1]);
  const bool vertical = b.lang == 1 ? nv > nh : nv > nh * 2;                      // :312-315
  const double* pvec = vertical ? v : h;
  const double pnorm = vertical ? nv : nh;
  const double font = std::nearbyint((vertical ? nh : nv) / n);                     // int(round(...)), half to even
  const int rot = (int)(std::atan2(pvec[1], pvec[0]) / kPi * 180);                  // :326, truncation
  b.dist.resize(n);
  for (int i = 0; i < n; ++i) {
    Line& l = b.lines[i];
    double d[2] = {((double)l.p[0] + (double)l.p[4]) / 2, ((double)l.p[1] + (double)l.p[5]) / 2};
    if (vertical) d[0] = d[0] - (double)im_w;                                       // origin (im_w, 0): right-to-left
    const double len = std::sqrt(d[0] * d[0] + d[1] * d[1]);
    const double c = (d[0] * pvec[0] + d[1] * pvec[1]) / (len * pnorm);
    l.dlen = len;
    l.dcos = c;
    l.dist = std::fabs(std::sin(npd::arccos(c)) * len);
    b.dist[i] = {l.dist, l.dcos, l.dlen};
  }
  b.angle = vertical ? rot - 90 : rot;
  if (std::abs(b.angle) < 3) b.angle = 0;
  b.font = font;
  b.font_float = false;
  b.vertical = vertical;
  b.vec[0] = pvec[0], b.vec[1] = pvec[1];
  b.norm = pnorm;
  (void)im_h;
  if (sort) {                                                                       // sort_lines (:100-105)
    std::vector<double> key(n);
    std::vector<long> idx(n);
    for (int i = 0; i < n; ++i) key[i] = b.lines[i].dist, idx[i] = i;
    if (n > 1 && npd::argsort_f64(key.data(), idx.data(), n)) {                      // numpy's own default-kind argsort (np_dispatch.h)
      std::vector<Line> sorted(n);
      for (int i = 0; i < n; ++i) sorted[i] = b.lines[idx[i]];
      b.lines.swap(sorted);
    } else {
      std::stable_sort(b.lines.begin(), b.lines.end(), [](const Line& x, const Line& y) { return dist_less(x.dist, y.dist); });
    }
    for (int i = 0; i < n; ++i) b.dist[i] = {b.lines[i].dist, b.lines[i].dcos, b.lines[i].dlen};
  }
}

// try_merge_textline (textblock.py:344-373)
bool try_merge(Blk& a, Blk& b, double fntsize_tol = 1.3, double distance_tol = 2) {
  if (b.merged) return false;
  const double ratio = a.font / b.font;
  const double na = (double)a.lines.size(), nb = (double)b.lines.size();
  const double avg = (a.font * na + b.font * nb) / (na + nb);
  const double vsum[2] = {a.vec[0] + b.vec[0], a.vec[1] + b.vec[1]};
  const double cosv = (a.vec[0] * b.vec[0] + a.vec[1] * b.vec[1]) / a.norm / b.norm;
  const double gap = b.dist.back().dist - a.dist.back().dist;
  const Line& la = a.lines.back();
  const Line& lb = b.lines.back();
  const double dx = (double)(lb.p[0] - la.p[0]), dy = (double)(lb.p[1] - la.p[1]);
  const double gap_p1 = std::sqrt(dx * dx + dy * dy);
  if (!quads_intersect(la.p, lb.p)) {
    if (ratio > fntsize_tol || 1 / ratio > fntsize_tol) return false;
    if (std::fabs(cosv) < 0.866) return false;
    if (gap > distance_tol * avg || gap_p1 > avg * 2.5) return false;
  }
  a.lines.push_back(b.lines[0]);
  a.vec[0] = vsum[0], a.vec[1] = vsum[1];
  a.angle = (int)std::nearbyint(std::atan2(vsum[1], vsum[0]) * (180.0 / kPi));     // int(round(np.rad2deg(...)))
  if (a.vertical) a.angle -= 90;
  a.norm = std::sqrt(vsum[0] * vsum[0] + vsum[1] * vsum[1]);
  a.dist.push_back(b.dist.back());
  a.font = avg;
  a.font_float = true;
  b.merged = true;
  return true;
}

// merge_textlines (textblock.py:375-388); returns indices into `pool`
void merge_textlines(std::vector<Blk>& pool, std::vector<int>& out) {
  out.clear();
  if (pool.size() < 2) {
    for (size_t i = 0; i < pool.size(); ++i) out.push_back((int)i);
    return;
  }
  std::vector<int> order(pool.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return pool[x].dist[0].dist < pool[y].dist[0].dist; });
  for (size_t i = 0; i < order.size(); ++i) {
    Blk& cur = pool[order[i]];
    if (cur.merged) continue;
    for (size_t j = i + 1; j < order.size(); ++j) try_merge(cur, pool[order[j]]);
    out.push_back(order[i]);
  }
  for (int i : out) adjust_bbox(pool[i], false);
}

// split_textblk (textblock.py:390-419).  Appends the parts to `out`; returns whether it split.
bool split_textblk(Blk& blk, std::vector<Blk>& out) {
  const double font = blk.font;
  const std::vector<Dist> dist = blk.dist;
  const int32_t fx = blk.lines[0].p[0], fy = blk.lines[0].p[1];
  const Line first = blk.lines[0];
  auto key = [&](const Line& l) {
    const double dx = (double)(l.p[0] - fx), dy = (double)(l.p[1] - fy);
    return std::sqrt(dx * dx + dy * dy);
  };
  std::stable_sort(blk.lines.begin(), blk.lines.end(), [&](const Line& a, const Line& b) { return key(a) < key(b); });
  const std::vector<Line>& lines = blk.lines;
  const size_t start = out.size();
  Blk cur = blk;               // copy.deepcopy(blk): every field, the full `distance` array included
  cur.lines.clear();
  cur.lines.push_back(first);
  out.push_back(cur);
  for (size_t j = 0; j + 1 < lines.size(); ++j) {
    const Line& line = lines[j + 1];
    bool split = false;
    if (!quads_intersect(lines[j].p, line.p)) {
      const double gap = std::fabs(dist[j + 1].dist - dist[j].dist);
      if (gap > font * 2) split = true;
      else if (blk.vertical && std::abs(blk.angle) < 15) {
        if (out.back().lines.size() > 1 || gap > font)
          split = std::abs(lines[j].p[1] - line.p[1]) > font;
      }
    }
    if (split) {
      Blk nb = out.back();     // copy.deepcopy(current_blk)
      nb.lines.clear();
      nb.lines.push_back(line);
      out.push_back(nb);
    } else {
      out.back().lines.push_back(line);
    }
  }
  if (out.size() - start > 1) {
    for (size_t i = start; i < out.size(); ++i) adjust_bbox(out[i], false);
    return true;
  }
  return false;
}

// sort_textblk_list (textblock.py:267-300)
void sort_blocks(std::vector<Blk>& blks, int im_w_in, int im_h) {
  if (blks.empty()) return;
  int nja = 0;
  for (const Blk& b : blks) nja += b.lang == 1;
  const bool rtl = (double)nja > (double)blks.size() / 2;
  const double full_w = im_w_in;
  const bool halved = im_w_in > im_h;
  const double im_w = halved ? (double)im_w_in / 2 : (double)im_w_in;
  const int gy = 4, gx = 3;
  const double area = (double)im_h * im_w;
  for (Blk& b : blks) {
    double cx = ((double)b.xyxy[0] + (double)b.xyxy[2]) / 2;
    if (rtl) cx = halved ? full_w - cx : im_w - cx;
    const int ix = (int)(cx / im_w * gx);
    const double cy = ((double)b.xyxy[1] + (double)b.xyxy[3]) / 2;
    const int iy = (int)(cy / im_h * gy);
    double w = (double)(iy * gx + ix) * area + 1.2 * (cx - (double)ix * im_w / gx) + (cy - (double)(iy * im_h) / gy);
    if (halved && ix >= gx) w += area * gy * gx;
    b.weight = w;
  }
  std::stable_sort(blks.begin(), blks.end(), [](const Blk& a, const Blk& b) { return a.weight < b.weight; });
}

}  // namespace

extern "C" int ctd_group_output(const int32_t* blines, const int32_t* cls, int32_t n_blk, const int32_t* lines_in,
                                int32_t n_lines, int32_t im_w, int32_t im_h, const uint8_t* mask, int32_t mask_pitch,
                                ctd_blk* blks_out, int32_t blk_cap, int32_t* lines_out, int32_t line_cap,
                                double* dist_out, int32_t dist_cap, int32_t* n_blk_out, int32_t* n_lines_out,
                                int32_t* n_dist_out) {
  if (n_blk < 0 || n_lines < 0 || (n_blk && (!blines || !cls)) || (n_lines && !lines_in) || !blks_out || !lines_out ||
      !dist_out || !n_blk_out || !n_lines_out || !n_dist_out || im_w < 1 || im_h < 1 || (mask && mask_pitch < im_w))
    return CTD_ERR_INVALID;
  const double bbox_thr = 0.4, mask_thr = 0.1;
  std::vector<Blk> blk_list(n_blk);
  for (int i = 0; i < n_blk; ++i) {
    for (int k = 0; k < 4; ++k) blk_list[i].xyxy[k] = blines[4 * i + k];
    const int c = cls[i];
    blk_list[i].lang = (c >= 0 && c <= 2) ? c : 2;
  }
  std::vector<Blk> hor, ver;
  // step 1: assign every line to the block that covers most of its bbox, else keep it if the mask agrees (:431-457)
  for (int i = 0; i < n_lines; ++i) {
    const int32_t* q = lines_in + 8 * i;
    int x1 = q[0], x2 = q[0], y1 = q[1], y2 = q[1];
    for (int k = 1; k < 4; ++k) {
      x1 = std::min(x1, q[2 * k]), x2 = std::max(x2, q[2 * k]);
      y1 = std::min(y1, q[2 * k + 1]), y2 = std::max(y2, q[2 * k + 1]);
    }
    const double area = (double)((int32_t)((y2 - y1) * (x2 - x1)));      // int32 product like numpy's
    double best = -1;
    int best_j = -1;
    for (int j = 0; j < n_blk; ++j) {
      const int* b = blk_list[j].xyxy;
      const int ix1 = std::max(b[0], x1), iy1 = std::max(b[1], y1), ix2 = std::min(b[2], x2), iy2 = std::min(b[3], y2);
      const double inter = (iy2 < iy1 || ix2 < ix1) ? -1.0 : (double)(iy2 - iy1) * (double)(ix2 - ix1);
      const double score = inter / area;                                  // x/0 -> inf / nan as in numpy
      if (best < score) best = score, best_j = j;                         // first maximum, strict '<' (:440-442)
    }
    Line ln;
    std::memcpy(ln.p, q, sizeof(ln.p));
    ln.dist = ln.dcos = ln.dlen = 0;
    if (best > bbox_thr) {
      blk_list[best_j].lines.push_back(ln);
      continue;
    }
    if (mask && mask_score(mask, mask_pitch, im_w, im_h, x1, y1, x2, y2) < mask_thr) continue;
    Blk t;
    t.xyxy[0] = x1, t.xyxy[1] = y1, t.xyxy[2] = x2, t.xyxy[3] = y2;
    t.lines.push_back(ln);
    examine(t, im_w, im_h, false);
    (t.vertical ? ver : hor).push_back(t);
  }
  // step 2: filter blocks, sort and split their lines (:460-485)
  std::vector<Blk> fin;
  for (Blk& t : blk_list) {
    if (t.lines.empty()) {
      const int x1 = t.xyxy[0], y1 = t.xyxy[1], x2 = t.xyxy[2], y2 = t.xyxy[3];
      if (mask && mask_score(mask, mask_pitch, im_w, im_h, x1, y1, x2, y2) < mask_thr) continue;
      Line ln;                                                            // xywh2xyxypoly (imgproc_utils.py:31-37)
      const int32_t p[8] = {x1, y1, x2, y1, x2, y2, x1, y2};
      std::memcpy(ln.p, p, sizeof(p));
      ln.dist = ln.dcos = ln.dlen = 0;
      t.lines.push_back(ln);
    }
    examine(t, im_w, im_h, true);
    bool was_split = false;
    if (t.lines.size() > 1 && (t.lang == 1 || t.vertical)) {
      was_split = split_textblk(t, fin);
    } else {
      fin.push_back(t);
    }
    if (!was_split) adjust_bbox(fin.back(), true);      // exactly one part when nothing was split
  }
  // step 3: merge the scattered lines, order the blocks on the page grid (:488-491)
  std::vector<int> idx;
  merge_textlines(hor, idx);
  for (int i : idx) fin.push_back(hor[i]);
  merge_textlines(ver, idx);
  for (int i : idx) fin.push_back(ver[i]);
  sort_blocks(fin, im_w, im_h);
  // English lines get a small margin (:492-506)
  for (Blk& t : fin) {
    if (t.lang == 0 && !t.vertical && !t.lines.empty()) {
      const int grow = std::max((int)(t.font * 0.1), 2);
      const double rad = (double)t.angle * (kPi / 180.0);                  // np.deg2rad
      const double sx = std::sin(rad), cy = std::cos(rad);
      static const int sgn[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
      for (Line& l : t.lines)
        for (int k = 0; k < 4; ++k) {
          double x = (double)l.p[2 * k] + (double)sgn[k][0] * sx * (double)grow;
          double y = (double)l.p[2 * k + 1] + (double)sgn[k][1] * cy * (double)grow;
          x = std::min(std::max(x, 0.0), (double)(im_w - 1));
          y = std::min(std::max(y, 0.0), (double)(im_h - 1));
          l.p[2 * k] = (int32_t)x;                                        // astype(np.int64): truncation
          l.p[2 * k + 1] = (int32_t)y;
        }
      t.font = t.font + grow;
    }
  }
  // ---- flatten ------------------------------------------------------------------------------------
  int nb = 0, nl = 0, nd = 0;
  for (const Blk& t : fin) {
    if (nb >= blk_cap || nl + (int)t.lines.size() > line_cap || nd + (int)t.dist.size() > dist_cap) return CTD_ERR_NOMEM;
    ctd_blk& o = blks_out[nb++];
    for (int k = 0; k < 4; ++k) o.xyxy[k] = t.xyxy[k];
    o.language = t.lang;
    o.vertical = t.vertical;
    o.angle = t.angle;
    o.font_is_float = t.font_float;
    o.font_size = t.font;
    o.vec[0] = t.vec[0], o.vec[1] = t.vec[1];
    o.norm = t.norm;
    o.weight = t.weight;
    o.merged = t.merged;
    o.line_off = nl, o.n_lines = (int)t.lines.size();
    o.dist_off = nd, o.n_dist = (int)t.dist.size();
    for (const Line& l : t.lines) std::memcpy(lines_out + 8 * (size_t)nl++, l.p, sizeof(l.p));
    for (const Dist& d : t.dist) {
      dist_out[3 * (size_t)nd] = d.dist, dist_out[3 * (size_t)nd + 1] = d.dcos, dist_out[3 * (size_t)nd + 2] = d.dlen;
      ++nd;
    }
  }
  *n_blk_out = nb, *n_lines_out = nl, *n_dist_out = nd;
  return CTD_OK;
}
