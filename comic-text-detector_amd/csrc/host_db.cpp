// Host-side (no device code) O(#components) geometry of the DB text-line stage:
// `SegDetectorRepresenter.boxes_from_bitmap` (reference utils/db_utils.py:134-211) after the two
// GPU labelling passes (`ctd_ccl`) have turned the bitmap into components.  The O(pixels) work is
// on the GPU; what is left per contour is a hull, a calipers rectangle, a polygon mean and the
// unclip -- scalar double arithmetic, done here natively instead of in the Python host.
//
// Built with g++ -ffp-contract=off: every expression is evaluated as written (mul, then add), so
// the results do not depend on FMA availability.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/ctd_hip.h"

namespace {

struct P {
  double x, y;
};
struct Pf {
  float x, y;
};

inline bool lt_xy(const P& a, const P& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); }

void chain(const std::vector<P>& p, bool reverse, std::vector<P>& out) {
  out.clear();
  const int n = (int)p.size();
  for (int i = 0; i < n; ++i) {
    const P& q = reverse ? p[n - 1 - i] : p[i];
    while (out.size() >= 2) {
      const P& a = out[out.size() - 2];
      const P& b = out[out.size() - 1];
      if ((b.x - a.x) * (q.y - a.y) - (b.y - a.y) * (q.x - a.x) <= 0) out.pop_back();
      else break;
    }
    out.push_back(q);
  }
}

// strict convex hull, counter-clockwise in x-right / y-down axes, no repeated vertex
void hull(std::vector<P>& pts, std::vector<P>& h, std::vector<P>& tmp) {
  std::sort(pts.begin(), pts.end(), lt_xy);
  pts.erase(std::unique(pts.begin(), pts.end(), [](const P& a, const P& b) { return a.x == b.x && a.y == b.y; }),
            pts.end());
  h.clear();
  if (pts.size() <= 2) {
    h = pts;
    return;
  }
  chain(pts, false, tmp);
  h.assign(tmp.begin(), tmp.end() - 1);
  chain(pts, true, tmp);
  h.insert(h.end(), tmp.begin(), tmp.end() - 1);
}

// minimum-area enclosing rectangle of the hull `h`, grown by `grow` on every side: all hull
// edges are evaluated (the optimum shares a side with the hull), first minimum in edge order.
void min_area_box(const std::vector<P>& h, double grow, Pf box[4], double& bw, double& bh) {
  const int n = (int)h.size();
  if (n == 0) {
    std::memset(box, 0, sizeof(Pf) * 4);
    bw = bh = 0;
    return;
  }
  if (n == 1) {
    const double cx = h[0].x, cy = h[0].y, g = grow;
    box[0] = {(float)(cx - g), (float)(cy - g)};
    box[1] = {(float)(cx + g), (float)(cy - g)};
    box[2] = {(float)(cx + g), (float)(cy + g)};
    box[3] = {(float)(cx - g), (float)(cy + g)};
    bw = bh = 2 * g;
    return;
  }
  const int ne = n == 2 ? 1 : n;
  bool have = false;
  double best_area = 0, b_lou = 0, b_hiu = 0, b_lov = 0, b_hiv = 0, b_ux = 0, b_uy = 0;
  for (int i = 0; i < ne; ++i) {
    const double ex = h[(i + 1) % n].x - h[i].x, ey = h[(i + 1) % n].y - h[i].y;
    const double L = std::hypot(ex, ey);
    if (!(L > 0)) continue;
    const double ux = ex / L, uy = ey / L, vx = -uy, vy = ux;
    double lou = 0, hiu = 0, lov = 0, hiv = 0;
    for (int j = 0; j < n; ++j) {
      const double pu = h[j].x * ux + h[j].y * uy;
      const double pv = h[j].x * vx + h[j].y * vy;
      if (j == 0) {
        lou = hiu = pu;
        lov = hiv = pv;
      } else {
        lou = std::min(lou, pu);
        hiu = std::max(hiu, pu);
        lov = std::min(lov, pv);
        hiv = std::max(hiv, pv);
      }
    }
    lou -= grow;
    hiu += grow;
    lov -= grow;
    hiv += grow;
    const double area = (hiu - lou) * (hiv - lov);
    // first minimum in edge order with a RELATIVE margin (oracle/cv_ref.py min_area_box: mathematically equal areas of two
    // hull edges must not be told apart by the rounding noise of the projections)
    if (!have || area < best_area * (1.0 - 1e-9)) {
      have = true;
      best_area = area;
      b_lou = lou, b_hiu = hiu, b_lov = lov, b_hiv = hiv, b_ux = ux, b_uy = uy;
    }
  }
  const double ux = b_ux, uy = b_uy, vx = -b_uy, vy = b_ux;
  box[0] = {(float)(ux * b_lou + vx * b_lov), (float)(uy * b_lou + vy * b_lov)};
  box[1] = {(float)(ux * b_hiu + vx * b_lov), (float)(uy * b_hiu + vy * b_lov)};
  box[2] = {(float)(ux * b_hiu + vx * b_hiv), (float)(uy * b_hiu + vy * b_hiv)};
  box[3] = {(float)(ux * b_lou + vx * b_hiv), (float)(uy * b_lou + vy * b_hiv)};
  bw = b_hiu - b_lou;
  bh = b_hiv - b_lov;
}

// `get_mini_boxes` ordering (reference utils/db_utils.py:178-194): stable sort by x, then TL TR BR BL
void order_box(Pf b[4]) {
  Pf s[4] = {b[0], b[1], b[2], b[3]};
  std::stable_sort(s, s + 4, [](const Pf& a, const Pf& c) { return a.x < c.x; });
  const int i1 = s[1].y > s[0].y ? 0 : 1, i4 = 1 - i1;
  const int i2 = s[3].y > s[2].y ? 2 : 3, i3 = 5 - i2;
  b[0] = s[i1], b[1] = s[i2], b[2] = s[i3], b[3] = s[i4];
}

// numpy's pairwise float64 summation (what ndarray.mean() of a contiguous float64 array does)
double pairwise(const double* a, size_t n) {
  if (n < 8) {
    double r = 0.0;
    for (size_t i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    size_t i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  size_t n2 = n / 2;
  n2 -= n2 % 8;
  return pairwise(a, n2) + pairwise(a + n2, n - n2);
}

// `pyclipper.PyclipperOffset().AddPath(quad, JT_ROUND, ET_CLOSEDPOLYGON); Execute(delta)` for one convex
// closed polygon and delta > 0 as Clipper 6.4.2 (the library pyclipper wraps) builds it: duplicate
// stripping of `ClipperOffset::AddPath`, `FixOrientations`, the step count of `DoOffset` from the arc
// tolerance (0.25), per-vertex `OffsetPoint` -> `DoRound`, every point rounded to the integer grid.
// The closing `Clipper::Execute(ctUnion, pftPositive)` leaves the ring of a convex input as it is
// (up to its start vertex and collinear points, irrelevant to the min-area rectangle taken next).
inline long long clip_round(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }

void clipper_offset_round(const long long (*in)[2], int n_in, double delta, std::vector<P>& out) {
  out.clear();
  long long poly[8][2];
  int hi = n_in - 1;
  while (hi > 0 && in[0][0] == in[hi][0] && in[0][1] == in[hi][1]) --hi;
  int n = 0;
  for (int i = 0; i <= hi && n < 8; ++i)
    if (n == 0 || poly[n - 1][0] != in[i][0] || poly[n - 1][1] != in[i][1]) poly[n][0] = in[i][0], poly[n][1] = in[i][1], ++n;
  if (n < 3) return;
  double a = 0;
  for (int i = 0, j = n - 1; i < n; j = i++) a += ((double)poly[j][0] + poly[i][0]) * ((double)poly[j][1] - poly[i][1]);
  if (-a * 0.5 < 0)   // FixOrientations
    for (int i = 0; i < n / 2; ++i) std::swap(poly[i][0], poly[n - 1 - i][0]), std::swap(poly[i][1], poly[n - 1 - i][1]);
  const double kPi = 3.141592653589793238, kTwoPi = kPi * 2;
  const double arc_tol = 0.25;
  const double y = arc_tol > std::fabs(delta) * 0.25 ? std::fabs(delta) * 0.25 : arc_tol;
  double steps = kPi / std::acos(1 - y / std::fabs(delta));
  if (steps > std::fabs(delta) * kPi) steps = std::fabs(delta) * kPi;
  double m_sin = std::sin(kTwoPi / steps);
  const double m_cos = std::cos(kTwoPi / steps);
  const double steps_per_rad = steps / kTwoPi;
  if (delta < 0) m_sin = -m_sin;
  double nx[8], ny[8];
  for (int i = 0; i < n; ++i) {
    const int k = (i + 1) % n;
    double dx = (double)(poly[k][0] - poly[i][0]), dy = (double)(poly[k][1] - poly[i][1]);
    const double f = 1.0 / std::sqrt(dx * dx + dy * dy);
    dx *= f;
    dy *= f;
    nx[i] = dy, ny[i] = -dx;
  }
  auto emit = [&](int j, double X, double Y) {
    out.push_back({(double)clip_round(poly[j][0] + X * delta), (double)clip_round(poly[j][1] + Y * delta)});
  };
  int k = n - 1;
  for (int j = 0; j < n; ++j) {
    double sin_a = nx[k] * ny[j] - nx[j] * ny[k];
    bool done = false;
    if (std::fabs(sin_a * delta) < 1.0) {
      const double cos_a = nx[k] * nx[j] + ny[j] * ny[k];
      if (cos_a > 0) {
        emit(j, nx[k], ny[k]);
        done = true;
      }
    } else if (sin_a > 1.0) sin_a = 1.0;
    else if (sin_a < -1.0) sin_a = -1.0;
    if (!done) {
      if (sin_a * delta < 0) {
        emit(j, nx[k], ny[k]);
        out.push_back({(double)poly[j][0], (double)poly[j][1]});
        emit(j, nx[j], ny[j]);
      } else {   // DoRound
        const double ang = std::atan2(sin_a, nx[k] * nx[j] + ny[k] * ny[j]);
        const int nst = std::max((int)clip_round(steps_per_rad * std::fabs(ang)), 1);
        double X = nx[k], Y = ny[k];
        for (int i = 0; i < nst; ++i) {
          emit(j, X, Y);
          const double X2 = X;
          X = X * m_cos - m_sin * Y;
          Y = X2 * m_sin + Y * m_cos;
        }
        emit(j, nx[j], ny[j]);
      }
    }
    k = j;
  }
}

struct Item {
  long long key;
  int kind, label;
};

// Pixels of the filled contour polygon, as prob values in row-major order.  m (rh x rw): 1 = the
// traced set (component, or hole), 3 = the hole's border ring, 0 = other.
//   outer border (hole_border = false): the component plus everything it encloses -- pixels not
//     reachable from the crop frame by 4-connected steps through non-component pixels;
//   hole border (hole_border = true): the ring plus the hole with everything the HOLE encloses --
//     pixels not reachable from the crop frame by 8-connected steps through non-hole pixels
//     (foreground the ring merely surrounds, e.g. the inside of a peninsula, stays outside).
void filled_values(std::vector<uint8_t>& m, int rh, int rw, bool hole_border, const float* prob, int W, int x0,
                   int y0, std::vector<int>& stack, std::vector<double>& vals) {
  stack.clear();
  auto push = [&](int y, int x) {
    uint8_t& c = m[(size_t)y * rw + x];
    if (c != 1 && !(c & 4)) {
      c |= 4;
      stack.push_back(y * rw + x);
    }
  };
  for (int x = 0; x < rw; ++x) push(0, x), push(rh - 1, x);
  for (int y = 0; y < rh; ++y) push(y, 0), push(y, rw - 1);
  while (!stack.empty()) {
    const int i = stack.back();
    stack.pop_back();
    const int y = i / rw, x = i % rw;
    if (y > 0) push(y - 1, x);
    if (y + 1 < rh) push(y + 1, x);
    if (x > 0) push(y, x - 1);
    if (x + 1 < rw) push(y, x + 1);
    if (hole_border) {
      if (y > 0 && x > 0) push(y - 1, x - 1);
      if (y > 0 && x + 1 < rw) push(y - 1, x + 1);
      if (y + 1 < rh && x > 0) push(y + 1, x - 1);
      if (y + 1 < rh && x + 1 < rw) push(y + 1, x + 1);
    }
  }
  vals.clear();
  for (int y = 0; y < rh; ++y) {
    const float* pr = prob + (size_t)(y0 + y) * W + x0;
    const uint8_t* mr = m.data() + (size_t)y * rw;
    for (int x = 0; x < rw; ++x)
      if (!(mr[x] & 4) || (mr[x] & 3) == 3) vals.push_back((double)pr[x]);
  }
}

// unclip (db_utils.py:168-174) + get_mini_boxes (:154) + the rescale / clip of :158-163 for one
// ordered mini box: distance = area * ratio / perimeter (shapely, float64 on the float32 corners),
// pyclipper truncates the corners to integers, Clipper's round-join ring, its min-area rectangle.
void unclip_to_box(const Pf box[4], double unclip_ratio, int W, int H, int16_t* out8, std::vector<P>& pts,
                   std::vector<P>& h, std::vector<P>& tmp) {
  const double bx[4] = {box[0].x, box[1].x, box[2].x, box[3].x}, by[4] = {box[0].y, box[1].y, box[2].y, box[3].y};
  const double s1 = ((bx[0] * by[1] + bx[1] * by[2]) + bx[2] * by[3]) + bx[3] * by[0];
  const double s2 = ((by[0] * bx[1] + by[1] * bx[2]) + by[2] * bx[3]) + by[3] * bx[0];
  const double area = std::fabs(s1 - s2) * 0.5;
  double perim = 0;
  for (int k = 0; k < 4; ++k) perim += std::hypot(bx[(k + 1) & 3] - bx[k], by[(k + 1) & 3] - by[k]);
  const double dist = area * unclip_ratio / perim;
  long long q[4][2];
  for (int k = 0; k < 4; ++k) q[k][0] = (long long)std::trunc(bx[k]), q[k][1] = (long long)std::trunc(by[k]);
  clipper_offset_round(q, 4, dist, pts);
  hull(pts, h, tmp);
  Pf ub[4];
  double bw, bh;
  min_area_box(h, 0.0, ub, bw, bh);
  order_box(ub);
  for (int k = 0; k < 4; ++k) {
    // dest size == bitmap size (reference inference.py:158): x / W * W in float32, round half even
    float fx = nearbyintf(ub[k].x / (float)W * (float)W), fy = nearbyintf(ub[k].y / (float)H * (float)H);
    fx = std::min(std::max(fx, 0.0f), (float)W);
    fy = std::min(std::max(fy, 0.0f), (float)H);
    out8[2 * k] = (int16_t)fx;
    out8[2 * k + 1] = (int16_t)fy;
  }
}

}  // namespace

extern "C" int ctd_db_boxes(const float* prob, const int32_t* lab_f, const int32_t* st_f, int n_f,
                            const int32_t* lab_b, const int32_t* st_b, int n_b, int W, int H, int max_candidates,
                            double unclip_ratio, int16_t* boxes, float* scores, int* n_out) {
  if (!prob || !lab_f || !lab_b || !boxes || !scores || !n_out || W <= 0 || H <= 0 || max_candidates < 0 ||
      (n_f > 0 && !st_f) || (n_b > 0 && !st_b))
    return CTD_ERR_INVALID;
  std::vector<Item> items;
  items.reserve((size_t)n_f + n_b);
  for (int l = 1; l <= n_f; ++l) {
    const int32_t* s = st_f + (size_t)(l - 1) * 5;
    const int x = s[0], y = s[1], w = s[2];
    const int32_t* row = lab_f + (size_t)y * W;
    int fx = x;
    while (fx < x + w && row[fx] != l) ++fx;
    items.push_back({(long long)y * W + fx, 0, l});
  }
  for (int l = 1; l <= n_b; ++l) {
    const int32_t* s = st_b + (size_t)(l - 1) * 5;
    const int x = s[0], y = s[1], w = s[2], h = s[3];
    if (x == 0 || y == 0 || x + w == W || y + h == H) continue;   // the outer background: no hole border
    const int32_t* row = lab_b + (size_t)y * W;
    int fx = x;
    while (fx < x + w && row[fx] != l) ++fx;
    items.push_back({(long long)y * W + fx - 1, 1, l});
  }
  // cv2.findContours(RETR_LIST) hands contours back newest first
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.key > b.key; });
  if ((int)items.size() > max_candidates) items.resize(max_candidates);
  const int n = (int)items.size();
  *n_out = n;
  std::memset(boxes, 0, sizeof(int16_t) * 8 * (size_t)n);
  std::memset(scores, 0, sizeof(float) * (size_t)n);

  std::vector<P> pts, h, tmp;
  std::vector<uint8_t> m;
  std::vector<int> stack;
  std::vector<double> vals;
  for (int idx = 0; idx < n; ++idx) {
    const Item& it = items[idx];
    int x0, y0, rw, rh;
    pts.clear();
    if (it.kind == 0) {
      const int32_t* s = st_f + (size_t)(it.label - 1) * 5;
      x0 = s[0], y0 = s[1], rw = s[2], rh = s[3];
      m.assign((size_t)rw * rh, 0);
      for (int y = 0; y < rh; ++y) {
        const int32_t* row = lab_f + (size_t)(y0 + y) * W + x0;
        int lo = -1, hi = -1;
        for (int x = 0; x < rw; ++x)
          if (row[x] == it.label) {
            m[(size_t)y * rw + x] = 1;
            if (lo < 0) lo = x;
            hi = x;
          }
        if (lo >= 0) {   // hull vertices are row extremes
          pts.push_back({(double)(x0 + lo), (double)(y0 + y)});
          if (hi != lo) pts.push_back({(double)(x0 + hi), (double)(y0 + y)});
        }
      }
    } else {
      // hole border = the 4-neighbour ring of the hole
      const int32_t* s = st_b + (size_t)(it.label - 1) * 5;
      const int hx = s[0], hy = s[1], hw = s[2], hh = s[3];
      x0 = hx - 1, y0 = hy - 1, rw = hw + 2, rh = hh + 2;
      m.assign((size_t)rw * rh, 0);
      for (int y = 0; y < hh; ++y) {
        const int32_t* row = lab_b + (size_t)(hy + y) * W + hx;
        for (int x = 0; x < hw; ++x)
          if (row[x] == it.label) m[(size_t)(y + 1) * rw + x + 1] = 1;
      }
      for (int y = 0; y < rh; ++y) {
        int lo = -1, hi = -1;
        for (int x = 0; x < rw; ++x) {
          const size_t i = (size_t)y * rw + x;
          if (m[i] == 1) continue;
          const bool ring = (y > 0 && m[i - rw] == 1) || (y + 1 < rh && m[i + rw] == 1) || (x > 0 && m[i - 1] == 1) ||
                            (x + 1 < rw && m[i + 1] == 1);
          if (ring) {
            m[i] = 3;
            if (lo < 0) lo = x;
            hi = x;
          }
        }
        if (lo >= 0) {
          pts.push_back({(double)(x0 + lo), (double)(y0 + y)});
          if (hi != lo) pts.push_back({(double)(x0 + hi), (double)(y0 + y)});
        }
      }
    }
    hull(pts, h, tmp);
    Pf box[4];
    double bw, bh;
    min_area_box(h, 0.0, box, bw, bh);
    // db_utils.py:146-147 -- on the float32 the reference sees (cv2.minAreaRect returns a Size2f): a side of mathematically
    // 2.0 comes out of the calipers as 2.0 or 1.9999999999999858 by rounding noise, and the gate must not follow the noise
    if ((float)std::min(bw, bh) < 2.0f) continue;
    order_box(box);
    filled_values(m, rh, rw, it.kind == 1, prob, W, x0, y0, stack, vals);
    // ndarray.mean() of a contiguous float64 array: pairwise sums of 8192-element chunks, added
    // in order, divided by the count (checked against numpy in tests/test_post_host.py)
    double sum = 0;
    for (size_t i = 0; i < vals.size(); i += 8192) {
      const double s = pairwise(vals.data() + i, std::min<size_t>(8192, vals.size() - i));
      sum = i == 0 ? s : sum + s;
    }
    scores[idx] = (float)(sum / (double)vals.size());
    unclip_to_box(box, unclip_ratio, W, H, boxes + (size_t)idx * 8, pts, h, tmp);
  }
  return CTD_OK;
}

// The same stage from the tables `launch_dbc` (csrc/kernels_tail.hip) compacts on the device: no label
// image, no probability map.  Components are walked newest first (descending first pixel; labels are
// already in first-pixel order, so this is a merge of the two label ranges from their ends).  A
// contour's filled polygon is a subtree of the containment forest
//     component -> the holes it rings -> the components inside those holes -> ...
// and everything enclosed has a later first pixel than what encloses it, so one pass in that order has
// every subtree sum complete when its root is reached:
//     outer border of c : c + all holes ringed by c, each with everything inside it
//     hole border of h  : h's border ring (pixels of the ringing component that 4-touch h) + h + everything inside h
// Hull points are the row extremes of the component / of the ring.
extern "C" int ctd_db_boxes_compact(int W, int H, int n_f, const int32_t* st_f, const int32_t* first_f,
                                    const int32_t* par_f, const int32_t* off_f, const double* sum_f, int n_b,
                                    const int32_t* st_b, const int32_t* first_b, const int32_t* par_b,
                                    const int32_t* off_b, const double* sum_b, const double* ring_sum,
                                    const int32_t* ring_cnt, const int32_t* row_lo, const int32_t* row_hi,
                                    int max_candidates, double unclip_ratio, int16_t* boxes, float* scores, int* n_out) {
  if (W <= 0 || H <= 0 || n_f < 0 || n_b < 0 || max_candidates < 0 || !boxes || !scores || !n_out ||
      (n_f && (!st_f || !first_f || !par_f || !off_f || !sum_f)) ||
      (n_b && (!st_b || !first_b || !par_b || !off_b || !sum_b || !ring_sum || !ring_cnt)) ||
      ((n_f || n_b) && (!row_lo || !row_hi)))
    return CTD_ERR_INVALID;
  std::vector<double> acc_sf(n_f, 0.0), acc_sb(n_b, 0.0);
  std::vector<long long> acc_nf(n_f, 0), acc_nb(n_b, 0);
  std::vector<P> pts, h, tmp;
  int n = 0;
  int lf = n_f, lb = n_b;                    // next candidates: labels lf, lb (1-based), walking down
  while (lb > 0 && par_b[lb - 1] <= 0) --lb;  // only holes have a border
  while (n < max_candidates && (lf > 0 || lb > 0)) {
    const long long kf = lf > 0 ? (long long)first_f[lf - 1] : -1;
    const long long kb = lb > 0 ? (long long)first_b[lb - 1] - 1 : -1;
    const bool take_hole = kb > kf;          // ties cannot happen: the pixel above a hole's first pixel belongs to its ring
    double sum;
    long long cnt;
    pts.clear();
    if (!take_hole) {
      const int c = lf - 1;
      --lf;
      sum = sum_f[c] + acc_sf[c];
      cnt = (long long)st_f[5 * c + 4] + acc_nf[c];
      const int pb = par_f[c];
      if (pb > 0 && pb <= n_b && par_b[pb - 1] > 0) acc_sb[pb - 1] += sum, acc_nb[pb - 1] += cnt;
      const int y0 = st_f[5 * c + 1], rows = st_f[5 * c + 3];
      for (int r = 0; r < rows; ++r) {
        const int lo = row_lo[off_f[c] + r], hi = row_hi[off_f[c] + r];
        if (hi < lo) continue;
        pts.push_back({(double)lo, (double)(y0 + r)});
        if (hi != lo) pts.push_back({(double)hi, (double)(y0 + r)});
      }
    } else {
      const int c = lb - 1;
      --lb;
      while (lb > 0 && par_b[lb - 1] <= 0) --lb;
      const double inner_s = sum_b[c] + acc_sb[c];
      const long long inner_n = (long long)st_b[5 * c + 4] + acc_nb[c];
      const int pf = par_b[c];
      if (pf > 0 && pf <= n_f) acc_sf[pf - 1] += inner_s, acc_nf[pf - 1] += inner_n;
      sum = inner_s + ring_sum[c];
      cnt = inner_n + ring_cnt[c];
      const int y0 = st_b[5 * c + 1] - 1, rows = st_b[5 * c + 3] + 2;
      for (int r = 0; r < rows; ++r) {
        const int lo = row_lo[off_b[c] + r], hi = row_hi[off_b[c] + r];
        if (hi < lo) continue;
        pts.push_back({(double)lo, (double)(y0 + r)});
        if (hi != lo) pts.push_back({(double)hi, (double)(y0 + r)});
      }
    }
    const int idx = n++;
    std::memset(boxes + (size_t)idx * 8, 0, sizeof(int16_t) * 8);
    scores[idx] = 0.f;
    hull(pts, h, tmp);
    Pf box[4];
    double bw, bh;
    min_area_box(h, 0.0, box, bw, bh);
    // db_utils.py:146-147 -- on the float32 the reference sees (cv2.minAreaRect returns a Size2f): a side of mathematically
    // 2.0 comes out of the calipers as 2.0 or 1.9999999999999858 by rounding noise, and the gate must not follow the noise
    if ((float)std::min(bw, bh) < 2.0f) continue;
    order_box(box);
    scores[idx] = cnt > 0 ? (float)(sum / (double)cnt) : 0.f;
    unclip_to_box(box, unclip_ratio, W, H, boxes + (size_t)idx * 8, pts, h, tmp);
  }
  *n_out = n;
  return CTD_OK;
}
