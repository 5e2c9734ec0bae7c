"""Host-side geometry used by the post-processing mirror (O(#components) scalar
work; SURVEY 2.1: "small host code for O(#contours) geometry").  Replaces the
reference's calls into OpenCV / pyclipper / shapely:

  cv2.minAreaRect + boxPoints   utils/db_utils.py:177-178   -> min_area_box
  pyclipper offset + minAreaRect utils/db_utils.py:171-173,154 -> min_area_box(grow=d)
  shapely area / length          utils/db_utils.py:169-170   -> quad_area / quad_perimeter
  shapely intersects             utils/textblock.py:355,400  -> quads_intersect
"""
from __future__ import annotations

import numpy as np


def hull(points: np.ndarray) -> np.ndarray:
    """Convex hull (counter-clockwise in x-right/y-down image axes = clockwise on screen),
    float64, no repeated vertex.  Vectorised pre-filter + monotone chain."""
    p = np.asarray(points, np.float64).reshape(-1, 2)
    if len(p) > 64:
        # keep only per-row extremes: every hull vertex is the min-x or max-x point of its row
        order = np.lexsort((p[:, 0], p[:, 1]))
        p = p[order]
        first = np.r_[True, p[1:, 1] != p[:-1, 1]]
        last = np.r_[first[1:], True]
        p = p[first | last]
    p = np.unique(p, axis=0)
    if len(p) <= 2:
        return p
    p = p[np.lexsort((p[:, 1], p[:, 0]))]

    def half(pts):
        out = []
        for q in pts:
            while len(out) >= 2:
                a, b = out[-2], out[-1]
                if (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0]) <= 0:
                    out.pop()
                else:
                    break
            out.append(q)
        return out

    lo = half(p)
    up = half(p[::-1])
    return np.array(lo[:-1] + up[:-1], np.float64)


def min_area_box(points: np.ndarray, grow: float = 0.0):
    """Corners (4,2) float32 of the minimum-area rectangle enclosing `points`, optionally
    grown by `grow` on every side, plus its two side lengths.  All hull edges are evaluated
    at once (rotating-calipers result: the optimum shares a side with the hull)."""
    h = hull(points)
    n = len(h)
    if n == 0:
        return np.zeros((4, 2), np.float32), 0.0, 0.0
    if n == 1:
        c, g = h[0], grow
        b = np.array([[c[0] - g, c[1] - g], [c[0] + g, c[1] - g], [c[0] + g, c[1] + g], [c[0] - g, c[1] + g]])
        return b.astype(np.float32), 2 * g, 2 * g
    e = np.roll(h, -1, axis=0) - h
    if n == 2:
        e = e[:1]
    L = np.hypot(e[:, 0], e[:, 1])
    keep = L > 0
    e, L = e[keep], L[keep]
    u = e / L[:, None]                       # (m,2)
    v = np.stack([-u[:, 1], u[:, 0]], 1)
    pu = h @ u.T                             # (n,m)
    pv = h @ v.T
    lo_u, hi_u = pu.min(0) - grow, pu.max(0) + grow
    lo_v, hi_v = pv.min(0) - grow, pv.max(0) + grow
    area = (hi_u - lo_u) * (hi_v - lo_v)
    # first minimum in hull-edge order, with the same tie tolerance as a sequential scan
    k = 0
    for i in range(1, len(area)):
        if area[i] < area[k] - 1e-12:
            k = i
    uu, vv = u[k], v[k]
    box = np.array([uu * lo_u[k] + vv * lo_v[k], uu * hi_u[k] + vv * lo_v[k],
                    uu * hi_u[k] + vv * hi_v[k], uu * lo_u[k] + vv * hi_v[k]])
    return box.astype(np.float32), float(hi_u[k] - lo_u[k]), float(hi_v[k] - lo_v[k])


def order_box(box: np.ndarray):
    """`get_mini_boxes` ordering (utils/db_utils.py:178-194): sort by x, then TL, TR, BR, BL."""
    pts = sorted([p for p in box], key=lambda p: p[0])
    i1, i4 = (0, 1) if pts[1][1] > pts[0][1] else (1, 0)
    i2, i3 = (2, 3) if pts[3][1] > pts[2][1] else (3, 2)
    return np.array([pts[i1], pts[i2], pts[i3], pts[i4]])


def quad_area(p: np.ndarray) -> float:
    p = np.asarray(p, np.float64).reshape(-1, 2)
    return abs(float(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1)))) * 0.5


def quad_perimeter(p: np.ndarray) -> float:
    p = np.asarray(p, np.float64).reshape(-1, 2)
    d = np.roll(p, -1, axis=0) - p
    return float(np.hypot(d[:, 0], d[:, 1]).sum())


def _orient(ax, ay, bx, by, cx, cy):
    v = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    return (v > 0) - (v < 0)


def _on_segment(ax, ay, bx, by, cx, cy):
    return min(ax, bx) <= cx <= max(ax, bx) and min(ay, by) <= cy <= max(ay, by)


def _inside(pt, poly):
    x, y = pt
    c = False
    n = len(poly)
    for k in range(n):
        x0, y0 = poly[k]
        x1, y1 = poly[(k + 1) % n]
        if (y0 > y) != (y1 > y) and x < x0 + (y - y0) * (x1 - x0) / (y1 - y0):
            c = not c
    return c


def quads_intersect(p, q) -> bool:
    """Polygon(p).intersects(Polygon(q)): boundaries cross/touch, or containment."""
    p = [(float(a), float(b)) for a, b in np.asarray(p).reshape(-1, 2)]
    q = [(float(a), float(b)) for a, b in np.asarray(q).reshape(-1, 2)]
    # quick reject on bounding boxes
    if max(a for a, _ in p) < min(a for a, _ in q) or max(a for a, _ in q) < min(a for a, _ in p) or \
       max(b for _, b in p) < min(b for _, b in q) or max(b for _, b in q) < min(b for _, b in p):
        return False
    for i in range(len(p)):
        a, b = p[i], p[(i + 1) % len(p)]
        for j in range(len(q)):
            c, d = q[j], q[(j + 1) % len(q)]
            o1, o2 = _orient(*a, *b, *c), _orient(*a, *b, *d)
            o3, o4 = _orient(*c, *d, *a), _orient(*c, *d, *b)
            if o1 != o2 and o3 != o4:
                return True
            if (o1 == 0 and _on_segment(*a, *b, *c)) or (o2 == 0 and _on_segment(*a, *b, *d)) or \
               (o3 == 0 and _on_segment(*c, *d, *a)) or (o4 == 0 and _on_segment(*c, *d, *b)):
                return True
    return _inside(p[0], q) or _inside(q[0], p)
