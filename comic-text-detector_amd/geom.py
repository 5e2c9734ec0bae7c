"""Host-side polygon predicate used by the text-block grouping (O(#lines^2) scalar work on a
few dozen quads).  Replaces the reference's call into shapely:

  shapely Polygon.intersects   utils/textblock.py:355,400  -> quads_intersect

The contour geometry of the DB stage (hull, calipers rectangle, unclip) lives in the native
host code behind `ctd_db_boxes` (csrc/host_db.cpp).
"""
from __future__ import annotations

import numpy as np


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
