"""TEST INFRASTRUCTURE.  numpy restatements of the third-party image/geometry
primitives the reference's post-processing calls (OpenCV `>=4.1.2`
requirements.txt:3, pyclipper and shapely unpinned).  None of
those wheels is installed in the build container and the reference ships no
tests, so every function here follows the PUBLISHED algorithm of the library
routine it stands in for and is marked with what could not be verified:

    PARITY UNPINNED at this boundary.

Each function names the reference call site(s) it serves.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np


# --------------------------------------------------------------------------
# colour / threshold / morphology   (reference utils/textmask.py:33-37,47,58-68,87-89,111)
# --------------------------------------------------------------------------

def cvt_bgr2gray(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY) for uint8: fixed point, 15 fractional bits
    (OpenCV 4.x imgproc/src/color_rgb.simd.hpp `RGB2Gray<uchar>`: BY15 = 3735, GY15 = 19235,
    RY15 = 9798, + (1 << 14) >> 15).  VERSION DEPENDENT: OpenCV 3.x and the first 4.x releases used
    the 14-bit set (1868, 9617, 4899, >> 14); the two differ by one grey level on some coloured
    pixels and agree on every grey pixel (B = G = R).  The reference asks for opencv-python>=4.1.2
    (requirements.txt:3) without an upper bound; this restatement follows current 4.x."""
    b, g, r = img[..., 0].astype(np.int64), img[..., 1].astype(np.int64), img[..., 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def threshold_binary(src: np.ndarray, thresh: float, maxval: int = 255) -> np.ndarray:
    """cv2.threshold(src, thresh, maxval, THRESH_BINARY): dst = maxval if src > thresh else 0."""
    return np.where(src.astype(np.float64) > thresh, maxval, 0).astype(np.uint8)


def otsu_threshold_value(src: np.ndarray) -> int:
    """OpenCV getThreshVal_Otsu_8u (imgproc/thresh.cpp): between-class variance in double."""
    h = np.bincount(src.ravel(), minlength=256).astype(np.float64)
    n = float(src.size)
    scale = 1.0 / n
    mu = float(np.dot(np.arange(256, dtype=np.float64), h)) * scale
    mu1, q1 = 0.0, 0.0
    max_sigma, max_val = 0.0, 0
    eps = float(np.finfo(np.float32).eps)
    for i in range(256):
        p_i = h[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, max_val = sigma, i
    return max_val


def threshold_otsu(src: np.ndarray) -> Tuple[int, np.ndarray]:
    """cv2.threshold(c, 1, 255, THRESH_OTSU + THRESH_BINARY) (reference utils/textmask.py:47)."""
    t = otsu_threshold_value(src)
    return t, threshold_binary(src, t, 255)


CROSS3 = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)   # getStructuringElement(MORPH_ELLIPSE, (3,3))
RECT3 = np.ones((3, 3), bool)                                 # np.ones((3,3)) / MORPH_RECT


def _morph(img: np.ndarray, kernel: np.ndarray, op) -> np.ndarray:
    """3x3 erode/dilate; pixels outside the image do not take part
    (cv2 morphologyDefaultBorderValue: +inf for erode, -inf for dilate)."""
    h, w = img.shape
    fill = 255 if op is np.minimum else 0
    pad = np.full((h + 2, w + 2), fill, img.dtype)
    pad[1:-1, 1:-1] = img
    out = np.full((h, w), fill, img.dtype)
    for dy in range(3):
        for dx in range(3):
            if kernel[dy, dx]:
                out = op(out, pad[dy:dy + h, dx:dx + w])
    return out


def erode(img: np.ndarray, kernel: np.ndarray = RECT3, iterations: int = 1) -> np.ndarray:
    for _ in range(iterations):
        img = _morph(img, kernel, np.minimum)
    return img


def dilate(img: np.ndarray, kernel: np.ndarray = RECT3, iterations: int = 1) -> np.ndarray:
    for _ in range(iterations):
        img = _morph(img, kernel, np.maximum)
    return img


def in_range_bounds(lo: float, hi: float):
    """Integer bounds cv2.inRange derives from double scalars for a CV_8U image (OpenCV
    core/src/arithm.cpp `inRange`, scalar branch): both bounds are converted to int32 with
    cvRound (round half to even); if lb > ub, lb > 255 or ub < 0 nothing matches; otherwise the
    bounds saturate to [0, 255].  Returns (lb, ub) with lb > ub meaning the empty range."""
    ilo, ihi = int(np.rint(float(lo))), int(np.rint(float(hi)))
    if ilo > ihi or ilo > 255 or ihi < 0:
        return 1, 0
    return max(ilo, 0), min(ihi, 255)


def in_range(img: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """cv2.inRange(img, lo, hi) with scalar (possibly fractional) bounds on uint8."""
    lb, ub = in_range_bounds(lo, hi)
    v = img.astype(np.int64)
    return np.where((v >= lb) & (v <= ub), 255, 0).astype(np.uint8)


def _linear_coeffs(dst: int, src: int, zero_frac_at_edges: bool):
    """OpenCV resize INTER_LINEAR coefficient tables (imgproc/resize.cpp):
    f = (float)((d + 0.5) * scale - 0.5); s = floor(f); f -= s; the pair (1-f, f) is stored as
    shorts saturate_cast<short>(c * 2048) (cvRound = round half to even).  For x the fraction is
    zeroed outside [0, src-1); for y the two rows are clamped instead."""
    scale = 1.0 / (dst / src)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    f = (f - s0.astype(np.float32)).astype(np.float32)
    if zero_frac_at_edges:
        lo = s0 < 0
        f = np.where(lo, np.float32(0), f)
        s0 = np.where(lo, 0, s0)
        hi = s0 >= src - 1
        f = np.where(hi, np.float32(0), f)
        s0 = np.where(hi, src - 1, s0)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    i0 = np.clip(s0, 0, src - 1)
    i1 = np.clip(s0 + 1, 0, src - 1)
    return i0, i1, c0, c1


def resize_linear_u8(img: np.ndarray, size_wh: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (w,h), interpolation=INTER_LINEAR) for uint8, 1 or 3 channels
    (reference utils/imgproc_utils.py:113, inference.py:165): OpenCV's fixed-point path
    (HResizeLinear -> int = S0*a0 + S1*a1, VResizeLinear ->
    (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2).
    UNPINNED: opencv-python wheels built with IPP may round differently by +-1."""
    w, h = size_wh
    sh, sw = img.shape[:2]
    x0, x1, a0, a1 = _linear_coeffs(w, sw, True)
    y0, y1, b0, b1 = _linear_coeffs(h, sh, False)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[..., None]
    hor = src[:, x0] * a0[None, :, None] + src[:, x1] * a1[None, :, None]        # (sh, w, C) ints
    r0, r1 = hor[y0], hor[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out


def letterbox(im: np.ndarray, new_shape=(1024, 1024)):
    """reference utils/imgproc_utils.py:86-117 with auto=False, scaleFill=False, scaleup=True:
    aspect-keeping resize, then pad bottom/right with 0.  Returns (image, (rw, rh), (dw, dh))."""
    shape = im.shape[:2]
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    dh, dw = int(dh), int(dw)
    if shape[::-1] != new_unpad:
        im = resize_linear_u8(im, new_unpad)
    out = np.zeros((im.shape[0] + dh, im.shape[1] + dw) + im.shape[2:], np.uint8)
    out[: im.shape[0], : im.shape[1]] = im
    return out, (r, r), (dw, dh)


# --------------------------------------------------------------------------
# contours  (reference utils/db_utils.py:136 findContours RETR_LIST, CHAIN_APPROX_SIMPLE)
# --------------------------------------------------------------------------

# 8-neighbourhood in clockwise order starting east (x right, y down)
_DIRS = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]   # (dy, dx)


def find_contours(binary: np.ndarray) -> List[np.ndarray]:
    """Suzuki-Abe border following (Suzuki & Abe 1985, the algorithm cv2.findContours
    implements) with RETR_LIST (outer and hole borders, no hierarchy) and
    CHAIN_APPROX_SIMPLE.  Returns contours as (n,2) int32 arrays of (x,y).
    Order: OpenCV links every new contour in FRONT of the list, so the contour found
    last by the raster scan comes first.  (Order matters only above max_candidates.)"""
    h, w = binary.shape
    f = np.zeros((h + 2, w + 2), np.int32)
    f[1:-1, 1:-1] = (binary != 0).astype(np.int32)
    nbd = 1
    contours: List[np.ndarray] = []
    for i in range(1, h + 1):
        row = f[i]
        for j in range(1, w + 1):
            v = row[j]
            if v == 0:
                continue
            if v == 1 and row[j - 1] == 0:
                start_dir = 4          # came from the west neighbour (i, j-1)
            elif v >= 1 and row[j + 1] == 0:
                start_dir = 0          # hole border: from the east neighbour (i, j+1)
            else:
                continue
            nbd += 1
            pts = _follow_border(f, i, j, start_dir, nbd)
            contours.append(_approx_simple(pts))
    contours.reverse()
    return contours


def _follow_border(f: np.ndarray, i: int, j: int, start_dir: int, nbd: int) -> List[Tuple[int, int]]:
    # step 3.1: clockwise from (i2,j2) find the first non-zero neighbour
    d = start_dir
    found = -1
    for k in range(8):
        dd = (d + k) % 8
        if f[i + _DIRS[dd][0], j + _DIRS[dd][1]] != 0:
            found = dd
            break
    if found < 0:
        f[i, j] = -nbd
        return [(j - 1, i - 1)]
    i1, j1 = i + _DIRS[found][0], j + _DIRS[found][1]
    i2, j2 = i1, j1
    i3, j3 = i, j
    pts = []
    while True:
        # step 3.3: counter-clockwise from the element after (i2,j2) find the first non-zero pixel
        d2 = _DIRS.index((i2 - i3, j2 - j3))
        east_zero_seen = False
        dd = d2
        for k in range(1, 9):
            dd = (d2 - k) % 8
            ni, nj = i3 + _DIRS[dd][0], j3 + _DIRS[dd][1]
            if f[ni, nj] != 0:
                break
            if dd == 0:
                east_zero_seen = True     # the pixel (i3, j3+1) was examined and is 0
        i4, j4 = i3 + _DIRS[dd][0], j3 + _DIRS[dd][1]
        # step 3.4
        if east_zero_seen:
            f[i3, j3] = -nbd
        elif f[i3, j3] == 1:
            f[i3, j3] = nbd
        pts.append((j3 - 1, i3 - 1))
        # step 3.5
        if (i4, j4) == (i, j) and (i3, j3) == (i1, j1):
            break
        i2, j2 = i3, j3
        i3, j3 = i4, j4
    return pts


def _approx_simple(pts: List[Tuple[int, int]]) -> np.ndarray:
    """CHAIN_APPROX_SIMPLE: drop points in the middle of straight (h/v/diagonal) runs."""
    n = len(pts)
    if n <= 2:
        return np.array(pts, np.int32).reshape(-1, 2)
    out = []
    for k in range(n):
        p0, p1, p2 = pts[k - 1], pts[k], pts[(k + 1) % n]
        if (p1[0] - p0[0], p1[1] - p0[1]) != (p2[0] - p1[0], p2[1] - p1[1]):
            out.append(p1)
    if not out:
        out = [pts[0]]
    return np.array(out, np.int32).reshape(-1, 2)


# --------------------------------------------------------------------------
# min-area rectangle  (reference utils/db_utils.py:177-178 minAreaRect + boxPoints)
# --------------------------------------------------------------------------

def convex_hull(points: np.ndarray) -> np.ndarray:
    """Andrew monotone chain, returns hull vertices (float64) without repetition."""
    pts = np.unique(np.asarray(points, np.float64).reshape(-1, 2), axis=0)
    if len(pts) <= 2:
        return pts
    pts = pts[np.lexsort((pts[:, 1], pts[:, 0]))]

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in pts[::-1]:
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return np.array(lower[:-1] + upper[:-1], np.float64)


def min_area_box(points: np.ndarray, grow: float = 0.0) -> Tuple[np.ndarray, float, float]:
    """Minimum-area enclosing rectangle of a point set (rotating calipers: one side is
    collinear with a hull edge) -- the 4 corners cv2.boxPoints(cv2.minAreaRect(pts)) yields,
    as float32 (4,2), plus (width, height) of the rectangle.  `grow` enlarges the rectangle by
    that distance on every side (used by `unclip`: the min-area rectangle of a polygon offset
    with round joins is the polygon's calipers rectangle grown by the offset)."""
    hull = convex_hull(points)
    n = len(hull)
    if n == 0:
        return np.zeros((4, 2), np.float32), 0.0, 0.0
    if n == 1:
        c = hull[0]
        g = grow
        box = np.array([[c[0] - g, c[1] - g], [c[0] + g, c[1] - g], [c[0] + g, c[1] + g], [c[0] - g, c[1] + g]])
        return box.astype(np.float32), 2 * g, 2 * g
    best = None
    m = n if n > 2 else 1
    for k in range(m):
        e = hull[(k + 1) % n] - hull[k]
        L = math.hypot(e[0], e[1])
        if L == 0:
            continue
        u = e / L
        v = np.array([-u[1], u[0]])
        pu = hull @ u
        pv = hull @ v
        lo_u, hi_u, lo_v, hi_v = pu.min() - grow, pu.max() + grow, pv.min() - grow, pv.max() + grow
        area = (hi_u - lo_u) * (hi_v - lo_v)
        # first minimum in edge order, with a RELATIVE margin: two hull edges of an integer-cornered ring often bound
        # rectangles of mathematically equal area (seed 59 of the tail sweep: 2778 twice), and an absolute 1e-12 let the
        # rounding noise of the projections (7e-12 there; numpy's dot against the product's mul + add) pick between them
        if best is None or area < best[0] * (1.0 - 1e-9):
            best = (area, u, v, lo_u, hi_u, lo_v, hi_v)
    _, u, v, lo_u, hi_u, lo_v, hi_v = best
    box = np.array([u * lo_u + v * lo_v, u * hi_u + v * lo_v, u * hi_u + v * hi_v, u * lo_u + v * hi_v])
    return box.astype(np.float32), float(hi_u - lo_u), float(hi_v - lo_v)


# --------------------------------------------------------------------------
# polygon fill + masked mean  (reference utils/db_utils.py:197-211 box_score_fast)
# --------------------------------------------------------------------------

def _line_pixels(x0: int, y0: int, x1: int, y1: int):
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x1 >= x0 else -1), (1 if y1 >= y0 else -1)
    err = dx - dy
    x, y = x0, y0
    while True:
        yield x, y
        if x == x1 and y == y1:
            return
        e2 = 2 * err
        if e2 > -dy:
            err -= dy
            x += sx
        if e2 < dx:
            err += dx
            y += sy


def fill_poly(shape_hw: Tuple[int, int], poly: np.ndarray) -> np.ndarray:
    """cv2.fillPoly(mask, [poly], 1) for one integer polygon: even-odd interior at pixel
    centres plus every pixel on the outline (OpenCV's FillEdgeCollection also draws the edges)."""
    h, w = shape_hw
    mask = np.zeros((h, w), np.uint8)
    p = np.asarray(poly, np.int64).reshape(-1, 2)
    n = len(p)
    if n == 0:
        return mask
    ys = np.arange(h)
    xs = np.arange(w)
    inside = np.zeros((h, w), bool)
    for k in range(n):
        x0, y0 = p[k]
        x1, y1 = p[(k + 1) % n]
        if y0 == y1:
            continue
        if y0 > y1:
            x0, y0, x1, y1 = x1, y1, x0, y0
        rows = ys[(ys >= y0) & (ys < y1)]
        if len(rows) == 0:
            continue
        xc = x0 + (rows - y0) * (x1 - x0) / (y1 - y0)
        inside[rows] ^= xs[None, :] < xc[:, None]          # toggles pixels left of the crossing
    mask[inside] = 1
    for k in range(n):
        for x, y in _line_pixels(int(p[k][0]), int(p[k][1]), int(p[(k + 1) % n][0]), int(p[(k + 1) % n][1])):
            if 0 <= x < w and 0 <= y < h:
                mask[y, x] = 1
    return mask


def masked_mean(values: np.ndarray, mask: np.ndarray) -> float:
    """cv2.mean(values, mask)[0]: double accumulation over mask != 0; 0 for an empty mask."""
    m = mask != 0
    c = int(m.sum())
    return float(values[m].astype(np.float64).sum() / c) if c else 0.0


# --------------------------------------------------------------------------
# Clipper polygon offset  (reference utils/db_utils.py:171-173: PyclipperOffset, JT_ROUND)
# --------------------------------------------------------------------------

def _clipper_round(v: float) -> int:
    """Clipper 6.4.2 `Round`: half away from zero through a C cast."""
    return int(v - 0.5) if v < 0 else int(v + 0.5)


def clipper_offset_round(path, delta: float, arc_tolerance: float = 0.25) -> np.ndarray:
    """`pyclipper.PyclipperOffset().AddPath(path, JT_ROUND, ET_CLOSEDPOLYGON); Execute(delta)` for ONE
    convex closed polygon and delta > 0, following Clipper 6.4.2 (the library pyclipper wraps):
    `ClipperOffset::AddPath` (duplicate stripping), `FixOrientations`, `DoOffset` (step count from the
    arc tolerance, per-vertex `OffsetPoint` -> `DoRound`, every emitted point rounded to the integer
    grid).  The final `Clipper::Execute(ctUnion, pftPositive)` only removes self-overlaps; for a
    convex input the raw offset ring is already simple, so the ring itself is returned (the union may
    rotate the start vertex or drop collinear points -- irrelevant to the min-area rectangle the
    reference takes next, db_utils.py:154).  `path`: integer (x, y) vertices (pyclipper casts float
    input to integers by truncation).  Returns (n, 2) int64."""
    pts = [(int(x), int(y)) for x, y in np.asarray(path).reshape(-1, 2)]
    hi = len(pts) - 1
    while hi > 0 and pts[0] == pts[hi]:
        hi -= 1
    poly = [pts[0]]
    for i in range(1, hi + 1):
        if pts[i] != poly[-1]:
            poly.append(pts[i])
    n = len(poly)
    if n < 3:
        return np.zeros((0, 2), np.int64)
    a = 0.0
    j = n - 1
    for i in range(n):
        a += (float(poly[j][0]) + poly[i][0]) * (float(poly[j][1]) - poly[i][1])
        j = i
    if -a * 0.5 < 0:                       # FixOrientations: Orientation() == Area() >= 0
        poly.reverse()
    if arc_tolerance <= 0.0:
        y = 0.25
    elif arc_tolerance > abs(delta) * 0.25:
        y = abs(delta) * 0.25
    else:
        y = arc_tolerance
    steps = math.pi / math.acos(1 - y / abs(delta))
    if steps > abs(delta) * math.pi:
        steps = abs(delta) * math.pi
    m_sin, m_cos = math.sin(2 * math.pi / steps), math.cos(2 * math.pi / steps)
    steps_per_rad = steps / (2 * math.pi)
    if delta < 0:
        m_sin = -m_sin

    def unit_normal(p1, p2):
        if p1 == p2:
            return (0.0, 0.0)
        dx, dy = float(p2[0] - p1[0]), float(p2[1] - p1[1])
        f = 1.0 / math.sqrt(dx * dx + dy * dy)
        dx *= f
        dy *= f
        return (dy, -dx)

    normals = [unit_normal(poly[i], poly[(i + 1) % n]) for i in range(n)]
    out = []
    k = n - 1
    for j in range(n):
        sin_a = normals[k][0] * normals[j][1] - normals[j][0] * normals[k][1]
        done = False
        if abs(sin_a * delta) < 1.0:
            cos_a = normals[k][0] * normals[j][0] + normals[j][1] * normals[k][1]
            if cos_a > 0:
                out.append((_clipper_round(poly[j][0] + normals[k][0] * delta),
                            _clipper_round(poly[j][1] + normals[k][1] * delta)))
                done = True
        elif sin_a > 1.0:
            sin_a = 1.0
        elif sin_a < -1.0:
            sin_a = -1.0
        if not done:
            if sin_a * delta < 0:
                out.append((_clipper_round(poly[j][0] + normals[k][0] * delta),
                            _clipper_round(poly[j][1] + normals[k][1] * delta)))
                out.append(poly[j])
                out.append((_clipper_round(poly[j][0] + normals[j][0] * delta),
                            _clipper_round(poly[j][1] + normals[j][1] * delta)))
            else:                           # DoRound
                ang = math.atan2(sin_a, normals[k][0] * normals[j][0] + normals[k][1] * normals[j][1])
                nst = max(_clipper_round(steps_per_rad * abs(ang)), 1)
                X, Y = normals[k]
                for _ in range(nst):
                    out.append((_clipper_round(poly[j][0] + X * delta), _clipper_round(poly[j][1] + Y * delta)))
                    X2 = X
                    X = X * m_cos - m_sin * Y
                    Y = X2 * m_sin + Y * m_cos
                out.append((_clipper_round(poly[j][0] + normals[j][0] * delta),
                            _clipper_round(poly[j][1] + normals[j][1] * delta)))
        k = j
    return np.array(out, np.int64).reshape(-1, 2)


# --------------------------------------------------------------------------
# shapely stand-ins (reference utils/db_utils.py:169-170, utils/textblock.py:355-356,400-402)
# --------------------------------------------------------------------------

def polygon_area(pts: np.ndarray) -> float:
    """shapely Polygon(pts).area (absolute shoelace)."""
    p = np.asarray(pts, np.float64).reshape(-1, 2)
    return abs(float(np.dot(p[:, 0], np.roll(p[:, 1], -1)) - np.dot(p[:, 1], np.roll(p[:, 0], -1)))) * 0.5


def polygon_length(pts: np.ndarray) -> float:
    """shapely Polygon(pts).length (closed perimeter)."""
    p = np.asarray(pts, np.float64).reshape(-1, 2)
    d = np.roll(p, -1, axis=0) - p
    return float(np.hypot(d[:, 0], d[:, 1]).sum())


def _seg_intersect(a, b, c, d) -> bool:
    def orient(p, q, r):
        v = (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
        return 0 if v == 0 else (1 if v > 0 else -1)

    def on_seg(p, q, r):
        return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])

    o1, o2, o3, o4 = orient(a, b, c), orient(a, b, d), orient(c, d, a), orient(c, d, b)
    if o1 != o2 and o3 != o4:
        return True
    return (o1 == 0 and on_seg(a, b, c)) or (o2 == 0 and on_seg(a, b, d)) or \
           (o3 == 0 and on_seg(c, d, a)) or (o4 == 0 and on_seg(c, d, b))


def _point_in_poly(pt, poly) -> bool:
    x, y = pt
    inside = False
    n = len(poly)
    for k in range(n):
        x0, y0 = poly[k]
        x1, y1 = poly[(k + 1) % n]
        if (y0 > y) != (y1 > y):
            if x < x0 + (y - y0) * (x1 - x0) / (y1 - y0):
                inside = not inside
    return inside


def polygons_intersect(p: np.ndarray, q: np.ndarray) -> bool:
    """shapely Polygon(p).intersects(Polygon(q)) for simple polygons: true when the
    boundaries cross or touch, or one polygon contains the other."""
    p = [tuple(map(float, v)) for v in np.asarray(p).reshape(-1, 2)]
    q = [tuple(map(float, v)) for v in np.asarray(q).reshape(-1, 2)]
    for i in range(len(p)):
        for j in range(len(q)):
            if _seg_intersect(p[i], p[(i + 1) % len(p)], q[j], q[(j + 1) % len(q)]):
                return True
    return _point_in_poly(p[0], q) or _point_in_poly(q[0], p)
