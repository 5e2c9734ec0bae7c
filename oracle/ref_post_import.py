"""TEST INFRASTRUCTURE.  Runs the reference's OWN post-processing code
(`utils/yolov5_utils.py`, `utils/db_utils.py`, `utils/textblock.py`, `utils/textmask.py` under
/root/reference, build container only) with FUNCTIONAL stand-ins for the third-party packages that
are not installed here:

    cv2          -> the OpenCV restatements of oracle/cv_ref.py, behind cv2's call signatures
    shapely      -> Polygon.area / .length / .intersects from oracle/cv_ref.py
    pyclipper    -> PyclipperOffset: integer-truncated input, round-join offset of a quad
    torchvision  -> ops.nms from oracle/postproc_ref.py

What this pins: the oracle's line-by-line restatement of the reference's control flow
(`non_max_suppression`, `SegDetectorRepresenter`, `group_output`, `refine_mask`,
`refine_undetected_mask`) against the reference's actual code -- both run on the same primitives,
so any difference is a restatement error.  What it does NOT pin: the third-party primitives
themselves (OpenCV rounding, Clipper's integer arcs, ...), which stay restated and unpinned
(DESIGN.md section 5).  Nothing is copied from the reference; it is imported where it lies.
"""
from __future__ import annotations

import sys
import types

import numpy as np

from . import cv_ref as cv
from . import postproc_ref as R
from . import ref_import as RI


# ---------------------------------------------------------------------------------- cv2 stand-in
class _RotatedRect(tuple):
    """((cx, cy), (w, h), angle) like cv2.minAreaRect, carrying the corner points for boxPoints."""
    box: np.ndarray


def _min_area_rect(points):
    pts = np.asarray(points).reshape(-1, 2)
    box, w, h = cv.min_area_box(pts)
    c = box.astype(np.float64).mean(0)
    r = _RotatedRect(((float(c[0]), float(c[1])), (w, h), 0.0))
    r.box = box
    return r


def _structuring_element(shape, ksize, anchor=None):
    assert tuple(ksize) == (3, 3), "the reference only builds 3x3 elements"
    return (cv.CROSS3 if shape == _CV.MORPH_ELLIPSE else cv.RECT3).astype(np.uint8)


def _kernel(k):
    k = np.asarray(k)
    assert k.shape == (3, 3)
    return k.astype(bool)


def _threshold(src, thresh, maxval, typ):
    if typ & _CV.THRESH_OTSU:
        return cv.threshold_otsu(src)
    return thresh, cv.threshold_binary(src, thresh, maxval)


def _cvt_color(img, code):
    if code == _CV.COLOR_BGR2GRAY:
        return cv.cvt_bgr2gray(img)
    if code == _CV.COLOR_BGR2RGB:
        return np.ascontiguousarray(img[..., ::-1])
    raise NotImplementedError(code)


def _fill_poly(img, pts, color):
    for poly in pts:
        m = cv.fill_poly(img.shape[:2], np.asarray(poly).reshape(-1, 2))
        img[m != 0] = color
    return img


def _ccws(img, connectivity=8, ltype=None):
    n, lab, stats = R.connected_components_with_stats(img, connectivity)
    return n, lab, stats, np.zeros((n, 2))


def _find_contours(img, mode, method):
    assert mode == _CV.RETR_LIST and method == _CV.CHAIN_APPROX_SIMPLE
    return tuple(c.reshape(-1, 1, 2).astype(np.int32) for c in cv.find_contours(img)), None


def _resize(img, size, interpolation=None):
    return cv.resize_linear_u8(img, size)


def _copy_make_border(im, top, bottom, left, right, border_type, value=0):
    assert top == 0 and left == 0
    shape = (im.shape[0] + bottom, im.shape[1] + right) + im.shape[2:]
    out = np.zeros(shape, im.dtype)
    out[...] = np.asarray(value, im.dtype) if np.ndim(value) else value
    out[: im.shape[0], : im.shape[1]] = im
    return out


class _CV:
    MORPH_RECT, MORPH_ELLIPSE = 0, 2
    THRESH_BINARY, THRESH_OTSU = 0, 8
    COLOR_BGR2GRAY, COLOR_BGR2RGB = 6, 4
    RETR_LIST, CHAIN_APPROX_SIMPLE = 1, 2
    CV_16U, CV_32S = 2, 4
    INTER_LINEAR, BORDER_CONSTANT = 1, 0


def _make_cv2() -> types.ModuleType:
    m = types.ModuleType("cv2")
    for k, v in vars(_CV).items():
        if not k.startswith("_"):
            setattr(m, k, v)
    m.getStructuringElement = _structuring_element
    m.dilate = lambda img, k, iterations=1: cv.dilate(img, _kernel(k), iterations)
    m.erode = lambda img, k, iterations=1: cv.erode(img, _kernel(k), iterations)
    m.bitwise_xor = lambda a, b: np.bitwise_xor(a, b)
    m.bitwise_or = lambda a, b: np.bitwise_or(a, b)
    m.bitwise_and = lambda a, b: np.bitwise_and(a, b)
    m.threshold = _threshold
    m.cvtColor = _cvt_color
    m.inRange = lambda img, lo, hi: cv.in_range(img, lo, hi)
    m.fillPoly = _fill_poly
    m.connectedComponentsWithStats = _ccws
    m.findContours = _find_contours
    m.minAreaRect = _min_area_rect
    m.boxPoints = lambda rect: rect.box
    m.mean = lambda values, mask=None: (cv.masked_mean(values, mask), 0.0, 0.0, 0.0)
    m.resize = _resize
    m.copyMakeBorder = _copy_make_border
    m.imshow = lambda *a, **k: None
    m.setNumThreads = lambda *a, **k: None

    def _other(name):          # constants only used as default arguments of functions that are not called
        if name.startswith("__"):
            raise AttributeError(name)
        return 0
    m.__getattr__ = _other
    return m


# ------------------------------------------------------------------------------ shapely stand-in
class Polygon:
    def __init__(self, pts):
        self.pts = np.asarray(pts, np.float64).reshape(-1, 2)

    @property
    def area(self):
        return cv.polygon_area(self.pts)

    @property
    def length(self):
        return cv.polygon_length(self.pts)

    def intersects(self, other):
        return cv.polygons_intersect(self.pts, other.pts)


# ---------------------------------------------------------------------------- pyclipper stand-in
class PyclipperOffset:
    """Round-join offset of one closed convex quad the way Clipper 6.4.2 (the library pyclipper
    wraps) builds it: float corners cast to integers by truncation, integer arc points
    (`cv_ref.clipper_offset_round`).  The reference only passes the ring to minAreaRect
    (db_utils.py:153-154)."""

    def AddPath(self, path, join_type, end_type):
        self.path = np.trunc(np.asarray(path, np.float64)).astype(np.int64)

    def Execute(self, distance):
        return [cv.clipper_offset_round(self.path, float(distance)).tolist()]


def reference_detector(ns, net_outputs, input_size=(1024, 1024)):
    """The reference's `TextDetector` object without its constructor (which loads a checkpoint):
    `net` returns the given (blks, mask, lines_map) torch tensors, everything else is the
    reference's own `__call__` (inference.py:141-178)."""
    det = ns.INF.TextDetector.__new__(ns.INF.TextDetector)
    det.net = lambda img_in: net_outputs
    det.backend = "torch"
    det.input_size = input_size
    det.device = "cpu"
    det.half = False
    det.conf_thresh, det.nms_thresh = 0.4, 0.35
    det.seg_rep = ns.DB.SegDetectorRepresenter(thresh=0.3)
    return det


_LOADED = None


def load_reference_post():
    """Returns a namespace with the reference's own functions / classes."""
    global _LOADED
    if _LOADED is not None:
        return _LOADED
    if not RI.reference_available():
        raise RuntimeError("reference tree not present (build container only)")
    import torch
    RI._install_stubs()
    saved = {k: sys.modules.get(k) for k in ("cv2", "shapely", "shapely.geometry", "pyclipper", "torchvision",
                                             "torchvision.ops")}
    sys.modules["cv2"] = _make_cv2()
    sh, shg = types.ModuleType("shapely"), types.ModuleType("shapely.geometry")
    shg.Polygon = Polygon
    sh.geometry = shg
    sys.modules["shapely"], sys.modules["shapely.geometry"] = sh, shg
    pc = types.ModuleType("pyclipper")
    pc.PyclipperOffset, pc.JT_ROUND, pc.ET_CLOSEDPOLYGON = PyclipperOffset, 1, 0
    sys.modules["pyclipper"] = pc
    tv, tvo = types.ModuleType("torchvision"), types.ModuleType("torchvision.ops")
    tvo.nms = lambda boxes, scores, iou: torch.from_numpy(
        R.torchvision_nms(boxes.cpu().numpy(), scores.cpu().numpy(), float(iou)).astype(np.int64))
    tv.ops = tvo
    sys.modules["torchvision"], sys.modules["torchvision.ops"] = tv, tvo
    ours = ("utils", "models", "basemodel", "inference")
    shadow = {k: sys.modules.pop(k) for k in list(sys.modules) if k in ours or k.startswith(("utils.", "models."))}
    sys.path.insert(0, RI.REFERENCE_ROOT)
    # utils/io_utils.py:11-13 names aliases NumPy 2 removed; lend them for the import only
    lent = [k for k in ("bool8", "float_") if not hasattr(np, k)]
    for k in lent:
        setattr(np, k, {"bool8": np.bool_, "float_": np.float64}[k])
    try:
        import utils.db_utils as DB
        import utils.textblock as TB
        import utils.textmask as TM
        import utils.yolov5_utils as YU
        import inference as INF            # TextDetector.__call__, preprocess_img, postprocess_* (inference.py:72-178)
    finally:
        for k in lent:
            delattr(np, k)
        sys.path.remove(RI.REFERENCE_ROOT)
        for k in [k for k in sys.modules if k in ours or k.startswith(("utils.", "models."))]:
            del sys.modules[k]
        sys.modules.update(shadow)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    # (`get_topk_color` orders the histogram bins with np.argsort's default kind, textmask.py:17: the order of bins with EQUAL
    # counts belongs to the numpy build and the host -- x86-simd-sort on AVX-512 / AVX2.  Rounds 1-5 gave this module a stable
    # argsort so that it could be compared with a restatement pinned to the stable order; since round 6 the restatement makes the
    # reference's call literally and the product calls numpy's own function (csrc/np_dispatch.h), so the module runs unpatched.)
    _LOADED = types.SimpleNamespace(DB=DB, TB=TB, TM=TM, YU=YU, INF=INF)
    return _LOADED
