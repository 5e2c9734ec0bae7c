"""`TextBlock` (the reference's result record, utils/textblock.py:12-265) and `group_output`
(utils/textblock.py:421-508) on top of the native host code `ctd_group_output`
(csrc/host_group.cpp): this module only turns the native records into the reference's Python
objects.  N is tiny (<= 300 blocks, <= 1000 lines per page); the grouping itself -- line -> block
assignment, orientation / font size / reading distance, splitting, merging of scattered lines,
reading order, the margin of English lines -- is C++ with the reference's float64 operation order
and truncation points.
"""
from __future__ import annotations

import copy
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import _lib as L

LANG_LIST = ["eng", "ja", "unknown"]                 # textblock.py:9
LANGCLS2IDX = {"eng": 0, "ja": 1, "unknown": 2}


# Non-detection fields of the reference's result record (utils/textblock.py:24-86): the detector
# never touches them, but `to_dict()` dumps every attribute in creation order and downstream callers
# (OCR, translation, rendering) fill them.  (attribute, constructor keyword, default)
_RECORD_TAIL = (
    ("text", "text", None), ("prob", None, 1), ("translation", "translation", ""),
    ("fg_r", "fg_r", 0), ("fg_g", "fg_g", 0), ("fg_b", "fg_b", 0),
    ("bg_r", "bg_r", 0), ("bg_g", "bg_g", 0), ("bg_b", "bg_b", 0),
    ("font_family", "font_family", ""), ("bold", "bold", False), ("underline", "underline", False),
    ("italic", "italic", False), ("alpha", "alpha", 255), ("rich_text", "rich_text", ""),
    ("line_spacing", "line_spacing", 1.0), ("_alignment", "alignment", -1), ("_target_lang", "target_lang", ""),
    ("_bounding_rect", "_bounding_rect", None), ("default_stroke_width", "default_stroke_width", 0.2),
    ("accumulate_color", "accumulate_color", True),
)


class TextBlock:
    """The reference's TextBlock record (textblock.py:12-265): detection state first, then the
    fields later pipeline stages fill, in the reference's attribute order (that order is the key
    order of the JSON record, `to_dict` :158-160).  Only what the detector and the annotation writers use is here:
    the geometry / colour / alignment helpers of the reference's class (`min_rect`, `bounding_rect`, `alignment`,
    `get_font_colors`, `get_transformed_region`, ... :110-265) serve the OCR, rendering and GUI stages downstream of
    `TextDetector.__call__` and are out of scope; a caller that needs them applies the
    reference's own class to `to_dict()`'s record."""

    def __init__(self, xyxy: Sequence, lines: Optional[list] = None, language: str = "unknown",
                 vertical: bool = False, font_size: float = -1, distance=None, angle: int = 0, vec=None,
                 norm: float = -1, merged: bool = False, weight: float = -1, **kwargs):
        self.xyxy = [int(v) for v in xyxy]
        self.lines = [] if lines is None else lines
        self.vertical = vertical
        self.language = language
        self.font_size = font_size
        self.distance = None if distance is None else np.array(distance, np.float64)
        self.angle = angle
        self.vec = None if vec is None else np.array(vec, np.float64)
        self.norm = norm
        self.merged = merged
        self.weight = weight
        for attr, kw, default in _RECORD_TAIL:
            v = kwargs.get(kw, default) if kw is not None else default
            setattr(self, attr, [] if (attr == "text" and v is None) else v)

    # -- accessors the reference exposes ------------------------------------
    @property
    def pts(self) -> np.ndarray:                                 # textblock.py:146-148
        return self.lines_array()

    def lines_array(self, dtype=np.float64) -> np.ndarray:
        return np.array(self.lines, dtype=dtype)

    def __len__(self) -> int:
        return len(self.lines)

    def __getitem__(self, idx):
        return self.lines[idx]

    def xywh(self):
        x, y, w, h = self.xyxy
        return [x, y, w - x, h - y]

    def center(self):
        a = np.array(self.xyxy)
        return (a[:2] + a[2:]) / 2

    def adjust_bbox(self, with_bbox: bool = False) -> None:
        """textblock.py:87-98."""
        pts = self.lines_array().astype(np.int32).reshape(-1, 2)
        lo, hi = pts.min(0), pts.max(0)
        if with_bbox:
            lo = np.minimum(lo, self.xyxy[:2])
            hi = np.maximum(hi, self.xyxy[2:])
        self.xyxy = [lo[0], lo[1], hi[0], hi[1]]

    def sort_lines(self) -> None:
        """textblock.py:100-105."""
        if self.distance is not None:
            order = np.argsort(self.distance)
            self.distance = self.distance[order]
            self.lines = np.array(self.lines, dtype=np.int32)[order].tolist()

    def to_dict(self) -> dict:
        """`copy.deepcopy(vars(self))` (textblock.py:158-160): every attribute, numpy values included;
        `annotations.RecordEncoder` turns it into the reference's JSON."""
        return copy.deepcopy(vars(self))


# --------------------------------------------------------------------------

_TAIL_DEFAULTS = tuple((attr, default) for attr, _, default in _RECORD_TAIL)
_TAIL_DICT = {attr: default for attr, default in _TAIL_DEFAULTS}          # "text" gets a fresh list per block


def _fast_block(xyxy, lines, language, vertical, font_size, distance, angle, vec, norm, merged, weight) -> TextBlock:
    """`TextBlock(...)` without the keyword plumbing: the attribute dict in the reference's creation order."""
    t = TextBlock.__new__(TextBlock)
    d = {"xyxy": xyxy, "lines": lines, "vertical": vertical, "language": language, "font_size": font_size,
         "distance": distance, "angle": angle, "vec": vec, "norm": norm, "merged": merged, "weight": weight}
    for attr, default in _TAIL_DEFAULTS:
        d[attr] = [] if attr == "text" else default
    t.__dict__ = d
    return t


# numpy view of `ctd_blk` (include/ctd_hip.h; _lib.CtdBlk): the records of a page become columns in one call
_BLK_DT = np.dtype([("xyxy", "<i4", (4,)), ("language", "<i4"), ("vertical", "<i4"), ("angle", "<i4"),
                    ("font_is_float", "<i4"), ("font_size", "<f8"), ("vec", "<f8", (2,)), ("norm", "<f8"),
                    ("weight", "<f8"), ("merged", "<i4"), ("line_off", "<i4"), ("n_lines", "<i4"), ("dist_off", "<i4"),
                    ("n_dist", "<i4"), ("pad_", "<i4")])
assert _BLK_DT.itemsize == C.sizeof(L.CtdBlk)


def blocks_from_records(recs, lines: np.ndarray, dist: np.ndarray, n: Optional[int] = None) -> List[TextBlock]:
    """Native records (`ctd_blk` array + line / distance pools) -> the reference's Python objects.
    `TextBlock.distance` is re-evaluated from its two operands with numpy's own arccos / sin
    (reference utils/textblock.py:327-328), so the record carries the bits the reference's numpy
    expression gives on this machine (csrc/host_group.cpp decided with libm's).
    Column-wise: the per-block Python work is one dict (the interpreter lock is what the tail workers of
    `detect_stream` share, and a crowded page has 60+ blocks)."""
    if isinstance(recs, list):                       # a slice of a ctypes array is a list of struct copies
        n = len(recs)
        recs = (L.CtdBlk * max(n, 1))(*recs)
    n = len(recs) if n is None else n                # `recs`: the ctypes array the native call filled, first n entries used
    if n == 0:
        return []
    a = np.frombuffer(recs, dtype=_BLK_DT, count=n)
    if len(dist):
        with np.errstate(divide="ignore", invalid="ignore"):
            dval = np.abs(np.sin(np.arccos(np.ascontiguousarray(dist[:, 1]))) * np.ascontiguousarray(dist[:, 2]))
    else:
        dval = np.zeros((0,), np.float64)
    all_lines = lines.reshape(-1, 4, 2).tolist()
    tail, new = _TAIL_DICT, TextBlock.__new__
    out = []
    # one row per block: plain Python values from .tolist() (C loops), np.float64 / (2,) float64 arrays where the
    # reference holds numpy values (`norm`, `weight`, `vec`); `distance` is a view into this page's distance array
    for xyxy, lang, vert, ang, mg, lo, nl, do, nd, fs, isf, vec, norm, weight in zip(
            a["xyxy"].tolist(), a["language"].tolist(), a["vertical"].tolist(), a["angle"].tolist(), a["merged"].tolist(),
            a["line_off"].tolist(), a["n_lines"].tolist(), a["dist_off"].tolist(), a["n_dist"].tolist(),
            a["font_size"].tolist(), a["font_is_float"].tolist(), a["vec"].copy(), a["norm"], a["weight"]):
        d = {"xyxy": xyxy, "lines": all_lines[lo: lo + nl], "vertical": vert != 0, "language": LANG_LIST[lang],
             "font_size": fs if isf else int(fs), "distance": dval[do: do + nd], "angle": ang,
             "vec": vec, "norm": norm, "merged": mg != 0, "weight": weight}
        d.update(tail)                               # the record's non-detection fields, in the reference's order
        d["text"] = []
        t = new(TextBlock)
        t.__dict__ = d
        out.append(t)
    return out


def group_output_native(blines: np.ndarray, cls: np.ndarray, lines, im_w: int, im_h: int,
                        mask: Optional[np.ndarray] = None):
    """One `ctd_group_output` call; returns (records, lines (n,8) i32, dist (m,3) f64)."""
    lib = L.lib()
    blines = np.ascontiguousarray(np.asarray(blines, np.int32).reshape(-1, 4))
    cls = np.ascontiguousarray(np.asarray(cls, np.int32).reshape(-1))
    lines = np.asarray(lines)
    lines = np.ascontiguousarray(lines.astype(np.int32).reshape(-1, 8)) if lines.size else np.zeros((0, 8), np.int32)
    nb, nl = len(blines), len(lines)
    cap = nb + nl
    dcap = cap * max(1, nl)
    recs = (L.CtdBlk * max(cap, 1))()
    lout = np.empty((max(cap, 1), 8), np.int32)
    dout = np.empty((max(dcap, 1), 3), np.float64)
    n_b, n_l, n_d = C.c_int32(), C.c_int32(), C.c_int32()
    mptr, pitch = None, 0
    if mask is not None:
        if mask.dtype != np.uint8 or mask.ndim != 2 or mask.shape != (im_h, im_w):
            raise ValueError("mask must be a uint8 (im_h, im_w) array")
        mask = np.ascontiguousarray(mask)
        mptr, pitch = mask.ctypes.data, mask.shape[1]
    L.check(lib.ctd_group_output(blines.ctypes.data, cls.ctypes.data, nb, lines.ctypes.data, nl, int(im_w), int(im_h),
                                 mptr, pitch, recs, cap, lout.ctypes.data, cap, dout.ctypes.data, dcap,
                                 C.byref(n_b), C.byref(n_l), C.byref(n_d)), "ctd_group_output")
    return recs[: n_b.value], lout[: n_l.value], dout[: n_d.value]


def group_output(blks, lines, im_w: int, im_h: int, mask: Optional[np.ndarray] = None,
                 sort_blklist: bool = True) -> List[TextBlock]:
    """textblock.py:421-508.  blks = (blines (n,4) i32, cls (n,) i32, confs (n,)); lines (m,4,2) i32."""
    if not sort_blklist:
        raise NotImplementedError("the detector always sorts the block list (reference inference.py:173)")
    recs, lout, dout = group_output_native(blks[0], blks[1], lines, im_w, im_h, mask)
    return blocks_from_records(recs, lout, dout)
