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


def rotate_polygons(center, polygons: np.ndarray, rotation, new_center=None, to_int: bool = True) -> np.ndarray:
    """reference utils/imgproc_utils.py:68-84 (float32 arithmetic like the reference)."""
    if new_center is None:
        new_center = center
    rotation = np.deg2rad(rotation)
    s, c = np.sin(rotation), np.cos(rotation)
    polygons = polygons.astype(np.float32)
    polygons[:, 1::2] -= center[1]
    polygons[:, ::2] -= center[0]
    rotated = np.copy(polygons)
    rotated[:, 1::2] = polygons[:, 1::2] * c - polygons[:, ::2] * s
    rotated[:, ::2] = polygons[:, 1::2] * s + polygons[:, ::2] * c
    rotated[:, 1::2] += new_center[1]
    rotated[:, ::2] += new_center[0]
    return rotated.astype(np.int64) if to_int else rotated


class TextBlock:
    """The reference's TextBlock record (textblock.py:12-265): detection state first, then the
    fields later pipeline stages fill, in the reference's attribute order (that order is the key
    order of the JSON record, `to_dict` :158-160), with the reference's numpy-only methods.
    `get_transformed_region` (:162-196, cv2.findHomography / warpPerspective) and
    `visualize_textblocks` (:510-523, cv2 drawing) belong to the OCR / debugging side and are not
    part of this package."""

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

    def aspect_ratio(self) -> float:
        """textblock.py:110-115."""
        min_rect = self.min_rect()
        mid = (min_rect[:, [1, 2, 3, 0]] + min_rect) / 2
        return np.linalg.norm(mid[:, 2] - mid[:, 0]) / np.linalg.norm(mid[:, 1] - mid[:, 3])

    def min_rect(self, rotate_back: bool = True) -> np.ndarray:
        """textblock.py:121-134: bounding rectangle of the lines in the block's rotated frame."""
        angled = self.angle != 0
        center = self.center()
        polygons = self.lines_array().reshape(-1, 8)
        if angled:
            polygons = rotate_polygons(center, polygons, self.angle)
        min_x, min_y = polygons[:, ::2].min(), polygons[:, 1::2].min()
        max_x, max_y = polygons[:, ::2].max(), polygons[:, 1::2].max()
        min_bbox = np.array([[min_x, min_y, max_x, min_y, max_x, max_y, min_x, max_y]])
        if angled and rotate_back:
            min_bbox = rotate_polygons(center, min_bbox, -self.angle)
        return min_bbox.reshape(-1, 4, 2).astype(np.int64)

    def bounding_rect(self):
        """textblock.py:136-144: Qt-style [x, y, w, h], ignoring the angle."""
        if self._bounding_rect is None:
            min_bbox = self.min_rect(rotate_back=False)[0]
            x, y = min_bbox[0]
            w, h = min_bbox[2] - min_bbox[0]
            return [x, y, w, h]
        return self._bounding_rect

    def get_text(self) -> str:
        """textblock.py:198-201."""
        if isinstance(self.text, str):
            return self.text
        return " ".join(self.text).strip()

    def set_font_colors(self, frgb, srgb, accumulate: bool = True) -> None:
        """textblock.py:203-211."""
        self.accumulate_color = accumulate
        num_lines = len(self.lines) if accumulate and len(self.lines) > 0 else 1
        self.fg_r, self.fg_g, self.fg_b = np.array(frgb) * num_lines
        self.bg_r, self.bg_g, self.bg_b = np.array(srgb) * num_lines

    def get_font_colors(self, bgr: bool = False):
        """textblock.py:213-228."""
        num_lines = len(self.lines)
        frgb = np.array([self.fg_r, self.fg_g, self.fg_b])
        brgb = np.array([self.bg_r, self.bg_g, self.bg_b])
        if self.accumulate_color:
            if num_lines > 0:
                frgb = (frgb / num_lines).astype(np.int32)
                brgb = (brgb / num_lines).astype(np.int32)
                return (frgb[::-1], brgb[::-1]) if bgr else (frgb, brgb)
            return [0, 0, 0], [0, 0, 0]
        return frgb, brgb

    def alignment(self) -> int:
        """textblock.py:234-255: 0 left, 1 centre."""
        if self._alignment >= 0:
            return self._alignment
        if self.vertical:
            return 0
        lines = self.lines_array()
        if len(lines) == 1:
            return 0
        polygons = lines.reshape(-1, 8)
        if self.angle != 0:
            polygons = rotate_polygons((0, 0), polygons, self.angle)
        polygons = polygons.reshape(-1, 4, 2)
        left_std = np.std(polygons[:, 0, 0])
        center_std = np.std((polygons[:, 0, 0] + polygons[:, 1, 0]) / 2)
        return 0 if left_std < center_std else 1

    def target_lang(self):
        return self._target_lang

    @property
    def stroke_width(self):
        """textblock.py:260-265."""
        var = np.abs(np.array([self.fg_r, self.fg_g, self.fg_b]) - np.array([self.bg_r, self.bg_g, self.bg_b])).sum()
        return self.default_stroke_width if var > 40 else 0

    def to_dict(self) -> dict:
        """`copy.deepcopy(vars(self))` (textblock.py:158-160): every attribute, numpy values included;
        `annotations.RecordEncoder` turns it into the reference's JSON."""
        return copy.deepcopy(vars(self))


# --------------------------------------------------------------------------

_TAIL_DEFAULTS = tuple((attr, default) for attr, _, default in _RECORD_TAIL)


def _fast_block(xyxy, lines, language, vertical, font_size, distance, angle, vec, norm, merged, weight) -> TextBlock:
    """`TextBlock(...)` without the keyword plumbing: the attribute dict in the reference's creation order."""
    t = TextBlock.__new__(TextBlock)
    d = {"xyxy": xyxy, "lines": lines, "vertical": vertical, "language": language, "font_size": font_size,
         "distance": distance, "angle": angle, "vec": vec, "norm": norm, "merged": merged, "weight": weight}
    for attr, default in _TAIL_DEFAULTS:
        d[attr] = [] if attr == "text" else default
    t.__dict__ = d
    return t


def blocks_from_records(recs, lines: np.ndarray, dist: np.ndarray) -> List[TextBlock]:
    """Native records (`ctd_blk` array + line / distance pools) -> the reference's Python objects.
    `TextBlock.distance` is re-evaluated from its two operands with numpy's own arccos / sin
    (reference utils/textblock.py:327-328), so the record carries the bits the reference's numpy
    expression gives on this machine (csrc/host_group.cpp decided with libm's)."""
    if len(dist):
        with np.errstate(divide="ignore", invalid="ignore"):
            dval = np.abs(np.sin(np.arccos(np.ascontiguousarray(dist[:, 1]))) * np.ascontiguousarray(dist[:, 2]))
    else:
        dval = np.zeros((0,), np.float64)
    out = []
    all_lines = lines.reshape(-1, 4, 2).tolist()
    for r in recs:
        lo, do = r.line_off, r.dist_off
        out.append(_fast_block(list(r.xyxy), all_lines[lo: lo + r.n_lines], LANG_LIST[r.language], bool(r.vertical),
                               float(r.font_size) if r.font_is_float else int(r.font_size),
                               dval[do: do + r.n_dist].copy(), int(r.angle), np.array((r.vec[0], r.vec[1]), np.float64),
                               np.float64(r.norm), bool(r.merged), np.float64(r.weight)))
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
