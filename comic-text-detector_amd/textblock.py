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


def _turn(pts: np.ndarray, about, degrees) -> np.ndarray:
    """Points (..., 2) turned by `degrees` about `about`, truncated to int64 -- the arithmetic of the reference's
    `rotate_polygons` (utils/imgproc_utils.py:68-84): float32 coordinates, y' = y cos - x sin, x' = y sin + x cos in
    image coordinates, the products formed exactly as numpy forms them there (so the result follows the installed
    numpy's promotion rules like the reference's does)."""
    a = np.deg2rad(degrees)
    s, c = np.sin(a), np.cos(a)
    p = np.asarray(pts).astype(np.float32)               # a copy: shifted in place, in float32 like the reference's
    p[..., 1] -= about[1]
    p[..., 0] -= about[0]
    x, y = p[..., 0], p[..., 1]
    out = np.empty_like(p)
    out[..., 1] = y * c - x * s
    out[..., 0] = y * s + x * c
    out[..., 1] += about[1]
    out[..., 0] += about[0]
    return out.astype(np.int64)


class TextBlock:
    """The reference's TextBlock record (textblock.py:12-265): detection state first, then the
    fields later pipeline stages fill, in the reference's attribute order (that order is the key
    order of the JSON record, `to_dict` :158-160).  The numpy-only helpers downstream stages call on the
    returned `blk_list` (`min_rect`, `bounding_rect`, `aspect_ratio`, `alignment`, `get_text`, the font-colour
    accessors, `stroke_width`, `target_lang` :110-265) are here as well, checked against the reference's own class in
    tests/test_textblock_helpers.py; they cost nothing on the hot path (`blocks_from_records` fills `__dict__`
    directly).  Only `get_transformed_region` (:162-196, cv2 homography of the OCR stage) is left out."""

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

    # -- geometry / text / colour helpers of the reference's class (used by OCR, rendering, GUI code downstream) ------
    def _frame(self, about, sign: int = 1) -> np.ndarray:
        """The lines' corner points (n,4,2), turned by `sign * angle` degrees about `about` when the block is angled."""
        pts = self.lines_array().reshape(-1, 4, 2)
        return _turn(pts, about, sign * self.angle) if self.angle != 0 else pts

    def min_rect(self, rotate_back: bool = True) -> np.ndarray:
        """(1,4,2) int64 corners of the lines' bounding rectangle in the block's own (rotated) frame, turned back into
        page coordinates unless `rotate_back` is False (reference textblock.py:121-134)."""
        c = self.center()
        pts = self._frame(c)
        lo, hi = pts.reshape(-1, 2).min(0), pts.reshape(-1, 2).max(0)
        box = np.array([[[lo[0], lo[1]], [hi[0], lo[1]], [hi[0], hi[1]], [lo[0], hi[1]]]])
        if rotate_back and self.angle != 0:
            box = _turn(box, c, -self.angle)
        return box.astype(np.int64)

    def bounding_rect(self):
        """Qt-style [x, y, w, h] in the rotated frame, or the rectangle a GUI stored (textblock.py:136-144)."""
        if self._bounding_rect is not None:
            return self._bounding_rect
        (x0, y0), _, (x1, y1), _ = self.min_rect(rotate_back=False)[0]
        return [x0, y0, x1 - x0, y1 - y0]

    def aspect_ratio(self) -> float:
        """Height over width of `min_rect()`, measured between the midpoints of opposite edges (textblock.py:110-115)."""
        r = self.min_rect()[0].astype(np.float64)
        mid = (r + np.roll(r, -1, axis=0)) / 2          # midpoints of edges 0-1, 1-2, 2-3, 3-0
        return np.linalg.norm(mid[2] - mid[0]) / np.linalg.norm(mid[1] - mid[3])

    def alignment(self) -> int:
        """0 = left aligned, 1 = centred: a stored value wins; vertical and one-line blocks are left aligned; otherwise
        whichever of the lines' left edges / centres scatters less (textblock.py:234-255)."""
        if self._alignment >= 0:
            return self._alignment
        if self.vertical or len(self.lines) == 1:
            return 0
        pts = self._frame((0, 0))
        left, right = pts[:, 0, 0], pts[:, 1, 0]
        return 0 if np.std(left) < np.std((left + right) / 2) else 1

    def target_lang(self):
        return self._target_lang

    def get_text(self) -> str:
        """The recognised text: a string as it is, a list of line strings joined by blanks (textblock.py:198-201)."""
        return self.text if isinstance(self.text, str) else " ".join(self.text).strip()

    def _colour_scale(self) -> int:
        return len(self.lines) if len(self.lines) > 0 else 1

    def set_font_colors(self, frgb, srgb, accumulate: bool = True) -> None:
        """Stores the fill / stroke colours; with `accumulate` as SUMS over the lines, which is what the per-line OCR
        results add up to (textblock.py:203-211)."""
        self.accumulate_color = accumulate
        k = self._colour_scale() if accumulate else 1
        self.fg_r, self.fg_g, self.fg_b = np.array(frgb) * k
        self.bg_r, self.bg_g, self.bg_b = np.array(srgb) * k

    def get_font_colors(self, bgr: bool = False):
        """(fill, stroke) colours; accumulated sums are averaged over the lines and truncated to int32
        (textblock.py:213-228; like the reference, `bgr` only matters for accumulated colours)."""
        fg = np.array([self.fg_r, self.fg_g, self.fg_b])
        bg = np.array([self.bg_r, self.bg_g, self.bg_b])
        if not self.accumulate_color:
            return fg, bg
        n = len(self.lines)
        if n == 0:
            return [0, 0, 0], [0, 0, 0]
        fg, bg = (fg / n).astype(np.int32), (bg / n).astype(np.int32)
        return (fg[::-1], bg[::-1]) if bgr else (fg, bg)

    @property
    def stroke_width(self):
        """`default_stroke_width` when fill and stroke colours differ by more than 40 in total, else 0 (:260-265)."""
        diff = np.abs(np.array([self.fg_r, self.fg_g, self.fg_b]) - np.array([self.bg_r, self.bg_g, self.bg_b])).sum()
        return self.default_stroke_width if diff > 40 else 0


    def to_dict(self) -> dict:
        """`copy.deepcopy(vars(self))` (textblock.py:158-160): every attribute, numpy values included;
        `annotations.RecordEncoder` turns it into the reference's JSON."""
        return copy.deepcopy(vars(self))


# --------------------------------------------------------------------------

_TAIL_DEFAULTS = tuple((attr, default) for attr, _, default in _RECORD_TAIL)
_TAIL_DICT = {attr: default for attr, default in _TAIL_DEFAULTS}          # "text" gets a fresh list per block
# every attribute in the reference's creation order (the key order of `to_dict()`): the C builder copies this and fills in
_TEMPLATE = dict.fromkeys(("xyxy", "lines", "vertical", "language", "font_size", "distance", "angle", "vec", "norm", "merged",
                           "weight"))
_TEMPLATE.update(_TAIL_DICT)

try:                                   # csrc/pyblocks.c, built by csrc/Makefile next to libctd_hip.so
    from . import _ctd_pyblocks as _PYB
except ImportError as _e:              # not built (a source checkout before `make`): the Python loop below does the same
    _PYB = None
    import logging as _logging
    _logging.getLogger(__name__).warning("csrc/pyblocks.c is not built (%s): TextBlock lists are built by the Python loop, "
                                         "~3 us more interpreter-lock time per block (make -C csrc)", _e)


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
BLK_DTYPE = _BLK_DT = np.dtype([("xyxy", "<i4", (4,)), ("language", "<i4"), ("vertical", "<i4"), ("angle", "<i4"),
                    ("font_is_float", "<i4"), ("font_size", "<f8"), ("vec", "<f8", (2,)), ("norm", "<f8"),
                    ("weight", "<f8"), ("merged", "<i4"), ("line_off", "<i4"), ("n_lines", "<i4"), ("dist_off", "<i4"),
                    ("n_dist", "<i4"), ("pad_", "<i4")])
assert _BLK_DT.itemsize == C.sizeof(L.CtdBlk)


def blocks_from_records(recs, lines: np.ndarray, dist: np.ndarray, n: Optional[int] = None, native: bool = True) -> List[TextBlock]:
    """Native records (`ctd_blk` array + line / distance pools) -> the reference's Python objects.
    `TextBlock.distance` is re-evaluated from its two operands with numpy's own arccos / sin
    (reference utils/textblock.py:327-328), so the record carries the bits the reference's numpy
    expression gives on this machine (csrc/host_group.cpp decided with libm's).
    Column-wise: the per-block Python work is one dict (the interpreter lock is what the tail workers of
    `detect_stream` share, and a crowded page has 60+ blocks) -- and with csrc/pyblocks.c built, one C loop
    (`native=False` forces the Python loop: the two are compared in tests/test_textblock_helpers.py)."""
    if isinstance(recs, list):                       # a slice of a ctypes array is a list of struct copies
        n = len(recs)
        recs = (L.CtdBlk * max(n, 1))(*recs)
    n = len(recs) if n is None else n                # `recs`: the ctypes array the native call filled, first n entries used
    if n == 0:
        return []
    a = recs[:n] if isinstance(recs, np.ndarray) else np.frombuffer(recs, dtype=_BLK_DT, count=n)
    if len(dist):
        with np.errstate(divide="ignore", invalid="ignore"):
            dval = np.abs(np.sin(np.arccos(np.ascontiguousarray(dist[:, 1]))) * np.ascontiguousarray(dist[:, 2]))
    else:
        dval = np.zeros((0,), np.float64)
    all_lines = lines.reshape(-1, 4, 2).tolist()
    if _PYB is not None and native:
        # one C loop over the columns (csrc/pyblocks.c): ~1 us per block instead of 3.7
        return _PYB.build_blocks(TextBlock, _TEMPLATE, n, a["xyxy"].tolist(), all_lines, a["line_off"].tolist(),
                                 a["n_lines"].tolist(), a["vertical"].tolist(), a["language"].tolist(), LANG_LIST,
                                 a["font_size"].tolist(), a["font_is_float"].tolist(), dval, a["dist_off"].tolist(),
                                 a["n_dist"].tolist(), a["angle"].tolist(), list(a["vec"]), list(a["norm"]),
                                 a["merged"].tolist(), list(a["weight"]))
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


def blocks_from_batch(recs: np.ndarray, lines: np.ndarray, dist: np.ndarray, counts: np.ndarray) -> List[List[TextBlock]]:
    """The `blk_list`s of EVERY page of a native tail run in one conversion: `recs` / `lines` / `dist` are the batch-wide
    arrays of `ctd_tail_batch_fetch` (pages back to back, per-page offsets inside the records), `counts[b] = (blocks, lines,
    distances)` of page b.  One pass over the columns and one C loop for the whole work item instead of one per page: the
    fixed cost of a conversion (a dozen column extractions, the `distance` expression) is paid once.  Same objects as
    `blocks_from_records` page by page (tests/test_textblock_helpers.py)."""
    counts = np.asarray(counts).reshape(-1, 3)
    B = len(counts)
    nb = int(counts[:, 0].sum())
    if nb == 0:
        return [[] for _ in range(B)]
    if _PYB is None:
        ob = np.concatenate(([0], np.cumsum(counts[:, 0]))).tolist()
        ol = np.concatenate(([0], np.cumsum(counts[:, 1]))).tolist()
        od = np.concatenate(([0], np.cumsum(counts[:, 2]))).tolist()
        return [blocks_from_records(recs[ob[b]: ob[b + 1]], lines[ol[b]: ol[b + 1]], dist[od[b]: od[b + 1]]) for b in range(B)]
    a = recs[:nb]
    nblk = counts[:, 0].astype(np.int64)
    # the records' line / distance offsets are page-relative: shift them to the batch-wide pools
    lbase = np.repeat(np.concatenate(([0], np.cumsum(counts[:-1, 1]))), nblk)
    dbase = np.repeat(np.concatenate(([0], np.cumsum(counts[:-1, 2]))), nblk)
    nd = int(counts[:, 2].sum())
    if nd:
        with np.errstate(divide="ignore", invalid="ignore"):
            dval = np.abs(np.sin(np.arccos(np.ascontiguousarray(dist[:nd, 1]))) * np.ascontiguousarray(dist[:nd, 2]))
    else:
        dval = np.zeros((0,), np.float64)
    nl = int(counts[:, 1].sum())
    flat = _PYB.build_blocks(TextBlock, _TEMPLATE, nb, a["xyxy"].tolist(), lines[:nl].reshape(-1, 4, 2).tolist(),
                             (a["line_off"] + lbase).tolist(), a["n_lines"].tolist(), a["vertical"].tolist(),
                             a["language"].tolist(), LANG_LIST, a["font_size"].tolist(), a["font_is_float"].tolist(), dval,
                             (a["dist_off"] + dbase).tolist(), a["n_dist"].tolist(), a["angle"].tolist(),
                             list(a["vec"]), list(a["norm"]), a["merged"].tolist(), list(a["weight"]))
    ob = np.concatenate(([0], np.cumsum(nblk))).tolist()
    return [flat[ob[b]: ob[b + 1]] for b in range(B)]


class BlockList:
    """A page's `blk_list` as the native tail left it -- `ctd_blk` records plus line / distance pools, views into the
    batch's arrays -- that turns into the reference's `TextBlock` objects when somebody looks at them.  It behaves like
    the list `TextDetector.__call__` returns (len, iteration, indexing, slicing, `==` with a list, truth value); the
    objects are built once, on first access, on the CONSUMER's thread -- not on the tail worker that produced the page:
    the workers of `detect_stream` share the interpreter lock with each other and with the thread that launches the
    forwards, and at 30-70 blocks per page the per-block Python work was the pipeline's critical resource.
    Columnar access without building anything: `.records` (structured array, one row per block: xyxy, language,
    vertical, angle, font_size, vec, norm, weight, merged, line_off, n_lines, ...), `.line_quads` (n,4,2) i32 in block
    order, `.n_lines`."""
    __slots__ = ("records", "_lines", "_dist", "_built")

    def __init__(self, records: np.ndarray, lines: np.ndarray, dist: np.ndarray):
        self.records, self._lines, self._dist, self._built = records, lines, dist, None

    # -- columnar -------------------------------------------------------------------------------------
    @property
    def line_quads(self) -> np.ndarray:
        return self._lines.reshape(-1, 4, 2)

    @property
    def n_lines(self) -> int:
        return int(self.records["n_lines"].sum()) if len(self.records) else 0

    # -- the reference's list ---------------------------------------------------------------------------
    def to_list(self) -> List["TextBlock"]:
        if self._built is None:
            self._built = blocks_from_records(self.records, self._lines, self._dist)
        return self._built

    def __len__(self) -> int:
        return len(self.records)

    def __iter__(self):
        return iter(self.to_list())

    def __getitem__(self, i):
        return self.to_list()[i]

    def __bool__(self) -> bool:
        return len(self.records) > 0

    def __eq__(self, other):
        if isinstance(other, BlockList):
            other = other.to_list()
        return self.to_list() == other

    def __repr__(self) -> str:
        return f"BlockList({len(self)} blocks, {self.n_lines} lines)"


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
