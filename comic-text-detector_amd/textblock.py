"""Block / line grouping: host mirror of reference utils/textblock.py
(`TextBlock` :12-160, `group_output` :421-508 and its helpers).  N is tiny
(<= 300 blocks, <= 1000 lines per page), the arithmetic is scalar float64 with
the reference's truncation points, so this stays on the host.

The implementation is array-oriented where the reference loops (line -> block
assignment is one broadcast; per-block geometry works on (n,4,2) arrays) but
every decision threshold and rounding rule is the reference's; each function
cites the lines it mirrors.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional, Sequence

import numpy as np

from . import geom

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
    """The reference's TextBlock record (textblock.py:12-86): detection state first, then the
    fields later pipeline stages fill, in the reference's attribute order (that order is the key
    order of the JSON record, `to_dict` :158-160)."""

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

def _mask_score(mask: Optional[np.ndarray], x1, y1, x2, y2) -> float:
    """mean(mask[y1:y2, x1:x2]) / 255 with numpy's empty-slice behaviour (nan)."""
    win = mask[y1:y2, x1:x2]
    return float("nan") if win.size == 0 else float(win.mean()) / 255


def examine_textblk(blk: TextBlock, im_w: int, im_h: int, sort: bool = False) -> None:
    """Orientation, angle, font size and reading distance of a block (textblock.py:302-342)."""
    L = blk.lines_array()                                      # (n,4,2)
    mid = (np.roll(L, -1, axis=1) + L) / 2                     # midpoints of edges 0-1,1-2,2-3,3-0
    v = (mid[:, 2] - mid[:, 0]).sum(0)                         # summed "vertical" vectors
    h = (mid[:, 1] - mid[:, 3]).sum(0)
    nv, nh = float(np.linalg.norm(v)), float(np.linalg.norm(h))
    vertical = nv > nh if blk.language == "ja" else nv > nh * 2
    centers = (L[:, 0] + L[:, 2]) / 2
    if vertical:
        pvec, pnorm = v, nv
        d = centers - np.array([[im_w, 0]], np.float64)        # vertical manga text reads right-to-left
        font = int(round(nh / len(L)))
    else:
        pvec, pnorm = h, nh
        d = centers.astype(np.float64)
        font = int(round(nv / len(L)))
    angle = int(math.atan2(pvec[1], pvec[0]) / math.pi * 180)  # truncation (:326)
    dist = np.linalg.norm(d, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        rad = np.arccos((d @ pvec) / (dist * pnorm))
    blk.lines = L.astype(np.int32).tolist()
    blk.distance = np.abs(np.sin(rad) * dist)
    blk.angle = angle - 90 if vertical else angle
    if abs(blk.angle) < 3:
        blk.angle = 0
    blk.font_size = font
    blk.vertical = vertical
    blk.vec = pvec
    blk.norm = pnorm
    if sort:
        blk.sort_lines()


def try_merge_textline(a: TextBlock, b: TextBlock, fntsize_tol: float = 1.3, distance_tol: float = 2) -> bool:
    """textblock.py:344-373."""
    if b.merged:
        return False
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = a.font_size / b.font_size
    na, nb = len(a), len(b)
    avg = (a.font_size * na + b.font_size * nb) / (na + nb)
    vsum = a.vec + b.vec
    cosv = (a.vec @ b.vec) / a.norm / b.norm
    gap = b.distance[-1] - a.distance[-1]
    gap_p1 = np.linalg.norm(np.array(b.lines[-1][0]) - np.array(a.lines[-1][0]))
    if not geom.quads_intersect(a.lines[-1], b.lines[-1]):
        if ratio > fntsize_tol or 1 / ratio > fntsize_tol:
            return False
        if abs(cosv) < 0.866:
            return False
        if gap > distance_tol * avg or gap_p1 > avg * 2.5:
            return False
    a.lines.append(b.lines[0])
    a.vec = vsum
    a.angle = int(round(np.rad2deg(math.atan2(vsum[1], vsum[0]))))
    if a.vertical:
        a.angle -= 90
    a.norm = np.linalg.norm(vsum)
    a.distance = np.append(a.distance, b.distance[-1])
    a.font_size = avg
    b.merged = True
    return True


def merge_textlines(blks: List[TextBlock]) -> List[TextBlock]:
    """Greedy merge of scattered single-line blocks (textblock.py:375-388)."""
    if len(blks) < 2:
        return blks
    blks.sort(key=lambda t: t.distance[0])
    out = []
    for i, cur in enumerate(blks):
        if cur.merged:
            continue
        for other in blks[i + 1:]:
            try_merge_textline(cur, other)
        out.append(cur)
    for t in out:
        t.adjust_bbox(with_bbox=False)
    return out


def _clone_without_lines(blk: TextBlock) -> TextBlock:
    """`copy.deepcopy(blk)` followed by `.lines = [...]` (textblock.py:395-396,409-410) without
    deep-copying the line list that is thrown away: every other mutable attribute gets its own copy."""
    new = copy.copy(blk)
    for k, v in vars(blk).items():
        if k != "lines" and isinstance(v, (list, dict, np.ndarray)):
            setattr(new, k, copy.deepcopy(v))
    new.lines = []
    return new


def split_textblk(blk: TextBlock):
    """Split a vertical / Japanese block at line gaps (textblock.py:390-419)."""
    font, dist, lines = blk.font_size, blk.distance, blk.lines
    first = np.array(lines[0])
    lines.sort(key=lambda q: np.linalg.norm(np.array(q[0]) - first[0]))
    cur = _clone_without_lines(blk)
    cur.lines = [first]
    parts = [cur]
    for j, line in enumerate(lines[1:]):
        split = False
        if not geom.quads_intersect(lines[j], line):
            gap = abs(dist[j + 1] - dist[j])
            if gap > font * 2:
                split = True
            elif blk.vertical and abs(blk.angle) < 15:
                if len(cur.lines) > 1 or gap > font:
                    split = abs(lines[j][0][1] - line[0][1]) > font
        if split:
            cur = _clone_without_lines(cur)
            cur.lines = [line]
            parts.append(cur)
        else:
            cur.lines.append(line)
    if len(parts) > 1:
        for p in parts:
            p.adjust_bbox(with_bbox=False)
        return True, parts
    return False, parts


def sort_textblk_list(blks: List[TextBlock], im_w: int, im_h: int) -> List[TextBlock]:
    """4x3 reading-order grid, right-to-left when most blocks are Japanese (textblock.py:267-300)."""
    if not blks:
        return blks
    box = np.array([t.xyxy for t in blks])
    rtl = sum(t.language == "ja" for t in blks) > len(blks) / 2
    full_w = im_w
    if im_w > im_h:
        im_w /= 2
    gy, gx = 4, 3
    area = im_h * im_w
    cx = (box[:, 0] + box[:, 2]) / 2
    if rtl:
        cx = (full_w - cx) if im_w != full_w else (im_w - cx)
    ix = (cx / im_w * gx).astype(np.int32)
    cy = (box[:, 1] + box[:, 3]) / 2
    iy = (cy / im_h * gy).astype(np.int32)
    w = (iy * gx + ix) * area + 1.2 * (cx - ix * im_w / gx) + (cy - iy * im_h / gy)
    if im_w != full_w:
        w[ix >= gx] += area * gy * gx
    for t, wt in zip(blks, w):
        t.weight = wt
    blks.sort(key=lambda t: t.weight)
    return blks


def group_output(blks, lines, im_w: int, im_h: int, mask: Optional[np.ndarray] = None,
                 sort_blklist: bool = True) -> List[TextBlock]:
    """textblock.py:421-508.  blks = (blines (n,4) i32, cls (n,) i32, confs (n,)); lines (m,4,2) i32."""
    blk_list = [TextBlock(bb, language=LANG_LIST[c]) for bb, c, _ in zip(*blks)]
    scattered = {"ver": [], "hor": []}
    bbox_thr, mask_thr = 0.4, 0.1
    lines = np.asarray(lines)
    if lines.size:
        lo, hi = lines.min(1), lines.max(1)                               # (m,2) bbox of every line
        if blk_list:
            bx = np.array([t.xyxy for t in blk_list], np.float64)       # (n,4)
            ix1 = np.maximum(bx[None, :, 0], lo[:, None, 0])
            iy1 = np.maximum(bx[None, :, 1], lo[:, None, 1])
            ix2 = np.minimum(bx[None, :, 2], hi[:, None, 0])
            iy2 = np.minimum(bx[None, :, 3], hi[:, None, 1])
            inter = np.where((iy2 < iy1) | (ix2 < ix1), -1.0, (iy2 - iy1) * (ix2 - ix1))   # imgproc_utils.py:13-20
            area = ((hi[:, 1] - lo[:, 1]) * (hi[:, 0] - lo[:, 0])).astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                score = inter / area[:, None]                            # (m,n)
        for i, line in enumerate(lines):
            best, best_j = -1, -1
            if blk_list:
                for j in range(len(blk_list)):                             # first maximum, strict '<' (:440-442)
                    if best < score[i, j]:
                        best, best_j = score[i, j], j
            if best > bbox_thr:
                blk_list[best_j].lines.append(line)
                continue
            x1, y1, x2, y2 = lo[i, 0], lo[i, 1], hi[i, 0], hi[i, 1]
            if mask is not None and _mask_score(mask, x1, y1, x2, y2) < mask_thr:
                continue
            t = TextBlock([x1, y1, x2, y2], [line])
            examine_textblk(t, im_w, im_h, sort=False)
            scattered["ver" if t.vertical else "hor"].append(t)

    final: List[TextBlock] = []
    for t in blk_list:
        if not t.lines:
            x1, y1, x2, y2 = t.xyxy
            if mask is not None and _mask_score(mask, x1, y1, x2, y2) < mask_thr:
                continue
            t.lines = [[[x1, y1], [x2, y1], [x2, y2], [x1, y2]]]          # xywh2xyxypoly (imgproc_utils.py:31-37)
        examine_textblk(t, im_w, im_h, sort=True)
        parts, was_split = [t], False
        if len(t.lines) > 1 and (t.language == "ja" or t.vertical):
            was_split, parts = split_textblk(t)
        if not was_split:
            for p in parts:
                p.adjust_bbox(with_bbox=True)
        final += parts

    final += merge_textlines(scattered["hor"])
    final += merge_textlines(scattered["ver"])
    if sort_blklist:
        final = sort_textblk_list(final, im_w, im_h)

    for t in final:                                                       # :492-506
        if t.language == "eng" and not t.vertical and t.lines:
            grow = max(int(t.font_size * 0.1), 2)
            rad = np.deg2rad(t.angle)
            shift = np.array([[[-1, -1], [1, -1], [1, 1], [-1, 1]]]) * np.array([[[np.sin(rad), np.cos(rad)]]]) * grow
            q = t.lines_array() + shift
            q[..., 0] = np.clip(q[..., 0], 0, im_w - 1)
            q[..., 1] = np.clip(q[..., 1], 0, im_h - 1)
            t.lines = q.astype(np.int64).tolist()
            t.font_size += grow
    return final
