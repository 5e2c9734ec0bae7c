"""Mask refinement: mirror of reference utils/textmask.py (`refine_mask` :159-169,
`refine_undetected_mask` :135-156 and helpers :16-132).

Formulation: for a candidate mask, a connected component is
OR-ed into the merged mask iff, among its pixels not merged yet, more lie on
predicted-text pixels than on predicted-background pixels -- equivalent to the
reference's `xor_merged < xor_origin` test on the component's bounding box, and
independent of the labelling order.  All connected-component labelling runs on
the GPU (`ctd_ccl`): the windows of a page are stacked into one canvas so a page
needs two labelling launches (candidates, then hole filling) instead of ~6 per
text block.  Candidate generation (Otsu, grey top-k ranges, xor distances) is
small-window integer work done with numpy on the host in this round.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import backend as BK
from .textblock import TextBlock

REFINEMASK_INPAINT = 0
REFINEMASK_ANNOTATION = 1

_RECT = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1))
_CROSS = ((-1, 0), (0, -1), (0, 0), (0, 1), (1, 0))


def _morph(img: np.ndarray, offsets, erode: bool) -> np.ndarray:
    """3x3 erode / dilate; neighbours outside the image are ignored (OpenCV's default border)."""
    h, w = img.shape
    fill = 255 if erode else 0
    pad = np.full((h + 2, w + 2), fill, np.uint8)
    pad[1:-1, 1:-1] = img
    acc = pad[1:-1, 1:-1].copy()
    op = np.minimum if erode else np.maximum
    for dy, dx in offsets:
        acc = op(acc, pad[1 + dy: 1 + dy + h, 1 + dx: 1 + dx + w])
    return acc


def bgr2gray(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY), 8-bit fixed point (textmask.py:58)."""
    c = img.astype(np.int32)
    return ((c[..., 0] * 1868 + c[..., 1] * 9617 + c[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def otsu_value(ch: np.ndarray) -> int:
    """Threshold picked by cv2.threshold(..., THRESH_OTSU) (textmask.py:47), vectorised:
    between-class variance for every split from cumulative histogram sums, first maximum."""
    hist = np.bincount(ch.ravel(), minlength=256).astype(np.float64)
    p = hist / ch.size
    i = np.arange(256, dtype=np.float64)
    q1 = np.cumsum(p)
    m1 = np.cumsum(p * i)
    mu = m1[-1]
    q2 = 1.0 - q1
    eps = float(np.finfo(np.float32).eps)
    valid = (np.minimum(q1, q2) >= eps) & (np.maximum(q1, q2) <= 1.0 - eps)
    with np.errstate(divide="ignore", invalid="ignore"):
        mu1 = m1 / q1
        mu2 = (mu - m1) / q2
        sigma = np.where(valid, q1 * q2 * (mu1 - mu2) ** 2, -1.0)
    best = int(np.argmax(sigma))          # first maximum, like the sequential '>' scan
    return best if sigma[best] > 0 else 0


def _closer_polarity(cand: np.ndarray, msk: np.ndarray) -> Tuple[np.ndarray, int]:
    """`minxor_thresh` (textmask.py:29-41): the candidate or its negative, whichever has the
    smaller L1 distance to the raw 0..255 mask (255^m = 255-m, 0^m = m)."""
    m = msk.astype(np.int64)
    on = cand == 255
    d_pos = int(np.where(on, 255 - m, m).sum())
    d_neg = int(np.where(on, m, 255 - m).sum())
    if d_neg < d_pos:
        return (255 - cand).astype(np.uint8), d_neg
    return cand, d_pos


def candidate_masks(im: np.ndarray, msk: np.ndarray) -> List[Tuple[np.ndarray, int]]:
    """`get_topk_masklist` + `get_otsuthresh_masklist(per_channel=False)` (textmask.py:43-71)."""
    out: List[Tuple[np.ndarray, int]] = []
    grey = bgr2gray(im)
    sel = grey[_morph(msk, _RECT, erode=True) > 127]
    counts, edges = np.histogram(sel, bins=255)                       # (:61) 255 bins over data min..max
    order = np.argsort(-counts, kind="stable")                        # (:17)
    colors, cnt = edges[order], counts[order]                         # edges[:255] pair with counts; edges has 256
    top = [colors[0]]
    tol = cnt.sum() * 0.001
    for c, n in zip(colors[1:], cnt[1:]):
        if np.abs(np.array(top) - c).min() > 10:
            top.append(c)
        if len(top) >= 3 or n < tol:
            break
    for c in top:
        hi = min(c + 30, 255)
        lo = hi - 60
        g = grey.astype(np.float64)
        cand = np.where((g >= lo) & (g <= hi), 255, 0).astype(np.uint8)
        out.append(_closer_polarity(cand, msk))
    best = None
    for ch in range(3):
        plane = np.ascontiguousarray(im[..., ch])
        t = otsu_value(plane)
        cand = np.where(plane > t, 255, 0).astype(np.uint8)
        r = _closer_polarity(cand, msk)
        if best is None or r[1] < best[1]:
            best = r
    out.append(best)
    return out


# --------------------------------------------------------------------------
# batched GPU labelling of many small masks
# --------------------------------------------------------------------------

def label_stack(masks: Sequence[np.ndarray], connectivity: int, device) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Labels every mask (foreground = non-zero) with ONE `ctd_ccl` launch: the masks are
    stacked vertically in a canvas, separated by an empty row, so components cannot join
    and the raster-order label ids of each mask form a contiguous range.
    Returns per mask (labels int32 with local ids 1..n, stats (n,5) [x,y,w,h,area] local)."""
    if not masks:
        return []
    wmax = max(m.shape[1] for m in masks)
    tot = sum(m.shape[0] + 1 for m in masks)
    canvas = np.zeros((tot, wmax), np.uint8)
    tops = []
    y = 0
    for m in masks:
        tops.append(y)
        canvas[y: y + m.shape[0], : m.shape[1]] = (m != 0)
        y += m.shape[0] + 1
    cap = max(1024, int(canvas.sum()) // 1 + 1)
    cap = min(cap, 1 << 20)
    labels, n, stats = BK.connected_components(torch.from_numpy(canvas).to(device), 0, connectivity, max_labels=cap)
    labels = labels[0].cpu().numpy()
    n = int(n[0])
    stats = stats[0, : min(n, cap)].cpu().numpy()
    out = []
    for m, top in zip(masks, tops):
        lab = labels[top: top + m.shape[0], : m.shape[1]]
        nz = lab[lab > 0]
        if nz.size == 0:
            out.append((np.zeros(m.shape, np.int32), np.zeros((0, 5), np.int32)))
            continue
        lo, hi = int(nz.min()), int(nz.max())
        local = np.where(lab > 0, lab - (lo - 1), 0).astype(np.int32)
        st = stats[lo - 1: hi].copy()
        st[:, 1] -= top
        out.append((local, st))
    return out


# --------------------------------------------------------------------------

def _accept(labels: np.ndarray, pred_bin: np.ndarray, merged: np.ndarray, allowed: np.ndarray) -> None:
    """OR the allowed components into `merged` when that lowers the xor distance to `pred_bin`."""
    nlab = allowed.shape[0]
    free = (merged == 0) & (labels > 0)
    on = np.bincount(labels[free & (pred_bin == 255)], minlength=nlab + 1)
    off = np.bincount(labels[free & (pred_bin == 0)], minlength=nlab + 1)
    take = np.zeros(nlab + 1, bool)
    take[1:] = allowed & (on[1:] > off[1:])
    merged[take[labels]] = 255


def refine_mask(img: np.ndarray, pred_mask: np.ndarray, blk_list: Sequence[TextBlock],
                refine_mode: int = REFINEMASK_INPAINT, device="cuda", labeler=None) -> np.ndarray:
    """textmask.py:159-169 for all blocks of one page.  `labeler(masks, connectivity)` defaults to
    the GPU `label_stack`; the CPU test-suite injects its own to exercise the host logic."""
    if labeler is None:
        labeler = lambda masks, conn: label_stack(masks, conn, device)   # noqa: E731
    refined = np.zeros_like(pred_mask)
    im_h, im_w = img.shape[:2]
    jobs = []
    for blk in blk_list:
        x1, y1, x2, y2 = blk.xyxy
        w, h = x2 - x1, y2 - y1
        pad = int(round((max(h, w) * 0.25 + min(h, w) * 0.75) / 16))       # expand_textwindow(expand_r=16)
        x1, y1 = max(0, x1 - pad), max(0, y1 - pad)
        x2, y2 = min(im_w - 1, x2 + pad), min(im_h - 1, y2 + pad)
        im = img[y1:y2, x1:x2]
        msk = np.ascontiguousarray(pred_mask[y1:y2, x1:x2])
        if im.size == 0 or msk.size == 0:
            continue
        cands = candidate_masks(im, msk)
        cands.sort(key=lambda c: c[1])                                      # stable (:74)
        pred_bin = np.where(_morph(msk, _CROSS, erode=True) > 60, 255, 0).astype(np.uint8)   # (:85-89)
        jobs.append(dict(win=(x1, y1, x2, y2), cands=[c[0] for c in cands], pred=pred_bin))
    if not jobs:
        return refined
    # labelling launch 1: every candidate of every window
    flat = [c for j in jobs for c in j["cands"]]
    lab1 = labeler(flat, 8)
    k = 0
    for j in jobs:
        merged = np.zeros_like(j["pred"])
        for _ in j["cands"]:
            labels, st = lab1[k]
            k += 1
            if len(st):
                _accept(labels, j["pred"], merged, (st[:, 2] * st[:, 3]) >= 3)          # (:98-99)
        if refine_mode == REFINEMASK_INPAINT:
            merged = _morph(merged, _RECT, erode=False)                                  # (:110-111)
        j["merged"] = merged
    # labelling launch 2: hole filling on the complements (:113-131)
    lab2 = labeler([255 - j["merged"] for j in jobs], 8)
    for j, (labels, st) in zip(jobs, lab2):
        merged = j["merged"]
        areas = np.r_[int((merged == 255).sum()), st[:, 4]] if len(st) else np.array([int((merged == 255).sum())])
        srt = np.sort(areas)
        thr = srt[-2] if len(srt) > 1 else srt[-1]
        if len(st):
            _accept(labels, j["pred"], merged, st[:, 4] < thr)
        x1, y1, x2, y2 = j["win"]
        refined[y1:y2, x1:x2] |= merged
    return refined


def refine_undetected_mask(img: np.ndarray, mask_pred: np.ndarray, mask_refined: np.ndarray,
                           blk_list: Sequence[TextBlock], refine_mode: int = REFINEMASK_INPAINT,
                           device="cuda", labeler=None) -> np.ndarray:
    """textmask.py:135-156.  Mutates `mask_pred` in place exactly like the reference (:136)."""
    mask_pred[mask_refined > 30] = 0
    if labeler is None:
        labels, n, stats = BK.connected_components(torch.from_numpy(mask_pred).to(device), 30, 4, max_labels=1 << 16)
        n = int(n[0])
        stats = stats[0, : min(n, 1 << 16)].cpu().numpy()
    else:
        stats = labeler([np.where(mask_pred > 30, 255, 0).astype(np.uint8)], 4)[0][1]
    # the reference's stats include the background row 0; `valid_labels[1:]` drops the first
    # row with area > 50, which is the background whenever it is larger than 50 px (:139-142)
    bg_area = mask_pred.size - int(stats[:, 4].sum()) if len(stats) else mask_pred.size
    areas = np.r_[bg_area, stats[:, 4]] if len(stats) else np.array([bg_area])
    valid = np.where(areas > 50)[0]
    new_blks = []
    for li in valid[1:]:
        x, y, w, h, _ = stats[li - 1]
        best = -1
        for blk in blk_list:
            bx1, by1 = max(blk.xyxy[0], x), max(blk.xyxy[1], y)
            bx2, by2 = min(blk.xyxy[2], x + w), min(blk.xyxy[3], y + h)
            s = -1 if (by2 < by1 or bx2 < bx1) else (by2 - by1) * (bx2 - bx1)
            best = max(best, s)
        if best / w / h < 0.5:
            new_blks.append(TextBlock([x, y, x + w, y + h]))
    if new_blks:
        mask_refined = mask_refined | refine_mask(img, mask_pred, new_blks, refine_mode, device, labeler)
    return mask_refined
