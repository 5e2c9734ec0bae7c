"""Mask refinement: mirror of reference utils/textmask.py (`refine_mask` :159-169,
`refine_undetected_mask` :135-156 and helpers :16-132).

GPU / host split (one page = a few launches, independent of the number of text blocks):

  HIP  ctd_win_hist    grey conversion, 3x3 erosion, the four histograms of every window
  host                 top-k grey colours (np.histogram semantics) and Otsu thresholds from them
  HIP  ctd_win_xor     xor distance of the <= 6 candidate rules of every window to the raw mask
  host                 polarity (`minxor_thresh`), best Otsu channel, ordering by distance
  HIP  ctd_win_render  the chosen candidates as bands of one labelling canvas
  HIP  ctd_ccl         8-connected components of all candidates of all windows (one launch)
  host                 accept / reject per component (bincount), 3x3 dilation, hole-fill threshold
  HIP  ctd_ccl         components of the complements (hole filling, one launch)

Accept rule: a component is OR-ed into the merged mask iff, among its
pixels not merged yet, more lie on predicted-text pixels than on predicted-background pixels --
equivalent to the reference's `xor_merged < xor_origin` on the component's bounding box and
independent of the labelling order inside one candidate.

The pure-numpy candidate path (`candidate_masks`) is kept for the CPU test-suite, which injects
a labeller and checks the host logic against the oracle without a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import backend as BK
from .textblock import TextBlock

REFINEMASK_INPAINT = 0
REFINEMASK_ANNOTATION = 1

_RECT = ((-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 0), (0, 1), (1, -1), (1, 0), (1, 1))
_CROSS = ((-1, 0), (0, -1), (0, 0), (0, 1), (1, 0))


# --------------------------------------------------------------------------
# small host helpers (exact integer / float64 arithmetic of the reference)
# --------------------------------------------------------------------------

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


def otsu_from_hist(hist: np.ndarray) -> int:
    """Threshold picked by cv2.threshold(..., THRESH_OTSU) (textmask.py:47) from a 256-bin
    histogram: between-class variance for every split from cumulative sums, first maximum."""
    hist = hist.astype(np.float64)
    p = hist / hist.sum()
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


def otsu_value(ch: np.ndarray) -> int:
    return otsu_from_hist(np.bincount(ch.ravel(), minlength=256))


def topk_colors_from_hist(hist_sel: np.ndarray) -> List[float]:
    """`get_topk_masklist`'s colour pick (textmask.py:61-62, 16-27) from the integer histogram of
    the selected grey values: np.histogram(px, bins=255) only depends on the multiset of values."""
    sel = np.repeat(np.arange(256, dtype=np.uint8), hist_sel.astype(np.int64))
    counts, edges = np.histogram(sel, bins=255)
    order = np.argsort(-counts, kind="stable")
    colors, cnt = edges[order], counts[order]
    top = [colors[0]]
    tol = cnt.sum() * 0.001
    for c, n in zip(colors[1:], cnt[1:]):
        if np.abs(np.array(top) - c).min() > 10:
            top.append(c)
        if len(top) >= 3 or n < tol:
            break
    return top


def _pick(d_pos: int, npix: int) -> Tuple[int, int]:
    """`minxor_thresh` (textmask.py:29-41): (invert, distance); the negative wins only if strictly closer."""
    d_neg = 255 * npix - d_pos
    return (1, d_neg) if d_neg < d_pos else (0, d_pos)


# --------------------------------------------------------------------------
# numpy candidate path (CPU test-suite / reference for the GPU path)
# --------------------------------------------------------------------------

def candidate_masks(im: np.ndarray, msk: np.ndarray) -> List[Tuple[np.ndarray, int]]:
    """`get_topk_masklist` + `get_otsuthresh_masklist(per_channel=False)` (textmask.py:43-71)."""
    grey = bgr2gray(im)
    sel = grey[_morph(msk, _RECT, erode=True) > 127]
    top = topk_colors_from_hist(np.bincount(sel, minlength=256))
    m = msk.astype(np.int64)
    out: List[Tuple[np.ndarray, int]] = []

    def add(on: np.ndarray):
        inv, d = _pick(int(np.where(on, 255 - m, m).sum()), m.size)
        return np.where(on != bool(inv), 255, 0).astype(np.uint8), d

    g = grey.astype(np.float64)
    for c in top:
        hi = min(c + 30, 255)
        out.append(add((g >= hi - 60) & (g <= hi)))
    best = None
    for ch in range(3):
        plane = im[..., ch]
        r = add(plane > otsu_value(np.ascontiguousarray(plane)))
        if best is None or r[1] < best[1]:
            best = r
    out.append(best)
    return out


# --------------------------------------------------------------------------
# GPU labelling of many small masks with one launch
# --------------------------------------------------------------------------

def _split_labels(labels: np.ndarray, stats: np.ndarray, shapes, tops):
    """Canvas labels -> per band (local labels 1..n, local stats).  Bands are stacked vertically,
    so the raster-order ids of a band form a contiguous range."""
    out = []
    for (h, w), top in zip(shapes, tops):
        lab = labels[top: top + h, :w]
        nz = lab[lab > 0]
        if nz.size == 0:
            out.append((np.zeros((h, w), np.int32), np.zeros((0, 5), np.int32)))
            continue
        lo, hi = int(nz.min()), int(nz.max())
        st = stats[lo - 1: hi].copy()
        st[:, 1] -= top
        out.append((np.where(lab > 0, lab - (lo - 1), 0).astype(np.int32), st))
    return out


def _label_canvas(canvas: torch.Tensor, shapes, tops, connectivity: int, fg_upper: int):
    cap = int(min(max(1024, fg_upper + 1), 1 << 20))
    labels, n, stats = BK.connected_components(canvas, 0, connectivity, max_labels=cap)
    labels = labels[0].cpu().numpy()
    stats = stats[0, : min(int(n[0]), cap)].cpu().numpy()
    return _split_labels(labels, stats, shapes, tops)


def _band_layout(shapes):
    tops, y = [], 0
    for h, _ in shapes:
        tops.append(y)
        y += h + 1                        # an empty row keeps neighbouring bands apart
    return tops, y, max(w for _, w in shapes)


def label_stack(masks: Sequence[np.ndarray], connectivity: int, device) -> List[Tuple[np.ndarray, np.ndarray]]:
    """Labels host masks (foreground = non-zero) with ONE `ctd_ccl` launch."""
    if not masks:
        return []
    shapes = [m.shape for m in masks]
    tops, rows, wmax = _band_layout(shapes)
    canvas = np.zeros((rows, wmax), np.uint8)
    for m, top in zip(masks, tops):
        canvas[top: top + m.shape[0], : m.shape[1]] = (m != 0)
    return _label_canvas(torch.from_numpy(canvas).to(device), shapes, tops, connectivity, int(canvas.sum()))


# --------------------------------------------------------------------------
# GPU candidate path
# --------------------------------------------------------------------------

def _gpu_candidates(img_gpu: torch.Tensor, mask_gpu: torch.Tensor, wins: Sequence[Tuple[int, int, int, int]]):
    """For every window (x1,y1,x2,y2): the ordered candidate masks as labelled components.
    Returns per window a list of (labels, stats) in the reference's merge order."""
    lib = L.lib()
    n = len(wins)
    stream = torch.cuda.current_stream(img_gpu.device).cuda_stream
    W = (L.CtdWindow * n)()
    for i, (x1, y1, x2, y2) in enumerate(wins):
        W[i].img, W[i].mask = img_gpu.data_ptr(), mask_gpu.data_ptr()
        W[i].img_w, W[i].mask_w = img_gpu.shape[1], mask_gpu.shape[1]
        W[i].x1, W[i].y1, W[i].w, W[i].h = x1, y1, x2 - x1, y2 - y1
    hist = torch.empty((n, 4, 256), dtype=torch.int32, device=img_gpu.device)
    L.check(lib.ctd_win_hist(W, n, hist.data_ptr(), stream), "ctd_win_hist")
    hist = hist.cpu().numpy().astype(np.int64)
    # rules: 0..2 grey ranges (top-k colours), 3..5 Otsu thresholds of B, G, R
    R = (L.CtdRule * (n * 6))()
    for i in range(n):
        top = topk_colors_from_hist(hist[i, 0])
        for k in range(3):
            r = R[i * 6 + k]
            if k < len(top):
                hi = min(top[k] + 30, 255)
                r.kind, r.lo, r.hi = 0, float(hi - 60), float(hi)
            else:
                r.kind = -1
        for ch in range(3):
            r = R[i * 6 + 3 + ch]
            r.kind, r.lo = 1 + ch, float(otsu_from_hist(hist[i, 1 + ch]))
    sums = torch.empty((n, 6), dtype=torch.int64, device=img_gpu.device)
    L.check(lib.ctd_win_xor(W, n, R, 6, sums.data_ptr(), stream), "ctd_win_xor")
    sums = sums.cpu().numpy()
    bands, shapes, owner = [], [], []
    for i, (x1, y1, x2, y2) in enumerate(wins):
        npix = (x2 - x1) * (y2 - y1)
        cands = []
        for k in range(3):
            if R[i * 6 + k].kind >= 0:
                inv, d = _pick(int(sums[i, k]), npix)
                cands.append((d, k, inv))
        best = None
        for ch in range(3):
            inv, d = _pick(int(sums[i, 3 + ch]), npix)
            if best is None or d < best[0]:
                best = (d, 3 + ch, inv)
        cands.append(best)
        cands.sort(key=lambda c: c[0])                    # stable, like mask_list.sort (textmask.py:74)
        for d, k, inv in cands:
            src = R[i * 6 + k]
            bands.append((src.kind, src.lo, src.hi, inv, i))
            shapes.append((y2 - y1, x2 - x1))
            owner.append(i)
    tops, rows, wmax = _band_layout(shapes)
    Bd = (L.CtdRule * len(bands))()
    for j, (kind, lo, hi, inv, i) in enumerate(bands):
        Bd[j].kind, Bd[j].lo, Bd[j].hi, Bd[j].invert, Bd[j].aux = kind, lo, hi, inv, i
    T = (C.c_int32 * len(bands))(*tops)
    canvas = torch.zeros((rows, wmax), dtype=torch.uint8, device=img_gpu.device)
    L.check(lib.ctd_win_render(W, n, Bd, T, len(bands), canvas.data_ptr(), wmax, stream), "ctd_win_render")
    labelled = _label_canvas(canvas, shapes, tops, 8, sum(h * w for h, w in shapes))
    per_win: List[list] = [[] for _ in range(n)]
    for o, lab in zip(owner, labelled):
        per_win[o].append(lab)
    return per_win


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
                refine_mode: int = REFINEMASK_INPAINT, device="cuda", labeler=None,
                gpu: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> np.ndarray:
    """textmask.py:159-169 for all blocks of one page.
    `gpu` = (page BGR u8, mask u8) already resident on the device (else they are uploaded).
    `labeler(masks, connectivity)`: the CPU test-suite injects its own labeller, which also
    selects the numpy candidate path (no GPU needed)."""
    refined = np.zeros_like(pred_mask)
    im_h, im_w = img.shape[:2]
    jobs = []
    for blk in blk_list:
        x1, y1, x2, y2 = blk.xyxy
        w, h = x2 - x1, y2 - y1
        pad = int(round((max(h, w) * 0.25 + min(h, w) * 0.75) / 16))       # expand_textwindow(expand_r=16)
        x1, y1 = max(0, x1 - pad), max(0, y1 - pad)
        x2, y2 = min(im_w - 1, x2 + pad), min(im_h - 1, y2 + pad)
        if x2 <= x1 or y2 <= y1:
            continue
        msk = np.ascontiguousarray(pred_mask[y1:y2, x1:x2])
        pred_bin = np.where(_morph(msk, _CROSS, erode=True) > 60, 255, 0).astype(np.uint8)   # (:85-89)
        jobs.append(dict(win=(int(x1), int(y1), int(x2), int(y2)), pred=pred_bin))
    if not jobs:
        return refined
    if labeler is None:
        if gpu is None:
            gpu = (torch.from_numpy(np.ascontiguousarray(img)).to(device),
                   torch.from_numpy(np.ascontiguousarray(pred_mask)).to(device))
        cand_labels = _gpu_candidates(gpu[0], gpu[1], [j["win"] for j in jobs])
        labeler2 = lambda masks, conn: label_stack(masks, conn, device)     # noqa: E731
    else:
        cand_labels = []
        for j in jobs:
            x1, y1, x2, y2 = j["win"]
            cands = candidate_masks(img[y1:y2, x1:x2], np.ascontiguousarray(pred_mask[y1:y2, x1:x2]))
            cands.sort(key=lambda c: c[1])                                  # stable (:74)
            cand_labels.append(labeler([c[0] for c in cands], 8))
        labeler2 = labeler
    for j, labs in zip(jobs, cand_labels):
        merged = np.zeros_like(j["pred"])
        for labels, st in labs:
            if len(st):
                _accept(labels, j["pred"], merged, (st[:, 2] * st[:, 3]) >= 3)          # (:98-99)
        if refine_mode == REFINEMASK_INPAINT:
            merged = _morph(merged, _RECT, erode=False)                                  # (:110-111)
        j["merged"] = merged
    # hole filling on the complements (:113-131): second labelling launch
    lab2 = labeler2([255 - j["merged"] for j in jobs], 8)
    for j, (labels, st) in zip(jobs, lab2):
        merged = j["merged"]
        bg = int((merged == 255).sum())
        areas = np.r_[bg, st[:, 4]] if len(st) else np.array([bg])
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
