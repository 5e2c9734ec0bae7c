"""Mask refinement: mirror of reference utils/textmask.py (`refine_mask` :159-169,
`refine_undetected_mask` :135-156 and helpers :16-132).

GPU / host split (one page = a few launches, independent of the number of text blocks):

  HIP  ctd_win_hist    grey conversion, 3x3 erosion, the four histograms of every window
  host                 top-k grey colours (np.histogram semantics) and Otsu thresholds from them
  HIP  ctd_win_xor     xor distance of the <= 6 candidate rules of every window to the raw mask
  host                 polarity (`minxor_thresh`), best Otsu channel, ordering by distance
  HIP  ctd_win_render  the chosen candidates as bands of one labelling canvas
  HIP  ctd_ccl         8-connected components of all candidates of all windows (one launch)
  HIP  ctd_win_accept  accept / reject per component, one round per candidate rank (count + apply)
  HIP  ctd_win_dilate  3x3 dilation, complement canvas, set-pixel count per window
  HIP  ctd_ccl         components of the complements (hole filling, one launch)
  host                 hole-fill area threshold per window from the component statistics
  HIP  ctd_win_accept  hole-filling accept round;  ctd_win_commit  OR into the page mask

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
    """cv2.cvtColor(BGR2GRAY), OpenCV 4.x 15-bit fixed point (textmask.py:58)."""
    c = img.astype(np.int32)
    return ((c[..., 0] * 3735 + c[..., 1] * 19235 + c[..., 2] * 9798 + 16384) >> 15).astype(np.uint8)


def inrange_bounds(lo: float, hi: float) -> Tuple[int, int]:
    """Integer bounds cv2.inRange uses for double scalars on a u8 image: cvRound (half to even) of
    both, the empty range when lb > ub, lb > 255 or ub < 0, else saturated to [0, 255]."""
    ilo, ihi = int(np.rint(float(lo))), int(np.rint(float(hi)))
    if ilo > ihi or ilo > 255 or ihi < 0:
        return 1, 0
    return max(ilo, 0), min(ihi, 255)


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
    # np.histogram(px, bins=255) without materialising px: the bin of a value only depends on the
    # value and on (min, max), so the grey levels present, weighted by their counts, give the same
    # counts (exact in float64) and the same edges
    hist_sel = np.asarray(hist_sel, np.int64)
    present = np.nonzero(hist_sel)[0]
    if present.size == 0:
        counts, edges = np.histogram(np.zeros(0, np.uint8), bins=255)
    else:
        counts, edges = np.histogram(present.astype(np.uint8), bins=255, range=(int(present[0]), int(present[-1])),
                                     weights=hist_sel[present].astype(np.float64))
        counts = counts.astype(np.int64)
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

    g = grey.astype(np.int32)
    for c in top:
        hi = min(c + 30, 255)
        lb, ub = inrange_bounds(hi - 60, hi)
        out.append(add((g >= lb) & (g <= ub)))
    best = None
    for ch in range(3):
        plane = im[..., ch]
        r = add(plane > otsu_value(np.ascontiguousarray(plane)))
        if best is None or r[1] < best[1]:
            best = r
    out.append(best)
    return out


# --------------------------------------------------------------------------
# canvas layout: windows / candidate masks stacked vertically, one labelling launch for all
# --------------------------------------------------------------------------

def _band_layout(shapes):
    tops, y = [], 0
    for h, _ in shapes:
        tops.append(y)
        y += h + 1                        # an empty row keeps neighbouring bands apart
    return tops, y, max(w for _, w in shapes)


# --------------------------------------------------------------------------
# GPU candidate path
# --------------------------------------------------------------------------

def _gpu_candidates(pages, wins):
    """pages[p] = (img_gpu (H,W,3) u8, mask_gpu (H,W) u8); wins[i] = (p, x1, y1, x2, y2).
    Builds the window table, picks every window's candidate rules (top-k grey ranges, best Otsu
    channel; polarity by xor distance), renders them in the reference's merge order into one canvas
    and labels it.  Nothing but histograms and xor sums leaves the device."""
    lib = L.lib()
    n = len(wins)
    dev = pages[0][0].device
    stream = torch.cuda.current_stream(dev).cuda_stream
    W = (L.CtdWindow * n)()
    for i, (p, x1, y1, x2, y2) in enumerate(wins):
        img_gpu, mask_gpu = pages[p]
        W[i].img, W[i].mask = img_gpu.data_ptr(), mask_gpu.data_ptr()
        W[i].img_w, W[i].mask_w = img_gpu.shape[1], mask_gpu.shape[1]
        W[i].x1, W[i].y1, W[i].w, W[i].h = x1, y1, x2 - x1, y2 - y1
    hist = torch.empty((n, 4, 256), dtype=torch.int32, device=dev)
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
                lb, ub = inrange_bounds(hi - 60, hi)
                r.kind, r.lo, r.hi = 0, float(lb), float(ub)
            else:
                r.kind = -1
        for ch in range(3):
            r = R[i * 6 + 3 + ch]
            r.kind, r.lo = 1 + ch, float(otsu_from_hist(hist[i, 1 + ch]))
    sums = torch.empty((n, 6), dtype=torch.int64, device=dev)
    L.check(lib.ctd_win_xor(W, n, R, 6, sums.data_ptr(), stream), "ctd_win_xor")
    sums = sums.cpu().numpy()
    bands, shapes, owner = [], [], []
    for i, (p, x1, y1, x2, y2) in enumerate(wins):
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
    canvas = torch.zeros((rows, wmax), dtype=torch.uint8, device=dev)
    L.check(lib.ctd_win_render(W, n, Bd, T, len(bands), canvas.data_ptr(), wmax, stream), "ctd_win_render")
    cap = _max_components(shapes)
    labels, nlab, stats = BK.connected_components(canvas, 0, 8, max_labels=cap)
    return W, owner, tops, labels[0], min(int(nlab[0]), cap), stats[0]


def _max_components(shapes) -> int:
    """Upper bound of the 8-connected components of the stacked bands (one per 2x2 cell), so the
    statistics buffer can never truncate."""
    return max(1024, sum(((h + 1) // 2) * ((w + 1) // 2) for h, w in shapes) + 1)


def _refine_gpu(pages, wins, refine_mode: int, out_shapes) -> List[np.ndarray]:
    """Candidates, merge rounds, dilation, hole filling and the final OR on the device for the
    windows of one or more pages at once; the host only sees the histograms, the xor sums and the
    component statistics of the hole-filling pass.  Returns one refined mask per page."""
    lib = L.lib()
    dev = pages[0][0].device
    stream = torch.cuda.current_stream(dev).cuda_stream
    n = len(wins)
    W, owner, tops, labels, nlab, stats = _gpu_candidates(pages, wins)
    shapes_w = [(y2 - y1, x2 - x1) for _, x1, y1, x2, y2 in wins]
    mtops, mrows, mw = _band_layout(shapes_w)
    MT = (C.c_int32 * n)(*mtops)
    merged_a = torch.zeros((mrows, mw), dtype=torch.uint8, device=dev)
    counters = torch.zeros((2 * (nlab + 1),), dtype=torch.int32, device=dev)
    # merge rounds: candidate r of every window in parallel, candidates of one window in order (:93-107)
    seen = [0] * n
    rounds: List[list] = []
    for j, o in enumerate(owner):
        r = seen[o]
        seen[o] += 1
        while len(rounds) <= r:
            rounds.append([])
        rounds[r].append((o, tops[j], mtops[o]))
    if nlab:
        for bands in rounds:
            Bd = (L.CtdBand * len(bands))()
            for k, (o, top, mtop) in enumerate(bands):
                Bd[k].win, Bd[k].top, Bd[k].mtop = o, top, mtop
            L.check(lib.ctd_win_accept(W, n, Bd, len(bands), labels.data_ptr(), labels.shape[1], stats.data_ptr(), None, 3,
                                       merged_a.data_ptr(), mw, counters.data_ptr(), stream), "ctd_win_accept")
    merged_b = torch.empty_like(merged_a)
    comp = torch.zeros_like(merged_a)
    count255 = torch.zeros((n,), dtype=torch.int32, device=dev)
    L.check(lib.ctd_win_dilate(W, n, MT, merged_a.data_ptr(), merged_b.data_ptr(), comp.data_ptr(), mw,
                               count255.data_ptr(), 1 if refine_mode == REFINEMASK_INPAINT else 0, stream), "ctd_win_dilate")
    # hole filling (:113-131): components of the complement, all but the largest area class allowed
    cap2 = _max_components(shapes_w)
    labels2, n2, stats2 = BK.connected_components(comp, 0, 8, max_labels=cap2)
    n2 = min(int(n2[0]), cap2)
    if n2:
        st2 = stats2[0, :n2].cpu().numpy()
        bg = count255.cpu().numpy().astype(np.int64)
        owner2 = np.searchsorted(np.asarray(mtops), st2[:, 1], side="right") - 1
        allowed = np.zeros(n2, np.uint8)
        order = np.argsort(owner2, kind="stable")
        bounds = np.searchsorted(owner2[order], np.arange(n + 1))
        for i in range(n):
            idx = order[bounds[i]: bounds[i + 1]]
            if idx.size == 0:
                continue
            srt = np.sort(np.r_[bg[i], st2[idx, 4]])
            allowed[idx] = st2[idx, 4] < srt[-2]
        allowed_dev = torch.from_numpy(allowed).to(dev)
        counters2 = torch.zeros((2 * (n2 + 1),), dtype=torch.int32, device=dev)
        Bd = (L.CtdBand * n)()
        for i in range(n):
            Bd[i].win, Bd[i].top, Bd[i].mtop = i, mtops[i], mtops[i]
        L.check(lib.ctd_win_accept(W, n, Bd, n, labels2[0].data_ptr(), labels2.shape[2], None, allowed_dev.data_ptr(), 0,
                                   merged_b.data_ptr(), mw, counters2.data_ptr(), stream), "ctd_win_accept")
    # OR into the page masks: one launch per page over that page's (contiguous) windows
    out = []
    first = 0
    for p, shape in enumerate(out_shapes):
        cnt = 0
        while first + cnt < n and wins[first + cnt][0] == p:
            cnt += 1
        page = torch.zeros(shape, dtype=torch.uint8, device=dev)
        if cnt:
            Wp = C.cast(C.byref(W, first * C.sizeof(L.CtdWindow)), C.POINTER(L.CtdWindow))
            MTp = C.cast(C.byref(MT, first * C.sizeof(C.c_int32)), C.POINTER(C.c_int32))
            L.check(lib.ctd_win_commit(Wp, cnt, MTp, merged_b.data_ptr(), mw, page.data_ptr(), shape[1], stream),
                    "ctd_win_commit")
        out.append(page)
        first += cnt
    return [BK.to_host(t, "refine.page").copy() for t in out]


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


def _block_windows(blk_list: Sequence[TextBlock], im_w: int, im_h: int) -> List[Tuple[int, int, int, int]]:
    """`expand_textwindow(expand_r=16)` of every block (reference imgproc_utils.py:151-161, textmask.py:162-164)."""
    wins = []
    for blk in blk_list:
        x1, y1, x2, y2 = blk.xyxy
        w, h = x2 - x1, y2 - y1
        pad = int(round((max(h, w) * 0.25 + min(h, w) * 0.75) / 16))
        x1, y1 = max(0, x1 - pad), max(0, y1 - pad)
        x2, y2 = min(im_w - 1, x2 + pad), min(im_h - 1, y2 + pad)
        if x2 <= x1 or y2 <= y1:
            continue
        wins.append((int(x1), int(y1), int(x2), int(y2)))
    return wins


_GROUP_PIXELS = 2 << 20      # window pixels labelled per launch group (bounds the canvas / stats buffers)


def refine_mask_batch(imgs: Sequence[np.ndarray], pred_masks: Sequence[np.ndarray],
                      blk_lists: Sequence[Sequence[TextBlock]], refine_mode: int = REFINEMASK_INPAINT,
                      device="cuda", gpu: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None) -> List[np.ndarray]:
    """`refine_mask` (textmask.py:159-169) for several pages at once: the windows of all pages share
    the launches (the reference calls it per page and labels per block and candidate).
    gpu[p] = (page BGR u8, mask u8) already resident on the device (else they are uploaded)."""
    n_pages = len(imgs)
    out: List[Optional[np.ndarray]] = [None] * n_pages
    group: List[int] = []
    pix = 0

    def flush():
        nonlocal group, pix
        if not group:
            return
        pages, wins, shapes = [], [], []
        for k, p in enumerate(group):
            if gpu is not None and gpu[p] is not None:
                pages.append(gpu[p])
            else:
                pages.append((torch.from_numpy(np.ascontiguousarray(imgs[p])).to(device),
                              torch.from_numpy(np.ascontiguousarray(pred_masks[p])).to(device)))
            wins += [(k,) + w for w in page_wins[p]]
            shapes.append(pred_masks[p].shape)
        for p, m in zip(group, _refine_gpu(pages, wins, refine_mode, shapes)):
            out[p] = m
        group, pix = [], 0

    page_wins = [_block_windows(blk_lists[p], imgs[p].shape[1], imgs[p].shape[0]) for p in range(n_pages)]
    for p in range(n_pages):
        if not page_wins[p]:
            out[p] = np.zeros_like(pred_masks[p])
            continue
        npx = sum((x2 - x1) * (y2 - y1) for x1, y1, x2, y2 in page_wins[p])
        if group and pix + npx > _GROUP_PIXELS:
            flush()
        group.append(p)
        pix += npx
    flush()
    return out      # type: ignore[return-value]


def refine_mask(img: np.ndarray, pred_mask: np.ndarray, blk_list: Sequence[TextBlock],
                refine_mode: int = REFINEMASK_INPAINT, device="cuda", labeler=None,
                gpu: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> np.ndarray:
    """textmask.py:159-169 for all blocks of one page.
    `gpu` = (page BGR u8, mask u8) already resident on the device (else they are uploaded).
    `labeler(masks, connectivity)`: the CPU test-suite injects its own labeller, which also
    selects the numpy candidate path (no GPU needed)."""
    if labeler is None:
        return refine_mask_batch([img], [pred_mask], [blk_list], refine_mode, device, None if gpu is None else [gpu])[0]
    # ---- numpy path of the CPU test-suite (injected labeller) ----
    refined = np.zeros_like(pred_mask)
    im_h, im_w = img.shape[:2]
    jobs = [dict(win=w) for w in _block_windows(blk_list, im_w, im_h)]
    if not jobs:
        return refined
    cand_labels = []
    for j in jobs:
        x1, y1, x2, y2 = j["win"]
        msk = np.ascontiguousarray(pred_mask[y1:y2, x1:x2])
        j["pred"] = np.where(_morph(msk, _CROSS, erode=True) > 60, 255, 0).astype(np.uint8)   # (:85-89)
        cands = candidate_masks(img[y1:y2, x1:x2], msk)
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
    # hole filling on the complements (:113-131): second labelling pass
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
