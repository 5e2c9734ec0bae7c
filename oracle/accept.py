"""TEST INFRASTRUCTURE.  End-to-end acceptance metrics of BASELINE.json's `north_star` ("bboxes /
text-lines IoU >= 0.999, segmentation mask bit-exact after uint8 threshold"): the product detector's
result for a page against the oracle's (fp32 oracle network -> oracle tail, i.e. the reference's
`TextDetector.__call__` restated).  Used by tests/test_gpu_accept.py and by bench.py's `parity` block."""
from __future__ import annotations

import numpy as np

from . import cv_ref as cv


def _iou_masks(a: np.ndarray, b: np.ndarray) -> float:
    u = np.logical_or(a, b).sum()
    return float(np.logical_and(a, b).sum() / u) if u else 1.0


def quad_iou(p, q, shape) -> float:
    """IoU of two quads by rasterisation (cv_ref.fill_poly) inside their common bounding box."""
    p, q = np.asarray(p, np.int64).reshape(-1, 2), np.asarray(q, np.int64).reshape(-1, 2)
    lo = np.minimum(p.min(0), q.min(0))
    hi = np.maximum(p.max(0), q.max(0)) + 1
    hw = (int(hi[1] - lo[1]), int(hi[0] - lo[0]))
    a = cv.fill_poly(hw, p - lo)
    b = cv.fill_poly(hw, q - lo)
    return _iou_masks(a > 0, b > 0)


def match_boxes(ours, theirs, shape):
    """Greedy best-IoU matching of two lists of quads; returns (mean IoU over the reference's boxes with
    unmatched ones counted as 0, min matched IoU, #exactly equal, len(ours), len(theirs))."""
    ours = [np.asarray(o).reshape(-1, 2) for o in ours]
    theirs = [np.asarray(t).reshape(-1, 2) for t in theirs]
    if not theirs:
        return (1.0 if not ours else 0.0), 1.0, 0, len(ours), 0
    used = set()
    ious, exact = [], 0
    for t in theirs:
        best, bj = 0.0, -1
        tlo, thi = t.min(0), t.max(0)
        for j, o in enumerate(ours):
            if j in used:
                continue
            olo, ohi = o.min(0), o.max(0)
            if (ohi < tlo).any() or (thi < olo).any():
                continue
            if np.array_equal(o, t):
                best, bj = 1.0, j
                break
            v = quad_iou(o, t, shape)
            if v > best:
                best, bj = v, j
        if bj >= 0:
            used.add(bj)
            exact += int(best == 1.0 and np.array_equal(ours[bj], t))
        ious.append(best)
    return float(np.mean(ious)), float(min(ious)), exact, len(ours), len(theirs)


def compare(result, ref_result) -> dict:
    """result / ref_result = (mask, mask_refined, blk_list) of the same page."""
    m, r, bl = result
    m0, r0, bl0 = ref_result
    shape = m0.shape
    lines = [ln for b in bl for ln in b.lines]
    lines0 = [ln for b in bl0 for ln in b.lines]
    lm, lmin, lex, ln, ln0 = match_boxes(lines, lines0, shape)
    xy = [[[b.xyxy[0], b.xyxy[1]], [b.xyxy[2], b.xyxy[1]], [b.xyxy[2], b.xyxy[3]], [b.xyxy[0], b.xyxy[3]]] for b in bl]
    xy0 = [[[b.xyxy[0], b.xyxy[1]], [b.xyxy[2], b.xyxy[1]], [b.xyxy[2], b.xyxy[3]], [b.xyxy[0], b.xyxy[3]]] for b in bl0]
    bm, bmin, bex, bn, bn0 = match_boxes(xy, xy0, shape)
    return {
        "mask_u8_equal_frac": round(float((m == m0).mean()), 6),
        "mask_u8_max_level_diff": int(np.abs(m.astype(np.int32) - m0.astype(np.int32)).max()),
        "mask_iou_at_127": round(_iou_masks(m > 127, m0 > 127), 6),
        "refined_mask_iou": round(_iou_masks(r > 0, r0 > 0), 6),
        "refined_mask_equal_frac": round(float((r == r0).mean()), 6),
        "lines": {"ours": ln, "ref": ln0, "identical": lex, "mean_iou": round(lm, 6), "min_iou": round(lmin, 6)},
        "blocks": {"ours": bn, "ref": bn0, "identical": bex, "mean_iou": round(bm, 6), "min_iou": round(bmin, 6)},
    }
