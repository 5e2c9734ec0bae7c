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


def band_report(prob_ref: np.ndarray, mask_ref: np.ndarray, bitmap: np.ndarray, mask_u8: np.ndarray, eps: float,
                prob: np.ndarray = None, mask: np.ndarray = None, bitmap_thresh: float = 0.3) -> dict:
    """Bounds a reduced-precision engine's deviation at the PIXEL level: every pixel whose thresholded value differs
    from the oracle's must lie within `eps` of the threshold in the ORACLE's map.

    prob_ref / mask_ref : the oracle's shrink map `lines_map[0, 0]` and mask `mask[0, 0]` (f32, HxW)
    bitmap / mask_u8    : the product's fused side outputs (`prob > 0.3`; `(uint8)(mask * 255)`)
    prob / mask         : the product's f32 maps (optional; gives max |delta|)
    Thresholds: the DB bitmap at `bitmap_thresh` (reference utils/db_utils.py:71-72) and the u8 mask at 127
    (BASELINE's "mask bit-exact after uint8 threshold"; u8 > 127 <=> mask * 255 >= 128)."""
    bm_ref = prob_ref > bitmap_thresh
    flips = bm_ref != bitmap.astype(bool)
    d_bm = np.abs(prob_ref.astype(np.float64) - bitmap_thresh)
    u8_ref = (mask_ref * 255).astype(np.uint8)                       # postprocess_mask: truncation
    flips_m = (u8_ref > 127) != (mask_u8 > 127)
    d_m = np.abs(mask_ref.astype(np.float64) * 255 - 128) / 255
    rep = {
        "eps": eps,
        "bitmap_flips": int(flips.sum()),
        "bitmap_flips_max_dist_to_thresh": float(d_bm[flips].max()) if flips.any() else 0.0,
        "bitmap_flips_out_of_band": int((d_bm[flips] >= eps).sum()),
        "bitmap_in_band_frac": round(float((d_bm < eps).mean()), 6),
        "mask127_flips": int(flips_m.sum()),
        "mask127_flips_max_dist_to_thresh": float(d_m[flips_m].max()) if flips_m.any() else 0.0,
        "mask127_flips_out_of_band": int((d_m[flips_m] >= eps).sum()),
        "mask127_in_band_frac": round(float((d_m < eps).mean()), 6),
    }
    if prob is not None:
        rep["prob_max_abs_delta"] = float(np.abs(prob.astype(np.float64) - prob_ref).max())
    if mask is not None:
        rep["mask_max_abs_delta"] = float(np.abs(mask.astype(np.float64) - mask_ref).max())
    rep["_flips"] = flips                                            # for explain_geometry; callers drop it before printing
    return rep


def _box_iou(a, b) -> float:
    x1, y1, x2, y2 = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    i = max(0.0, x2 - x1) * max(0.0, y2 - y1)
    u = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i
    return i / u if u > 0 else 0.0


def explain_geometry(result, ref_result, flips: np.ndarray, ratio_xy=(1.0, 1.0), dets=None, ref_dets=None,
                     score_band_boxes=None, candidates=None, margin: int = 3) -> dict:
    """Attributes every text line / block that is not IDENTICAL between the product's result and the oracle's to one of
    the threshold-type decisions the network's small deviation can move (the tail itself is bit-exact on equal inputs):

      flip   its bounding box (grown by `margin` px) contains a DB-bitmap pixel that flipped (`flips`, network resolution;
             `ratio_xy` maps page to network coordinates) -- a pixel band_report proves to lie within eps of 0.3;
      near   it has a counterpart whose coordinates all lie within 2 px: box coordinates that differ by a fraction of a pixel
             before `astype(int32)` truncation (reference inference.py:106-113) and block-level adjustments fed by them;
      det    it lies in the region of a yolo detection that NMS kept on one side only, or kept with coordinates more than
             1 px apart (`dets` / `ref_dets`: the float (n, >=4) NMS outputs of both sides, network coordinates): greedy NMS
             is order dependent, and random-weight confidences sit on a plateau where a 1e-3 change swaps the order;
      score  it overlaps a DB box whose score in the ORACLE lies within eps of the 0.6 gate (reference inference.py:159;
             `score_band_boxes`: those boxes, (n,4,2) in network coordinates): the score is a mean of the shrink map over the
             box's polygon, so it moves by less than the map error and the line appears on one side only;
      cut    the page has more contours than `max_candidates` (1000, reference utils/db_utils.py:33,137: only the FIRST
             1000 contours are looked at) on both sides, and it overlaps a candidate box that made the cut on one side only:
             a flipped pixel elsewhere adds or removes a contour and shifts the window (`candidates` = (ours (n,4,2), the
             oracle's (m,4,2), max_candidates), network coordinates);
      lines  (blocks only) one of its lines differs, and that line is attributed above.

    `*_unexplained` must be 0 for the claim "every difference comes from a decision variable within the engine's error of
    its threshold"."""
    def boxes_of(res):
        lines = [np.asarray(ln).reshape(-1, 2) for b in res[2] for ln in b.lines]
        blks = [(tuple(int(v) for v in b.xyxy), [np.asarray(ln).reshape(-1, 2) for ln in b.lines]) for b in res[2]]
        return lines, blks

    H, W = flips.shape
    rx, ry = ratio_xy
    ii = np.zeros((H + 1, W + 1), np.int64)
    ii[1:, 1:] = flips.astype(np.int64).cumsum(0).cumsum(1)

    def touched(x1, y1, x2, y2):
        x1, x2 = int(np.floor(x1 * rx)) - margin, int(np.ceil(x2 * rx)) + margin + 1
        y1, y2 = int(np.floor(y1 * ry)) - margin, int(np.ceil(y2 * ry)) + margin + 1
        x1, y1, x2, y2 = max(x1, 0), max(y1, 0), min(x2, W), min(y2, H)
        if x2 <= x1 or y2 <= y1:
            return False
        return (ii[y2, x2] - ii[y1, x2] - ii[y2, x1] + ii[y1, x1]) > 0

    changed = []                                          # regions (page coordinates) of detections that differ
    if dets is not None and ref_dets is not None:
        P, O = np.asarray(dets, np.float64).reshape(-1, np.asarray(dets).shape[-1] if np.size(dets) else 6), \
            np.asarray(ref_dets, np.float64).reshape(-1, np.asarray(ref_dets).shape[-1] if np.size(ref_dets) else 6)
        used = set()
        for o in O:
            best, bj = -1.0, -1
            for j, q in enumerate(P):
                if j not in used:
                    v = _box_iou(o, q)
                    if v > best:
                        best, bj = v, j
            if best > 0.9 and np.abs(P[bj][:4] - o[:4]).max() <= 1.0:
                used.add(bj)
            else:
                changed.append(o[:4])
        changed += [q[:4] for j, q in enumerate(P) if j not in used]
        changed = [np.array([c[0] / rx, c[1] / ry, c[2] / rx, c[3] / ry]) for c in changed]

    def in_changed(x1, y1, x2, y2, m=4):
        return any(not (x2 < c[0] - m or x1 > c[2] + m or y2 < c[1] - m or y1 > c[3] + m) for c in changed)

    sb_boxes = []
    if score_band_boxes is not None and len(score_band_boxes):
        q = np.asarray(score_band_boxes, np.float64).reshape(-1, 4, 2)
        sb_boxes = [np.array([b[:, 0].min() / rx, b[:, 1].min() / ry, b[:, 0].max() / rx, b[:, 1].max() / ry]) for b in q]

    def in_score_band(x1, y1, x2, y2, m=4):
        return any(not (x2 < c[0] - m or x1 > c[2] + m or y2 < c[1] - m or y1 > c[3] + m) for c in sb_boxes)

    cut_boxes = []
    if candidates is not None:
        ca, cb, cmax = candidates
        ca, cb = np.asarray(ca).reshape(-1, 4, 2).astype(np.int64), np.asarray(cb).reshape(-1, 4, 2).astype(np.int64)
        if len(ca) >= cmax and len(cb) >= cmax:
            ka, kb = {q.tobytes() for q in ca}, {q.tobytes() for q in cb}
            one = [q for q in ca if q.tobytes() not in kb] + [q for q in cb if q.tobytes() not in ka]
            cut_boxes = [np.array([b[:, 0].min() / rx, b[:, 1].min() / ry, b[:, 0].max() / rx, b[:, 1].max() / ry]) for b in one]

    def in_cut(x1, y1, x2, y2, m=4):
        return any(not (x2 < c[0] - m or x1 > c[2] + m or y2 < c[1] - m or y1 > c[3] + m) for c in cut_boxes)

    la, ba = boxes_of(result)
    lb, bb = boxes_of(ref_result)
    keyl = lambda q: q.astype(np.int64).tobytes()          # noqa: E731
    sa, sb = {keyl(q) for q in la}, {keyl(q) for q in lb}
    out = {"detections_changed": len(changed), "boxes_in_score_band": len(sb_boxes), "candidates_cut_differently": len(cut_boxes)}
    cat = {"flip": 0, "near": 0, "det": 0, "score": 0, "cut": 0, "unexplained": 0}
    for q, other in [(q, lb) for q in la if keyl(q) not in sb] + [(q, la) for q in lb if keyl(q) not in sa]:
        box = (q[:, 0].min(), q[:, 1].min(), q[:, 0].max(), q[:, 1].max())
        if touched(*box):
            cat["flip"] += 1
        elif any(o.shape == q.shape and np.abs(q - o).max() <= 2 for o in other):
            cat["near"] += 1
        elif in_changed(*box):
            cat["det"] += 1
        elif in_score_band(*box):
            cat["score"] += 1
        elif in_cut(*box):
            cat["cut"] += 1
        else:
            cat["unexplained"] += 1
    out["lines_differing"] = sum(cat.values())
    out["lines_by_cause"] = dict(cat)
    out["lines_unexplained"] = cat["unexplained"]
    keyb = lambda t: (t[0], tuple(sorted(keyl(q) for q in t[1])))      # noqa: E731
    ka, kb = [keyb(t) for t in ba], [keyb(t) for t in bb]
    ska, skb = set(ka), set(kb)
    both = sa & sb
    cat = {"lines": 0, "near": 0, "flip": 0, "det": 0, "score": 0, "cut": 0, "unexplained": 0}
    for (xy, ls), other in [(t, kb) for t in ka if t not in skb] + [(t, ka) for t in kb if t not in ska]:
        if any(ln not in both for ln in ls):
            cat["lines"] += 1
        elif any(o[1] == ls and max(abs(u - v) for u, v in zip(xy, o[0])) <= 2 for o in other):
            cat["near"] += 1
        elif touched(*xy):
            cat["flip"] += 1
        elif in_changed(*xy):
            cat["det"] += 1
        elif in_score_band(*xy):
            cat["score"] += 1
        elif in_cut(*xy):
            cat["cut"] += 1
        else:
            cat["unexplained"] += 1
    out["blocks_differing"] = sum(cat.values())
    out["blocks_by_cause"] = dict(cat)
    out["blocks_unexplained"] = cat["unexplained"]
    return out


def score_band_boxes(lines_map: np.ndarray, input_size, eps: float, box_thresh: float = 0.6) -> np.ndarray:
    """The oracle's DB boxes (all contours, `SegDetectorRepresenter.__call__`) whose score lies within eps of the gate the
    detector applies to them (reference inference.py:159-161): (n,4,2) in network coordinates."""
    from . import postproc_ref as R
    boxes, scores = R.seg_rep(input_size, lines_map)
    b, sc = np.asarray(boxes[0]), np.asarray(scores[0])
    keep = np.abs(sc.astype(np.float64) - box_thresh) < eps
    return b[keep].reshape(-1, 4, 2)
