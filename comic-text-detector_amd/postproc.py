"""Post-processing of the detector outputs: the host mirror of the reference's
second-level seams, driving the HIP kernels for the O(pixels)
work and doing the O(#boxes) scalar geometry on the host.

  postprocess_yolo      reference inference.py:101-114      -> ctd_nms (HIP)
  SegRepresenter        reference utils/db_utils.py:32-211  -> ctd_ccl (HIP) x2 + host geometry
  group_output          reference utils/textblock.py:421-508 -> textblock.py (host, tiny N)
  refine_mask           reference utils/textmask.py:159-169  -> textmask.py

Nothing here imports `oracle/`.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch

from . import backend as BK
from . import geom


# --------------------------------------------------------------------------
# YOLO blocks
# --------------------------------------------------------------------------

def postprocess_yolo(blks: torch.Tensor, conf_thresh: float, nms_thresh: float, resize_ratios):
    """blks (B,rows,no) on the GPU -> per page (blines i32 (n,4), cls i32 (n,), confs f32 (n,)).
    `resize_ratios`: one (rx, ry) per page (reference inference.py:148)."""
    dets, counts = BK.nms(blks, conf_thresh, nms_thresh)
    dets = dets.cpu().numpy()
    counts = counts.cpu().numpy()
    out = []
    for b in range(dets.shape[0]):
        d = dets[b, : counts[b]].copy()
        rx, ry = resize_ratios[b]
        d[:, [0, 2]] = d[:, [0, 2]] * rx                       # float32 * python float -> float32, as numpy does
        d[:, [1, 3]] = d[:, [1, 3]] * ry
        out.append((d[:, 0:4].astype(np.int32), d[:, 5].astype(np.int32), np.round(d[:, 4], 3)))
    return out


# --------------------------------------------------------------------------
# DB text lines: bitmap -> boxes
# --------------------------------------------------------------------------

class SegRepresenter:
    """`SegDetectorRepresenter` (reference utils/db_utils.py:32-69) for the box output.

    The reference walks `cv2.findContours(RETR_LIST)` contours: one per 8-connected
    foreground component (its outer border) and one per enclosed 4-connected background
    region (a hole border).  The same set is obtained here from two GPU labelling passes
    (`ctd_ccl`, 8-connectivity on the bitmap, 4-connectivity on its complement):
      * min-area rectangle of a contour = that of its component's pixels (outer) or of the
        hole grown by its 4-neighbourhood (hole border pixels);
      * `box_score_fast` fills the contour polygon = the component with everything it
        encloses (outer) / the hole, its border ring and any islands inside (hole).
    """

    def __init__(self, thresh=0.3, max_candidates=1000, unclip_ratio=1.5):
        self.thresh = thresh
        self.max_candidates = max_candidates
        self.unclip_ratio = unclip_ratio

    def __call__(self, prob: torch.Tensor, bitmap: torch.Tensor) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """prob (B,H,W) f32 cuda (lines_map[:,0]); bitmap (B,H,W) u8 cuda (prob > thresh, fused
        in the DB tail kernel).  Returns per page boxes int16 (n,4,2) and scores f32 (n,)."""
        B, H, W = bitmap.shape
        cap = 1 << 16
        lab_f, n_f, st_f = BK.connected_components(bitmap, 0, 8, max_labels=cap)
        inv = (bitmap == 0).to(torch.uint8)
        lab_b, n_b, st_b = BK.connected_components(inv, 0, 4, max_labels=cap)
        lab_f, lab_b = lab_f.cpu().numpy(), lab_b.cpu().numpy()
        n_f, n_b = n_f.cpu().numpy(), n_b.cpu().numpy()
        st_f, st_b = st_f.cpu().numpy(), st_b.cpu().numpy()
        prob = prob.cpu().numpy()
        boxes_batch, scores_batch = [], []
        for b in range(B):
            boxes, scores = self._page(prob[b], lab_f[b], st_f[b, : min(n_f[b], cap)], lab_b[b],
                                       st_b[b, : min(n_b[b], cap)], W, H)
            boxes_batch.append(boxes)
            scores_batch.append(scores)
        return boxes_batch, scores_batch

    # one page ---------------------------------------------------------------
    def _page(self, prob, lab_f, st_f, lab_b, st_b, W, H):
        from scipy import ndimage
        items = []   # (discovery key, kind, label)
        for l, (x, y, w, h, a) in enumerate(st_f, start=1):
            # outer border starts at the component's first pixel in raster order
            row = lab_f[y, x: x + w]
            fx = x + int(np.argmax(row == l))
            items.append((y * W + fx, 0, l))
        for l, (x, y, w, h, a) in enumerate(st_b, start=1):
            if x == 0 or y == 0 or x + w == W or y + h == H:
                continue                      # touches the frame: the outer background, no hole border
            row = lab_b[y, x: x + w]
            fx = x + int(np.argmax(row == l))
            items.append((y * W + fx - 1, 1, l))    # hole border starts at the pixel left of the hole
        # OpenCV hands RETR_LIST contours back newest first
        items.sort(key=lambda t: t[0], reverse=True)
        items = items[: self.max_candidates]
        n = len(items)
        boxes = np.zeros((n, 4, 2), np.int16)
        scores = np.zeros((n,), np.float32)
        for idx, (_, kind, l) in enumerate(items):
            if kind == 0:
                x, y, w, h, _ = st_f[l - 1]
                comp = lab_f[y: y + h, x: x + w] == l
                ys, xs = np.nonzero(comp)
                pts = np.stack([xs + x, ys + y], 1)
                region = ndimage.binary_fill_holes(comp)
                x0, y0 = x, y
            else:
                x, y, w, h, _ = st_b[l - 1]
                x0, y0 = x - 1, y - 1
                hole = np.zeros((h + 2, w + 2), bool)
                hole[1:-1, 1:-1] = lab_b[y: y + h, x: x + w] == l
                ring = np.zeros_like(hole)
                ring[1:, :] |= hole[:-1, :]
                ring[:-1, :] |= hole[1:, :]
                ring[:, 1:] |= hole[:, :-1]
                ring[:, :-1] |= hole[:, 1:]
                ring &= ~hole
                ys, xs = np.nonzero(ring)
                pts = np.stack([xs + x0, ys + y0], 1)
                region = ndimage.binary_fill_holes(hole | ring)
            box, bw, bh = geom.min_area_box(pts)
            if min(bw, bh) < 2:                                   # db_utils.py:146-147
                continue
            box = geom.order_box(box)
            rh, rw = region.shape
            scores[idx] = prob[y0: y0 + rh, x0: x0 + rw][region].astype(np.float64).mean()
            # unclip (db_utils.py:168-174) + get_mini_boxes (:154): pyclipper truncates the corners
            # to integers; the round-join offset's min-area rectangle = calipers rectangle + distance
            dist = geom.quad_area(box) * self.unclip_ratio / geom.quad_perimeter(box)
            ub, _, _ = geom.min_area_box(np.trunc(box.astype(np.float64)), grow=dist)
            ub = geom.order_box(ub)
            ub[:, 0] = np.clip(np.round(ub[:, 0] / W * W), 0, W)    # dest size == bitmap size (inference.py:158)
            ub[:, 1] = np.clip(np.round(ub[:, 1] / H * H), 0, H)
            boxes[idx] = ub.astype(np.int16)
        return boxes, scores
