"""Post-processing of the detector outputs: the host mirror of the reference's
second-level seams, driving the HIP kernels for the O(pixels)
work and doing the O(#boxes) scalar geometry on the host.

  postprocess_yolo      reference inference.py:101-114      -> ctd_nms (HIP)
  SegRepresenter        reference utils/db_utils.py:32-211  -> ctd_ccl (HIP) x2 + ctd_db_boxes (native host geometry)
  group_output          reference utils/textblock.py:421-508 -> textblock.py (host, tiny N)
  refine_mask           reference utils/textmask.py:159-169  -> textmask.py

Nothing here imports `oracle/`.
"""
from __future__ import annotations

from typing import List, Tuple

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import backend as BK


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
        n_f, n_b = n_f.cpu().numpy(), n_b.cpu().numpy()
        nmax_f, nmax_b = int(min(n_f.max(initial=0), cap)), int(min(n_b.max(initial=0), cap))
        lab_f, lab_b = BK.to_host(lab_f, "db.lab_f"), BK.to_host(lab_b, "db.lab_b")
        st_f, st_b = BK.to_host(st_f[:, :max(nmax_f, 1)], "db.st_f"), BK.to_host(st_b[:, :max(nmax_b, 1)], "db.st_b")
        prob = BK.to_host(prob, "db.prob")
        boxes_batch, scores_batch = [], []
        for b in range(B):
            boxes, scores = self._page(prob[b], lab_f[b], st_f[b, : min(n_f[b], cap)], lab_b[b],
                                       st_b[b, : min(n_b[b], cap)], W, H)
            boxes_batch.append(boxes)
            scores_batch.append(scores)
        return boxes_batch, scores_batch

    # one page ---------------------------------------------------------------
    def _page(self, prob, lab_f, st_f, lab_b, st_b, W, H):
        """Contours -> boxes for one page: `ctd_db_boxes` (native host geometry, csrc/host_db.cpp)."""
        lib = L.lib()
        cap = int(self.max_candidates)
        boxes = np.zeros((cap, 4, 2), np.int16)
        scores = np.zeros((cap,), np.float32)
        n = C.c_int32(0)
        prob = np.ascontiguousarray(prob, np.float32)
        lab_f, lab_b = np.ascontiguousarray(lab_f, np.int32), np.ascontiguousarray(lab_b, np.int32)
        st_f, st_b = np.ascontiguousarray(st_f, np.int32), np.ascontiguousarray(st_b, np.int32)
        L.check(lib.ctd_db_boxes(prob.ctypes.data, lab_f.ctypes.data, st_f.ctypes.data, len(st_f),
                                 lab_b.ctypes.data, st_b.ctypes.data, len(st_b), W, H, cap,
                                 float(self.unclip_ratio), boxes.ctypes.data, scores.ctypes.data, C.byref(n)),
                "ctd_db_boxes")
        return boxes[: n.value], scores[: n.value]
