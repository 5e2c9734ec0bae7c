"""Post-processing of the detector outputs, stage by stage: the host mirror of the reference's
second-level seams.  `TextDetector` runs all of them in one native call
(`tail.Tail.run`); the pieces here serve callers and tests that want one stage.

  postprocess_yolo      reference inference.py:101-114      -> ctd_nms (HIP) + the float32 rescale
  SegRepresenter        reference utils/db_utils.py:32-211  -> ctd_tail_db_boxes: 2x labelling + contour tables on
                                                              the GPU, hull / min-area rectangle / unclip on the host
  group_output          reference utils/textblock.py:421-508 -> textblock.py (ctd_group_output, native host)
  refine_mask           reference utils/textmask.py:159-169  -> textmask.py (ctd_tail_refine)

Nothing here imports `oracle/`.
"""
from __future__ import annotations

from typing import List, Tuple

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import backend as BK
from .tail import thread_tail


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


class SegRepresenter:
    """`SegDetectorRepresenter` (reference utils/db_utils.py:32-69) for the box output.

    The reference walks `cv2.findContours(RETR_LIST)` contours: one per 8-connected foreground
    component (its outer border) and one per enclosed 4-connected background region (a hole border).
    The same set comes from two GPU labelling passes (8-connectivity on the bitmap, 4-connectivity on
    its complement); per contour the GPU also compacts what the host geometry needs -- the row
    extremes (hull points), the sums of the probability map, the containment links
    (csrc/kernels_tail.hip `launch_dbc`) -- so no label image or probability map is downloaded:
      * min-area rectangle of a contour = that of its component's pixels (outer) or of the
        ringing component's pixels that 4-touch the hole (hole border);
      * `box_score_fast` fills the contour polygon = the component with everything it
        encloses (outer) / the hole, its border ring and everything inside the hole (hole border).
    """

    def __init__(self, thresh=0.3, max_candidates=1000, unclip_ratio=1.5):
        self.thresh = thresh
        self.max_candidates = max_candidates
        self.unclip_ratio = unclip_ratio

    def __call__(self, prob: torch.Tensor, bitmap: torch.Tensor) -> Tuple[List[np.ndarray], List[np.ndarray]]:
        """prob (B,H,W) f32 cuda (lines_map[:,0]); bitmap (B,H,W) u8 cuda (prob > thresh, fused
        in the DB tail kernel).  Returns per page boxes int16 (n,4,2) and scores f32 (n,)."""
        return thread_tail(bitmap.device).db_boxes(prob, bitmap, self.max_candidates, self.unclip_ratio)

    # one page from host label images (the path the tail falls back to when a bitmap has more
    # components than the compact tables hold) ---------------------------------------------------
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
