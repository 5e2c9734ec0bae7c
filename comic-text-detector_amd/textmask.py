"""Mask refinement: mirror of reference utils/textmask.py (`refine_mask` :159-169,
`refine_undetected_mask` :135-156) on top of the native tail (`ctd_tail_refine`, csrc/tail.hip).

GPU / host split (a whole batch of pages = a fixed number of launches, independent of the number of
pages, text blocks or candidates):

  HIP  tw_hist      grey conversion, 3x3 erosion, the four histograms of every window
  host              top-k grey colours (np.histogram semantics) and Otsu thresholds (csrc/host_refine.cpp)
  HIP  tw_xor       xor distance of the <= 6 candidate rules of every window to the raw mask
  host              polarity (`minxor_thresh`), best Otsu channel, ordering by distance
  HIP  tw_lds       merge_mask_list of a window as ONE block on bit planes in LDS (round 6, csrc/kernels_twlds.hip): render,
                    8-connected components as a union-find over run ids, merge rounds, dilation, hole filling, commit
  ... and for windows whose planes do not fit the LDS (or whose run table overflowed there) the canvas path:
  HIP  tw_render    the chosen candidates as bands of one packed labelling canvas
  HIP  ccl          8-connected components of all candidates of all windows (one launch)
  HIP  tw_accept    accept / reject per component, one round per candidate rank (count + apply)
  HIP  tw_dilate    3x3 dilation, complement canvas, set-pixel count per window
  HIP  ccl          components of the complements (hole filling, one launch)
  HIP  tw_holes     per-window area threshold (second largest entry), hole-filling accept round
  HIP  tw_commit    OR into the page masks

Accept rule: a component is OR-ed into the merged mask iff, among its
pixels not merged yet, more lie on predicted-text pixels than on predicted-background pixels --
equivalent to the reference's `xor_merged < xor_origin` on the component's bounding box and
independent of the labelling order inside one candidate.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import backend as BK
from .tail import thread_tail
from .textblock import TextBlock

REFINEMASK_INPAINT = 0
REFINEMASK_ANNOTATION = 1


def _to_gpu(img, device) -> torch.Tensor:
    if isinstance(img, torch.Tensor):
        return img.to(device).contiguous()
    return torch.from_numpy(np.ascontiguousarray(img)).to(device)


def refine_mask_batch(imgs: Sequence, pred_masks: Sequence[np.ndarray], blk_lists: Sequence[Sequence[TextBlock]],
                      refine_mode: int = REFINEMASK_INPAINT, device="cuda",
                      gpu: Optional[Sequence[torch.Tensor]] = None) -> List[np.ndarray]:
    """`refine_mask` (textmask.py:159-169) for several pages at once: the windows of all pages share
    the launches (the reference calls it per page and labels per block and candidate).
    gpu[p] = the page BGR u8 already resident on the device (else it is uploaded)."""
    pages = [gpu[p] if gpu is not None and gpu[p] is not None else _to_gpu(imgs[p], device) for p in range(len(imgs))]
    dev = pages[0].device if pages else torch.device(device)
    boxes = [[b.xyxy for b in bl] for bl in blk_lists]
    refined, _ = thread_tail(dev).refine(pages, pred_masks, boxes, refine_mode, False)
    return refined


def refine_mask(img, pred_mask: np.ndarray, blk_list: Sequence[TextBlock], refine_mode: int = REFINEMASK_INPAINT,
                device="cuda", gpu: Optional[torch.Tensor] = None) -> np.ndarray:
    """textmask.py:159-169 for all blocks of one page."""
    return refine_mask_batch([img], [pred_mask], [blk_list], refine_mode, device, None if gpu is None else [gpu])[0]


def refine_undetected_mask(img, mask_pred: np.ndarray, mask_refined: np.ndarray, blk_list: Sequence[TextBlock],
                           refine_mode: int = REFINEMASK_INPAINT, device="cuda") -> np.ndarray:
    """textmask.py:135-156 with the caller's `mask_refined`.  Mutates `mask_pred` in place exactly like
    the reference (:136).  (`TextDetector` runs the same pass inside the native tail, on the device.)"""
    mask_pred[mask_refined > 30] = 0
    labels, n, stats = BK.connected_components(torch.from_numpy(mask_pred).to(device), 30, 4, max_labels=1 << 16)
    n = int(n[0])
    stats = stats[0, : min(n, 1 << 16)].cpu().numpy()
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
        mask_refined = mask_refined | refine_mask(img, mask_pred, new_blks, refine_mode, device)
    return mask_refined
