"""TEST INFRASTRUCTURE.  CPU restatement (numpy) of the reference's
post-processing, written line-by-line against the reference sources.

PARITY UNPINNED at the third-party boundary: `torchvision.ops.nms`,
`cv2.connectedComponentsWithStats` etc. are not installed in the build
container and the reference ships no tests / golden vectors, so these follow
the published algorithms of those libraries and are cross-checked against
scipy.ndimage / brute force only (tests/test_oracle_post.py).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


# --------------------------------------------------------------------------
# P1: non_max_suppression  (reference utils/yolov5_utils.py:124-218)
# --------------------------------------------------------------------------

def xywh2xyxy(x: np.ndarray) -> np.ndarray:
    """reference utils/yolov5_utils.py:220-227 (float32 arithmetic)."""
    y = np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def torchvision_nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.nms CPU kernel (torchvision/csrc/ops/cpu/nms_kernel.cpp,
    `>= 0.8.1` per reference requirements.txt:6): sort by score descending, greedy,
    suppress when inter / (area_i + area_j - inter) > thr.  float32 throughout.
    Ties in score keep the lower index first (stable sort), torchvision leaves
    that order unspecified."""
    boxes = boxes.astype(np.float32)
    x1, y1, x2, y2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    order = np.argsort(-scores.astype(np.float32), kind="stable")
    n = len(order)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = np.float32(iou_threshold)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return np.array(keep, dtype=np.int64)


def non_max_suppression(prediction: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        max_det: int = 300) -> List[np.ndarray]:
    """reference utils/yolov5_utils.py:124-218 with classes=None, agnostic=False,
    multi_label=False, labels=() (the only configuration inference.py uses).
    prediction: (B, rows, 5+nc) f32.  Returns a list of (n,6) [xyxy, conf, cls]."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1                      # :139-140
    prediction = prediction.astype(np.float32)
    xc = prediction[..., 4] > np.float32(conf_thres)                          # :136
    max_wh, max_nms = 4096, 30000                                             # :143-144
    output = [np.zeros((0, 6), np.float32)] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[xc[xi]].copy()                                                  # :155
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]                                                 # :171
        box = xywh2xyxy(x[:, :4])                                             # :174
        j = np.argmax(x[:, 5:], axis=1)                                       # :181 (first max on ties)
        conf = x[np.arange(len(x)), 5 + j]
        keep = conf > np.float32(conf_thres)                                  # :182
        x = np.concatenate([box, conf[:, None], j[:, None].astype(np.float32)], 1)[keep]
        n = x.shape[0]
        if not n:
            continue
        elif n > max_nms:                                                     # :196-197
            x = x[np.argsort(-x[:, 4], kind="stable")[:max_nms]]
        c = x[:, 5:6] * np.float32(max_wh)                                    # :200
        boxes, scores = x[:, :4] + c, x[:, 4]                                 # :201
        i = torchvision_nms(boxes, scores, iou_thres)                         # :202
        if i.shape[0] > max_det:                                              # :203-204
            i = i[:max_det]
        output[xi] = x[i]
    return output


def postprocess_yolo(det: np.ndarray, conf_thresh: float, nms_thresh: float, resize_ratio):
    """reference inference.py:101-114 (det = the (1,rows,no) Detect output)."""
    det = non_max_suppression(det, conf_thresh, nms_thresh)[0].copy()
    det[..., [0, 2]] = det[..., [0, 2]] * resize_ratio[0]
    det[..., [1, 3]] = det[..., [1, 3]] * resize_ratio[1]
    blines = det[..., 0:4].astype(np.int32)
    confs = np.round(det[..., 4], 3)
    cls = det[..., 5].astype(np.int32)
    return blines, cls, confs


# --------------------------------------------------------------------------
# P2/P3: postprocess_mask, binarize
# --------------------------------------------------------------------------

def postprocess_mask(mask: np.ndarray) -> np.ndarray:
    """reference inference.py:85-99 with thresh=None: squeeze, *255, astype(uint8) (truncation)."""
    return (np.squeeze(mask) * 255).astype(np.uint8)


def binarize(pred: np.ndarray, thresh: float = 0.3) -> np.ndarray:
    """reference utils/db_utils.py:71-72."""
    return pred > thresh


# --------------------------------------------------------------------------
# connected components with stats (cv2.connectedComponentsWithStats stand-in)
# --------------------------------------------------------------------------

def connected_components_with_stats(img: np.ndarray, connectivity: int = 8) -> Tuple[int, np.ndarray, np.ndarray]:
    """Semantics of cv2.connectedComponentsWithStats(img, connectivity, CV_16U|CV_32S)
    as the reference uses it (utils/textmask.py:93,113,138): foreground = img != 0,
    label 0 = background, stats rows [x, y, w, h, area] incl. the background row 0.
    Labels are numbered in raster order of each component's first pixel (what
    OpenCV's SAUF gives for 4-connectivity; for 8-connectivity OpenCV's BBDT scans
    2x2 blocks so its numbering can differ -- the reference's results do not depend
    on the numbering)."""
    from scipy import ndimage
    fg = img != 0
    structure = np.ones((3, 3), int) if connectivity == 8 else None
    lab, n = ndimage.label(fg, structure=structure)
    # scipy numbers components in raster order of first pixel already; make it explicit
    if n:
        first = ndimage.minimum_position  # noqa: F841 (documentation only)
        flat = lab.ravel()
        idx = np.nonzero(flat)[0]
        _, first_pos = np.unique(flat[idx], return_index=True)
        order = np.argsort(idx[first_pos], kind="stable")       # component ids sorted by first pixel
        remap = np.zeros(n + 1, np.int64)
        remap[np.arange(1, n + 1)[order]] = np.arange(1, n + 1)
        lab = remap[lab]
    stats = np.zeros((n + 1, 5), np.int32)
    h, w = img.shape
    ys, xs = np.nonzero(~fg)
    if len(ys):
        stats[0] = [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1, len(ys)]
    if n:
        objs = ndimage.find_objects(lab)
        areas = np.bincount(lab.ravel(), minlength=n + 1)
        for l, sl in enumerate(objs, start=1):
            stats[l] = [sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start, areas[l]]
    return n + 1, lab.astype(np.int32), stats
