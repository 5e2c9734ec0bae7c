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


# --------------------------------------------------------------------------
# P4-P7: SegDetectorRepresenter.boxes_from_bitmap  (reference utils/db_utils.py:32-72,123-211)
# --------------------------------------------------------------------------
from . import cv_ref as cv  # noqa: E402


def get_mini_boxes(contour: np.ndarray, grow: float = 0.0):
    """reference utils/db_utils.py:176-195: minAreaRect -> boxPoints -> sort by x -> TL,TR,BR,BL;
    returns (4 points, short side)."""
    box, w, h = cv.min_area_box(contour, grow)
    points = sorted([p for p in box], key=lambda p: p[0])            # :178 (stable on x)
    if points[1][1] > points[0][1]:
        i1, i4 = 0, 1
    else:
        i1, i4 = 1, 0
    if points[3][1] > points[2][1]:
        i2, i3 = 2, 3
    else:
        i2, i3 = 3, 2
    return [points[i1], points[i2], points[i3], points[i4]], min(w, h)


def box_score_fast(bitmap: np.ndarray, _box: np.ndarray) -> float:
    """reference utils/db_utils.py:197-211: mean of the prob map over the filled polygon."""
    h, w = bitmap.shape[:2]
    box = _box.copy().astype(np.float64)
    xmin = int(np.clip(np.floor(box[:, 0].min()), 0, w - 1))
    xmax = int(np.clip(np.ceil(box[:, 0].max()), 0, w - 1))
    ymin = int(np.clip(np.floor(box[:, 1].min()), 0, h - 1))
    ymax = int(np.clip(np.ceil(box[:, 1].max()), 0, h - 1))
    box[:, 0] -= xmin
    box[:, 1] -= ymin
    mask = cv.fill_poly((ymax - ymin + 1, xmax - xmin + 1), box.astype(np.int32))
    return cv.masked_mean(bitmap[ymin:ymax + 1, xmin:xmax + 1].astype(np.float32), mask)


def unclip_box(points: np.ndarray, unclip_ratio: float = 1.5):
    """reference utils/db_utils.py:168-174 followed by get_mini_boxes (:154).
    distance = area * ratio / perimeter (shapely, float64); pyclipper truncates the float
    corners to integers before offsetting (C cast in _to_clipper_path); the min-area rectangle
    of the JT_ROUND offset polygon is the calipers rectangle of the truncated quad grown by
    `distance` on every side.  UNPINNED: Clipper's integer arc approximation (arc tolerance
    0.25) can move a side by <1 px before the final np.round."""
    pts = np.asarray(points, np.float32)
    distance = cv.polygon_area(pts) * unclip_ratio / cv.polygon_length(pts)
    ipts = np.trunc(pts.astype(np.float64)).astype(np.int64)
    return get_mini_boxes(ipts, grow=distance)


def boxes_from_bitmap(pred: np.ndarray, bitmap: np.ndarray, dest_width: int, dest_height: int,
                      max_candidates: int = 1000, unclip_ratio: float = 1.5):
    """reference utils/db_utils.py:123-166."""
    assert bitmap.ndim == 2                                              # :129
    height, width = bitmap.shape
    contours = cv.find_contours((bitmap * 255).astype(np.uint8))         # :136
    num_contours = min(len(contours), max_candidates)                    # :137
    boxes = np.zeros((num_contours, 4, 2), dtype=np.int16)               # :138
    scores = np.zeros((num_contours,), dtype=np.float32)
    for index in range(num_contours):
        contour = contours[index]
        points, sside = get_mini_boxes(contour)                          # :143
        if sside < 2:                                                    # :146-147
            continue
        points = np.array(points)
        score = box_score_fast(pred, contour)                            # :149 (the CONTOUR polygon)
        box, sside = unclip_box(points, unclip_ratio)                    # :153-154
        box = np.array(box)
        box[:, 0] = np.clip(np.round(box[:, 0] / width * dest_width), 0, dest_width)      # :162
        box[:, 1] = np.clip(np.round(box[:, 1] / height * dest_height), 0, dest_height)   # :163
        boxes[index, :, :] = box.astype(np.int16)                        # :164
        scores[index] = score
    return boxes, scores


def seg_rep(input_size, pred: np.ndarray, thresh: float = 0.3):
    """SegDetectorRepresenter.__call__ (reference utils/db_utils.py:40-69): pred (B,2,H,W)."""
    p0 = pred[:, 0]
    seg = binarize(p0, thresh)
    boxes_batch, scores_batch = [], []
    for b in range(p0.shape[0]):
        h, w = p0.shape[1:]
        boxes, scores = boxes_from_bitmap(p0[b], seg[b], w, h)
        boxes_batch.append(boxes)
        scores_batch.append(scores)
    return boxes_batch, scores_batch
