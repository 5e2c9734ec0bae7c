"""TEST INFRASTRUCTURE.  CPU restatement (numpy) of the reference's
post-processing, written line-by-line against the reference sources.

Pinning: the CONTROL FLOW restated here (`non_max_suppression`, `SegDetectorRepresenter`,
`group_output`, `refine_mask`, `refine_undetected_mask`) is checked against the reference's OWN
code, imported from /root/reference and run with functional stand-ins for the missing wheels
(`ref_post_import.py`; tests/test_reference_pin.py in the build container, golden outputs
tests/golden/post_seed*.npz + tests/test_oracle_post_golden.py everywhere).

PARITY UNPINNED at the third-party boundary: `torchvision.ops.nms`,
`cv2.connectedComponentsWithStats`, `findContours`, `minAreaRect`, Clipper offsets etc. are not
installed in the build container and the reference ships no tests / golden vectors, so those
primitives (`cv_ref.py` and the helpers below) follow the published algorithms of the libraries
and are cross-checked against scipy.ndimage / brute force only (tests/test_post_host.py).
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
    # the short side as the float32 cv2.minAreaRect returns (Size2f): `sside < 2` (:146) must not depend on whether a side of
    # mathematically 2.0 left the calipers as 2.0 or as 1.9999999999999858 (round 6: 2 of 1 800 random maps)
    return [points[i1], points[i2], points[i3], points[i4]], float(np.float32(min(w, h)))


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
    corners to integers before offsetting (C cast in _to_clipper_path); the JT_ROUND offset ring
    is built the way Clipper 6.4.2 builds it (integer arc points, `cv.clipper_offset_round`) and its
    min-area rectangle is taken."""
    pts = np.asarray(points, np.float32)
    distance = cv.polygon_area(pts) * unclip_ratio / cv.polygon_length(pts)
    ipts = np.trunc(pts.astype(np.float64)).astype(np.int64)
    return get_mini_boxes(cv.clipper_offset_round(ipts, distance))


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


# --------------------------------------------------------------------------
# P9: TextBlock + group_output  (reference utils/textblock.py, utils/imgproc_utils.py)
# --------------------------------------------------------------------------
import copy  # noqa: E402
import math  # noqa: E402

LANG_LIST = ["eng", "ja", "unknown"]      # reference utils/textblock.py:9


class TextBlock:
    """The fields of reference utils/textblock.py:12-86 that the detection path reads/writes."""

    def __init__(self, xyxy, lines=None, language="unknown", vertical=False, font_size=-1, distance=None,
                 angle=0, vec=None, norm=-1, merged=False, weight=-1, text=None, translation="",
                 fg_r=0, fg_g=0, fg_b=0, bg_r=0, bg_g=0, bg_b=0, line_spacing=1., font_family="", bold=False,
                 underline=False, italic=False, alignment=-1, alpha=255, rich_text="", _bounding_rect=None,
                 accumulate_color=True, default_stroke_width=0.2, target_lang="", **kwargs):
        # attribute creation order = key order of the JSON record (textblock.py:45-86)
        self.xyxy = [int(v) for v in xyxy]
        self.lines = [] if lines is None else lines
        self.vertical = vertical
        self.language = language
        self.font_size = font_size
        self.distance = None if distance is None else np.array(distance, np.float64)
        self.angle = angle
        self.vec = None if vec is None else np.array(vec, np.float64)
        self.norm = norm
        self.merged = merged
        self.weight = weight
        self.text = text if text is not None else []              # :58
        self.prob = 1                                              # :59
        self.translation = translation                             # :61
        self.fg_r, self.fg_g, self.fg_b = fg_r, fg_g, fg_b          # :64-66
        self.bg_r, self.bg_g, self.bg_b = bg_r, bg_g, bg_b          # :67-69
        self.font_family = font_family                             # :72
        self.bold = bold
        self.underline = underline
        self.italic = italic
        self.alpha = alpha
        self.rich_text = rich_text
        self.line_spacing = line_spacing
        self._alignment = alignment                                # :80
        self._target_lang = target_lang
        self._bounding_rect = _bounding_rect                       # :83
        self.default_stroke_width = default_stroke_width
        self.accumulate_color = accumulate_color

    def lines_array(self, dtype=np.float64):
        return np.array(self.lines, dtype=dtype)

    def __len__(self):
        return len(self.lines)

    def adjust_bbox(self, with_bbox=False):                       # textblock.py:87-98
        lines = self.lines_array().astype(np.int32)
        if with_bbox:
            self.xyxy[0] = min(lines[..., 0].min(), self.xyxy[0])
            self.xyxy[1] = min(lines[..., 1].min(), self.xyxy[1])
            self.xyxy[2] = max(lines[..., 0].max(), self.xyxy[2])
            self.xyxy[3] = max(lines[..., 1].max(), self.xyxy[3])
        else:
            self.xyxy[0] = lines[..., 0].min()
            self.xyxy[1] = lines[..., 1].min()
            self.xyxy[2] = lines[..., 0].max()
            self.xyxy[3] = lines[..., 1].max()

    def sort_lines(self):                                          # textblock.py:100-105
        if self.distance is not None:
            # the reference's call, literally: numpy's DEFAULT kind, which is not stable and not one algorithm (x86-simd-sort
            # for 64-bit keys on AVX-512 / AVX2 hosts, an introsort elsewhere) -- the order of EQUAL distances in a block of more
            # than 16 lines belongs to the numpy build and the host.  The product calls numpy's own function where the process
            # has it (csrc/host_group.cpp `find_numpy_argsort`), so both follow the reference on the same machine; round 6.
            idx = np.argsort(self.distance)
            self.distance = self.distance[idx]
            lines = np.array(self.lines, dtype=np.int32)
            self.lines = lines[idx].tolist()

    def to_dict(self):                                             # textblock.py:158-160
        import copy
        return copy.deepcopy(vars(self))


def union_area(bboxa, bboxb):
    """reference utils/imgproc_utils.py:13-20 -- the INTERSECTION area, -1 if disjoint."""
    x1, y1 = max(bboxa[0], bboxb[0]), max(bboxa[1], bboxb[1])
    x2, y2 = min(bboxa[2], bboxb[2]), min(bboxa[3], bboxb[3])
    if y2 < y1 or x2 < x1:
        return -1
    return (y2 - y1) * (x2 - x1)


def xywh2xyxypoly(xywh: np.ndarray) -> np.ndarray:
    """reference utils/imgproc_utils.py:31-37."""
    p = np.tile(xywh[:, [0, 1]], 4)
    p[:, [2, 4]] += xywh[:, [2]]
    p[:, [5, 7]] += xywh[:, [3]]
    return p.astype(np.int64)


def examine_textblk(blk: TextBlock, im_w: int, im_h: int, sort: bool = False) -> None:
    """reference utils/textblock.py:302-342."""
    lines = blk.lines_array()
    middle = (lines[:, [1, 2, 3, 0]] + lines) / 2
    vec_v = middle[:, 2] - middle[:, 0]
    vec_h = middle[:, 1] - middle[:, 3]
    center = (lines[:, 0] + lines[:, 2]) / 2
    v, h = np.sum(vec_v, axis=0), np.sum(vec_h, axis=0)
    norm_v, norm_h = np.linalg.norm(v), np.linalg.norm(h)
    vertical = norm_v > norm_h if blk.language == "ja" else norm_v > norm_h * 2        # :312-315
    if vertical:
        primary_vec, primary_norm = v, norm_v
        dvec = center - np.array([[im_w, 0]], dtype=np.float64)
        font_size = int(round(norm_h / len(lines)))
    else:
        primary_vec, primary_norm = h, norm_h
        dvec = center - np.array([[0, 0]], dtype=np.float64)
        font_size = int(round(norm_v / len(lines)))
    rotation_angle = int(math.atan2(primary_vec[1], primary_vec[0]) / math.pi * 180)   # :326 truncation
    distance = np.linalg.norm(dvec, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        rad = np.arccos(np.einsum("ij, j->i", dvec, primary_vec) / (distance * primary_norm))
    distance = np.abs(np.sin(rad) * distance)
    blk.lines = lines.astype(np.int32).tolist()
    blk.distance = distance
    blk.angle = rotation_angle
    if vertical:
        blk.angle -= 90
    if abs(blk.angle) < 3:
        blk.angle = 0
    blk.font_size = font_size
    blk.vertical = vertical
    blk.vec = primary_vec
    blk.norm = primary_norm
    if sort:
        blk.sort_lines()


def try_merge_textline(blk: TextBlock, blk2: TextBlock, fntsize_tol=1.3, distance_tol=2) -> bool:
    """reference utils/textblock.py:344-373."""
    if blk2.merged:
        return False
    with np.errstate(divide="ignore", invalid="ignore"):
        fntsize_div = blk.font_size / blk2.font_size
    n1, n2 = len(blk), len(blk2)
    fntsz_avg = (blk.font_size * n1 + blk2.font_size * n2) / (n1 + n2)
    vec_prod = blk.vec @ blk2.vec
    vec_sum = blk.vec + blk2.vec
    cos_vec = vec_prod / blk.norm / blk2.norm
    distance = blk2.distance[-1] - blk.distance[-1]
    distance_p1 = np.linalg.norm(np.array(blk2.lines[-1][0]) - np.array(blk.lines[-1][0]))
    if not cv.polygons_intersect(blk.lines[-1], blk2.lines[-1]):
        if fntsize_div > fntsize_tol or 1 / fntsize_div > fntsize_tol:
            return False
        if abs(cos_vec) < 0.866:
            return False
        if distance > distance_tol * fntsz_avg or distance_p1 > fntsz_avg * 2.5:
            return False
    blk.lines.append(blk2.lines[0])
    blk.vec = vec_sum
    blk.angle = int(round(np.rad2deg(math.atan2(vec_sum[1], vec_sum[0]))))
    if blk.vertical:
        blk.angle -= 90
    blk.norm = np.linalg.norm(vec_sum)
    blk.distance = np.append(blk.distance, blk2.distance[-1])
    blk.font_size = fntsz_avg
    blk2.merged = True
    return True


def merge_textlines(blk_list):
    """reference utils/textblock.py:375-388."""
    if len(blk_list) < 2:
        return blk_list
    blk_list.sort(key=lambda blk: blk.distance[0])
    merged_list = []
    for ii, cur in enumerate(blk_list):
        if cur.merged:
            continue
        for blk in blk_list[ii + 1:]:
            try_merge_textline(cur, blk)
        merged_list.append(cur)
    for blk in merged_list:
        blk.adjust_bbox(with_bbox=False)
    return merged_list


def split_textblk(blk: TextBlock):
    """reference utils/textblock.py:390-419."""
    font_size, distance, lines = blk.font_size, blk.distance, blk.lines
    l0 = np.array(blk.lines[0])
    lines.sort(key=lambda line: np.linalg.norm(np.array(line[0]) - l0[0]))
    distance_tol = font_size * 2
    current = copy.deepcopy(blk)
    current.lines = [l0]
    sub = [current]
    for jj, line in enumerate(lines[1:]):
        split = False
        if not cv.polygons_intersect(lines[jj], line):
            line_distance = abs(distance[jj + 1] - distance[jj])
            if line_distance > distance_tol:
                split = True
            elif blk.vertical and abs(blk.angle) < 15:
                if len(current.lines) > 1 or line_distance > font_size:
                    split = abs(lines[jj][0][1] - line[0][1]) > font_size
        if split:
            current = copy.deepcopy(current)
            current.lines = [line]
            sub.append(current)
        else:
            current.lines.append(line)
    splitted = len(sub) > 1
    if splitted:
        for c in sub:
            c.adjust_bbox(with_bbox=False)
    return splitted, sub


def sort_textblk_list(blk_list, im_w: int, im_h: int):
    """reference utils/textblock.py:267-300."""
    if len(blk_list) == 0:
        return blk_list
    num_ja = sum(1 for b in blk_list if b.language == "ja")
    xyxy = np.array([b.xyxy for b in blk_list])
    flip_lr = num_ja > len(blk_list) / 2
    im_oriw = im_w
    if im_w > im_h:
        im_w /= 2
    num_gridy, num_gridx = 4, 3
    img_area = im_h * im_w
    center_x = (xyxy[:, 0] + xyxy[:, 2]) / 2
    if flip_lr:
        center_x = (im_oriw - center_x) if im_w != im_oriw else (im_w - center_x)
    grid_x = (center_x / im_w * num_gridx).astype(np.int32)
    center_y = (xyxy[:, 1] + xyxy[:, 3]) / 2
    grid_y = (center_y / im_h * num_gridy).astype(np.int32)
    grid_indices = grid_y * num_gridx + grid_x
    grid_weights = grid_indices * img_area + 1.2 * (center_x - grid_x * im_w / num_gridx) + \
        (center_y - grid_y * im_h / num_gridy)
    if im_w != im_oriw:
        grid_weights[np.where(grid_x >= num_gridx)] += img_area * num_gridy * num_gridx
    for blk, weight in zip(blk_list, grid_weights):
        blk.weight = weight
    blk_list.sort(key=lambda blk: blk.weight)
    return blk_list


def group_output(blks, lines, im_w, im_h, mask=None, sort_blklist=True):
    """reference utils/textblock.py:421-508."""
    blk_list = []
    scattered = {"ver": [], "hor": []}
    for bbox, cls, conf in zip(*blks):
        blk_list.append(TextBlock(bbox, language=LANG_LIST[cls]))
    bbox_score_thresh, mask_score_thresh = 0.4, 0.1
    for line in lines:
        line = np.asarray(line)
        bx1, bx2 = line[:, 0].min(), line[:, 0].max()
        by1, by2 = line[:, 1].min(), line[:, 1].max()
        bbox_score, bbox_idx = -1, -1
        line_area = (by2 - by1) * (bx2 - bx1)
        for jj, blk in enumerate(blk_list):
            with np.errstate(divide="ignore", invalid="ignore"):
                score = union_area(blk.xyxy, [bx1, by1, bx2, by2]) / line_area
            if bbox_score < score:
                bbox_score, bbox_idx = score, jj
        if bbox_score > bbox_score_thresh:
            blk_list[bbox_idx].lines.append(line)
        else:
            if mask is not None:
                with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
                    __import__("warnings").simplefilter("ignore")
                    mask_score = mask[by1:by2, bx1:bx2].mean() / 255
                if mask_score < mask_score_thresh:
                    continue
            blk = TextBlock([bx1, by1, bx2, by2], [line])
            examine_textblk(blk, im_w, im_h, sort=False)
            scattered["ver" if blk.vertical else "hor"].append(blk)

    final = []
    for blk in blk_list:
        if len(blk.lines) == 0:
            bx1, by1, bx2, by2 = blk.xyxy
            if mask is not None:
                with np.errstate(invalid="ignore"), __import__("warnings").catch_warnings():
                    __import__("warnings").simplefilter("ignore")
                    mask_score = mask[by1:by2, bx1:bx2].mean() / 255
                if mask_score < mask_score_thresh:
                    continue
            xywh = np.array([[bx1, by1, bx2 - bx1, by2 - by1]])
            blk.lines = xywh2xyxypoly(xywh).reshape(-1, 4, 2).tolist()
        examine_textblk(blk, im_w, im_h, sort=True)
        splitted = False
        if len(blk.lines) > 1 and (blk.language == "ja" or blk.vertical):
            splitted = True
        if splitted:
            splitted, sub = split_textblk(blk)
        else:
            sub = [blk]
        if not splitted:
            for b in sub:
                b.adjust_bbox(with_bbox=True)
        final += sub

    final += merge_textlines(scattered["hor"])
    final += merge_textlines(scattered["ver"])
    if sort_blklist:
        final = sort_textblk_list(final, im_w, im_h)

    for blk in final:
        if blk.language == "eng" and not blk.vertical:
            if len(blk.lines) == 0:
                continue
            expand_size = max(int(blk.font_size * 0.1), 2)
            rad = np.deg2rad(blk.angle)
            shifted = np.array([[[-1, -1], [1, -1], [1, 1], [-1, 1]]])
            shifted = shifted * np.array([[[np.sin(rad), np.cos(rad)]]]) * expand_size
            lines_ = blk.lines_array() + shifted
            lines_[..., 0] = np.clip(lines_[..., 0], 0, im_w - 1)
            lines_[..., 1] = np.clip(lines_[..., 1], 0, im_h - 1)
            blk.lines = lines_.astype(np.int64).tolist()
            blk.font_size += expand_size
    return final


# --------------------------------------------------------------------------
# P10-P12: mask refinement  (reference utils/textmask.py:16-169, imgproc_utils.py:151-161)
# --------------------------------------------------------------------------
REFINEMASK_INPAINT, REFINEMASK_ANNOTATION = 0, 1


def expand_textwindow(img_size, xyxy, expand_r=8):
    """reference utils/imgproc_utils.py:151-161."""
    im_h, im_w = img_size[:2]
    x1, y1, x2, y2 = xyxy
    w, h = x2 - x1, y2 - y1
    paddings = int(round((max(h, w) * 0.25 + min(h, w) * 0.75) / expand_r))
    x1, y1 = max(0, x1 - paddings), max(0, y1 - paddings)
    x2, y2 = min(im_w - 1, x2 + paddings), min(im_h - 1, y2 + paddings)
    return [x1, y1, x2, y2]


def get_topk_color(color_list, bins, k=3, color_var=10, bin_tol=0.001):
    """reference utils/textmask.py:16-27.  The argsort is the reference's call, literally: numpy's default kind (not stable; bins of
    EQUAL count beyond 16 elements come out in the numpy build's own order -- the product calls numpy's own function where the
    process has it, csrc/np_dispatch.h)."""
    idx = np.argsort(bins * -1)
    color_list, bins = color_list[idx], bins[idx]
    top_colors = [color_list[0]]
    bin_tol = np.sum(bins) * bin_tol
    if len(color_list) > 1:
        for color, bin_ in zip(color_list[1:], bins[1:]):
            if np.abs(np.array(top_colors) - color).min() > color_var:
                top_colors.append(color)
            if len(top_colors) >= k or bin_ < bin_tol:
                break
    return top_colors


def minxor_thresh(threshed, mask):
    """reference utils/textmask.py:29-41 (dilate=False): a {0,255} candidate or its negative,
    whichever is closer (sum of XOR with the raw 0..255 mask) to the predicted mask."""
    neg = 255 - threshed
    neg_xor_sum = np.bitwise_xor(neg, mask).sum(dtype=np.uint64)
    xor_sum = np.bitwise_xor(threshed, mask).sum(dtype=np.uint64)
    if neg_xor_sum < xor_sum:
        return neg, int(neg_xor_sum)
    return threshed, int(xor_sum)


def get_otsuthresh_masklist(img, pred_mask):
    """reference utils/textmask.py:43-54 (per_channel=False): best of the 3 channel Otsu masks."""
    mask_list = []
    for c in range(3):
        _, threshed = cv.threshold_otsu(np.ascontiguousarray(img[..., c]))
        threshed, xor_sum = minxor_thresh(threshed, pred_mask)
        mask_list.append([threshed, xor_sum])
    mask_list.sort(key=lambda x: x[1])
    return [mask_list[0]]


def get_topk_masklist(im_grey, pred_mask):
    """reference utils/textmask.py:56-71."""
    if im_grey.ndim == 3 and im_grey.shape[-1] == 3:
        im_grey = cv.cvt_bgr2gray(im_grey)
    msk = np.ascontiguousarray(pred_mask)
    cand = im_grey[np.where(cv.erode(msk, cv.RECT3, 1) > 127)]
    bin_, his = np.histogram(cand, bins=255)              # names as in the reference (:61)
    topk_color = get_topk_color(his, bin_, color_var=10, k=3)
    color_range = 30
    mask_list = []
    for color in topk_color:
        c_top = min(color + color_range, 255)
        c_bottom = c_top - 2 * color_range
        threshed = cv.in_range(im_grey, c_bottom, c_top)
        threshed, xor_sum = minxor_thresh(threshed, msk)
        mask_list.append([threshed, xor_sum])
    return mask_list


def merge_mask_list(mask_list, pred_mask, pred_thresh=30, refine_mode=REFINEMASK_INPAINT):
    """reference utils/textmask.py:73-132 (blk/filter_with_lines unused by refine_mask)."""
    mask_list.sort(key=lambda x: x[1])
    if pred_thresh > 0:
        pred_mask = cv.erode(pred_mask, cv.CROSS3, 1)                    # :87-88 MORPH_ELLIPSE 3x3
        pred_mask = cv.threshold_binary(pred_mask, 60, 255)              # :89
    mask_merged = np.zeros_like(pred_mask)

    def try_components(labels, stats, area_filter):
        nonlocal mask_merged
        for label_index in range(len(stats)):
            x, y, w, h, area = stats[label_index]
            if not area_filter(label_index, w, h, area):
                continue
            x1, y1, x2, y2 = x, y, x + w, y + h
            local = labels[y1:y2, x1:x2] == label_index
            tmp = np.where(local, 255, mask_merged[y1:y2, x1:x2]).astype(np.uint8)
            pm = pred_mask[y1:y2, x1:x2]
            xor_merged = np.bitwise_xor(tmp, pm).sum(dtype=np.uint64)
            xor_origin = np.bitwise_xor(mask_merged[y1:y2, x1:x2], pm).sum(dtype=np.uint64)
            if xor_merged < xor_origin:
                mask_merged[y1:y2, x1:x2] = tmp

    for candidate, _ in mask_list:
        n, labels, stats = connected_components_with_stats(candidate, 8)                 # :93
        try_components(labels, stats, lambda li, w, h, area: li != 0 and w * h >= 3)     # :95-99

    if refine_mode == REFINEMASK_INPAINT:
        mask_merged = cv.dilate(mask_merged, cv.RECT3, 1)                                # :110-111
    # fill holes (:113-131)
    n, labels, stats = connected_components_with_stats(255 - mask_merged, 8)
    sorted_area = np.sort(stats[:, -1])
    area_thresh = sorted_area[-2] if len(sorted_area) > 1 else sorted_area[-1]
    try_components(labels, stats, lambda li, w, h, area: area < area_thresh)
    return mask_merged


def refine_mask(img, pred_mask, blk_list, refine_mode=REFINEMASK_INPAINT):
    """reference utils/textmask.py:159-169."""
    mask_refined = np.zeros_like(pred_mask)
    for blk in blk_list:
        bx1, by1, bx2, by2 = expand_textwindow(img.shape, blk.xyxy, expand_r=16)
        im = np.ascontiguousarray(img[by1:by2, bx1:bx2])
        msk = np.ascontiguousarray(pred_mask[by1:by2, bx1:bx2])
        if im.size == 0 or msk.size == 0:
            continue
        mask_list = get_topk_masklist(im, msk)
        mask_list += get_otsuthresh_masklist(im, msk)
        mask_merged = merge_mask_list(mask_list, msk, refine_mode=refine_mode)
        mask_refined[by1:by2, bx1:bx2] = np.bitwise_or(mask_refined[by1:by2, bx1:bx2], mask_merged)
    return mask_refined


def refine_undetected_mask(img, mask_pred, mask_refined, blk_list, refine_mode=REFINEMASK_INPAINT):
    """reference utils/textmask.py:135-156 (mutates mask_pred in place like the reference, :136)."""
    mask_pred[np.where(mask_refined > 30)] = 0
    pred_mask_t = cv.threshold_binary(mask_pred, 30, 255)
    n, labels, stats = connected_components_with_stats(pred_mask_t, 4)
    valid_labels = np.where(stats[:, -1] > 50)[0]
    seg_blk_list = []
    if len(valid_labels) > 0:
        for lab_index in valid_labels[1:]:
            x, y, w, h, area = stats[lab_index]
            bbox = [x, y, x + w, y + h]
            bbox_score = -1
            for blk in blk_list:
                bbox_s = union_area(blk.xyxy, bbox)
                if bbox_s > bbox_score:
                    bbox_score = bbox_s
            if bbox_score / w / h < 0.5:
                seg_blk_list.append(TextBlock(bbox))
    if len(seg_blk_list) > 0:
        mask_refined = np.bitwise_or(mask_refined, refine_mask(img, mask_pred, seg_blk_list, refine_mode=refine_mode))
    return mask_refined


# --------------------------------------------------------------------------
# P8 + the whole tail of TextDetector.__call__  (reference inference.py:148-178)
# --------------------------------------------------------------------------

def detector_tail(img, blks, mask, lines_map, input_size=(1024, 1024), dw=0, dh=0, conf_thresh=0.4,
                  nms_thresh=0.35, refine_mode=REFINEMASK_INPAINT, keep_undetected_mask=False):
    """Everything after `self.net(img_in)` for ONE page.  img: BGR uint8 (H,W,3);
    blks (1,rows,no), mask (1,1,Hn,Wn), lines_map (1,2,Hn,Wn) float32 numpy."""
    im_h, im_w = img.shape[:2]
    resize_ratio = (im_w / (input_size[0] - dw), im_h / (input_size[1] - dh))                 # :148
    blks_ = postprocess_yolo(blks, conf_thresh, nms_thresh, resize_ratio)                      # :149
    mask_u8 = postprocess_mask(mask)                                                           # :156
    lines, scores = seg_rep(input_size, lines_map)                                             # :158
    idx = np.where(scores[0] > 0.6)                                                            # :159-161
    lines, scores = lines[0][idx], scores[0][idx]
    mask_u8 = mask_u8[: mask_u8.shape[0] - dh, : mask_u8.shape[1] - dw]                        # :164
    mask_u8 = cv.resize_linear_u8(mask_u8, (im_w, im_h))                                       # :165
    if lines.size == 0:
        lines = []
    else:
        lines = lines.astype(np.float64)
        lines[..., 0] *= resize_ratio[0]
        lines[..., 1] *= resize_ratio[1]
        lines = lines.astype(np.int32)
    blk_list = group_output(blks_, lines, im_w, im_h, mask_u8)                                 # :173
    mask_refined = refine_mask(img, mask_u8, blk_list, refine_mode=refine_mode)                # :174
    if keep_undetected_mask:
        mask_refined = refine_undetected_mask(img, mask_u8, mask_refined, blk_list, refine_mode=refine_mode)
    return mask_u8, mask_refined, blk_list
