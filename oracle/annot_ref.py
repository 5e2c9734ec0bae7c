"""TEST INFRASTRUCTURE -- CPU restatement of the reference's annotation writers
(`model2annotations`, reference inference.py:19-70, and the helpers it calls).  Only tests may
import this.  Text formats only; the PNG codec (cv2.imencode in the reference) is not restated --
PNG is lossless, tests compare decoded pixels.

Pinned against the reference's OWN `TextBlock`, `xyxy2yolo` and `get_yololabel_strings`
(imported from /root/reference with stub modules by `oracle/gen_golden_annot.py`, golden strings
in tests/golden/annot_*.json).  `NumpyEncoder` cannot be imported under NumPy 2 (it names the removed
`np.bool8` / `np.float_`, io_utils.py:11-13) and is restated here.
"""
from __future__ import annotations

import io
import json
import os.path as osp
from pathlib import Path

import numpy as np


class NumpyEncoder(json.JSONEncoder):                                  # reference utils/io_utils.py:16-27
    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        elif isinstance(obj, np.generic):
            if isinstance(obj, np.bool_):
                return bool(obj)
            elif isinstance(obj, np.floating):
                return float(obj)
            elif isinstance(obj, np.integer):
                return int(obj)
        return json.JSONEncoder.default(self, obj)


def get_yololabel_strings(clslist, labellist):                         # reference utils/imgproc_utils.py:22-28
    """One `cls x y w h` row per label, numbers through str(); rows joined by newlines, none trailing."""
    rows = [" ".join([str(int(c))] + [str(v) for v in xywh]) for c, xywh in zip(clslist, labellist)]
    return "\n".join(rows)


def xyxy2yolo(xyxy, w, h):                                             # reference utils/imgproc_utils.py:39-51
    """Corner boxes -> normalised (cx, cy, w, h) in float64; None for an empty list."""
    if len(xyxy) == 0:
        return None
    box = np.array(xyxy, dtype=np.float64).reshape(-1, 4)
    box[:, 0::2] = box[:, 0::2] / w
    box[:, 1::2] = box[:, 1::2] / h
    box[:, 2:4] -= box[:, 0:2]
    box[:, 0:2] += box[:, 2:4] / 2
    return box


def page_annotation_texts(imgname, im_w, im_h, blk_list, save_dir="", save_json=True):
    """The text files `model2annotations` writes for one page (inference.py:29-66): {path: content}."""
    imname = imgname.replace(Path(imgname).suffix, '')                 # :32
    polys, blk_xyxy, blk_dict_list = [], [], []
    for blk in blk_list:                                               # :38-41
        polys += blk.lines
        blk_xyxy.append(blk.xyxy)
        blk_dict_list.append(blk.to_dict())
    blk_xyxy = xyxy2yolo(blk_xyxy, im_w, im_h)                         # :42
    if blk_xyxy is not None:
        yolo_label = get_yololabel_strings([1] * len(blk_xyxy), blk_xyxy)   # :43-45
    else:
        yolo_label = ''
    out = {osp.join(save_dir, imname + '.txt'): yolo_label}            # :48-49
    if len(polys) != 0:                                                # :59-63
        arr = np.array(polys).reshape(-1, 8)
        buf = io.StringIO()
        np.savetxt(buf, arr, fmt='%d')
        out[osp.join(save_dir, 'line-' + imname + '.txt')] = buf.getvalue()
    if save_json:                                                      # :64-66
        out[osp.join(save_dir, imname + '.json')] = json.dumps(blk_dict_list, ensure_ascii=False, cls=NumpyEncoder)
    return out


def png_name(img_path, ext='.png'):                                    # reference utils/io_utils.py:47-53
    suffix = Path(img_path).suffix
    if suffix != '':
        return img_path.replace(suffix, ext)
    return img_path + ext
