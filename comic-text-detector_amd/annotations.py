"""Annotation writers: the data formats on the output side of the detector, mirror of the
reference's `model2annotations` (reference inference.py:19-70) and `utils/io_utils.py`.

Per page it writes, into `save_dir`:
  <name>.txt        YOLO labels of the text blocks, class 1, `cls cx cy w h` normalised
                    (inference.py:41-48; utils/imgproc_utils.py:22-28,39-51)
  line-<name>.txt   one text line per row, its 4 corner points as 8 integers (inference.py:59-63,
                    `np.savetxt(fmt='%d')`), only when the page has lines
  <name>.json       the TextBlock records (`to_dict` + NumpyEncoder, inference.py:64-66), optional
  <name>.png        the page itself, re-encoded as PNG (inference.py:67; io_utils.py:47-53)
  mask-<name>.png   the refined mask (inference.py:68)

`model2annotations` drives `TextDetector.detect_batch` (the reference is one page per call) and
keeps image decoding and file writing off the GPU's critical path with a thread pool: decode of
batch k+1 and the writes of batch k-1 overlap the detection of batch k.

PNG / JPEG codecs: Pillow here, OpenCV in the reference; PNG is lossless (pixel-identical files,
not byte-identical), JPEG decoding can differ by a rounding step between codec builds.
"""
from __future__ import annotations

import glob
import io
import json
import os
import os.path as osp
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import List, Sequence, Union

import numpy as np

from .textblock import TextBlock
from .textmask import REFINEMASK_ANNOTATION

IMG_EXT = [".bmp", ".jpg", ".png", ".jpeg"]          # reference utils/io_utils.py:9


class RecordEncoder(json.JSONEncoder):
    """`NumpyEncoder` (reference utils/io_utils.py:16-27): arrays as lists, numpy scalars as Python
    scalars (np.float64 is already a float for json)."""

    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, np.bool_):
            return bool(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.integer):
            return int(obj)
        return json.JSONEncoder.default(self, obj)


def find_all_imgs(img_dir: str, abs_path: bool = False) -> List[str]:
    """reference utils/io_utils.py:29-41: files of `img_dir` whose suffix (case-insensitive) is an image one."""
    out = []
    for filep in glob.glob(osp.join(img_dir, "*")):
        name = osp.basename(filep)
        if Path(name).suffix.lower() in IMG_EXT:
            out.append(filep if abs_path else name)
    return out


def imread(path: str) -> np.ndarray:
    """`cv2.imdecode(..., IMREAD_COLOR)` (io_utils.py:43): 3-channel BGR uint8, alpha dropped, grey expanded."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def png_bytes(img: np.ndarray) -> bytes:
    from PIL import Image
    arr = np.asarray(img)
    pil = Image.fromarray(arr[:, :, ::-1] if arr.ndim == 3 else arr)      # BGR -> RGB; 2-D = greyscale
    buf = io.BytesIO()
    pil.save(buf, format="PNG")
    return buf.getvalue()


def png_path(img_path: str) -> str:
    """`imwrite(path, img, ext='.png')` naming (io_utils.py:47-53): the suffix is REPLACED (first occurrence
    in the whole path, as `str.replace` does) or '.png' appended when there is none."""
    suffix = Path(img_path).suffix
    return img_path.replace(suffix, ".png") if suffix != "" else img_path + ".png"


def yolo_labels(blk_list: Sequence[TextBlock], im_w: int, im_h: int) -> str:
    """inference.py:38-46: class 1 for every block; xyxy -> normalised centre/size in float64
    (imgproc_utils.py:39-51); numbers printed with str(np.float64); no trailing newline."""
    if len(blk_list) == 0:
        return ""
    yolo = np.array([blk.xyxy for blk in blk_list]).astype(np.float64)
    yolo[:, [0, 2]] = yolo[:, [0, 2]] / im_w
    yolo[:, [1, 3]] = yolo[:, [1, 3]] / im_h
    yolo[:, [2, 3]] -= yolo[:, [0, 1]]
    yolo[:, [0, 1]] += yolo[:, [2, 3]] / 2
    return "\n".join("1 " + " ".join(str(e) for e in row) for row in yolo)


def line_polys_text(blk_list: Sequence[TextBlock]) -> str:
    """inference.py:33-37,59-63: all lines of all blocks, 8 integers per row; '' when there are none
    (the reference then writes no file)."""
    polys: list = []
    for blk in blk_list:
        polys += blk.lines
    if len(polys) == 0:
        return ""
    arr = np.array(polys).reshape(-1, 8)
    buf = io.StringIO()
    np.savetxt(buf, arr, fmt="%d")
    return buf.getvalue()


def blocks_json(blk_list: Sequence[TextBlock]) -> str:
    return json.dumps([blk.to_dict() for blk in blk_list], ensure_ascii=False, cls=RecordEncoder)


def page_files(save_dir: str, imgname: str, img: np.ndarray, mask_refined: np.ndarray,
               blk_list: Sequence[TextBlock], save_json: bool = False) -> dict:
    """File name -> content (str or bytes) of one page, without touching the disk."""
    im_h, im_w = img.shape[:2]
    imname = imgname.replace(Path(imgname).suffix, "")              # inference.py:32
    files = {osp.join(save_dir, imname + ".txt"): yolo_labels(blk_list, im_w, im_h)}
    polys = line_polys_text(blk_list)
    if polys:
        files[osp.join(save_dir, "line-" + imname + ".txt")] = polys
    if save_json:
        files[osp.join(save_dir, imname + ".json")] = blocks_json(blk_list)
    files[png_path(osp.join(save_dir, imgname))] = png_bytes(img)
    files[png_path(osp.join(save_dir, "mask-" + imname + ".png"))] = png_bytes(mask_refined)
    return files


def write_files(files: dict) -> None:
    for path, content in files.items():
        if isinstance(content, bytes):
            with open(path, "wb") as f:
                f.write(content)
        else:
            with open(path, "w", encoding="utf8") as f:
                f.write(content)


def write_page_annotations(save_dir, imgname, img, mask_refined, blk_list, save_json=False) -> None:
    write_files(page_files(save_dir, imgname, img, mask_refined, blk_list, save_json))


def model2annotations(model_path: Union[str, dict], img_dir_list, save_dir: str, save_json: bool = False,
                      batch_size: int = 8, device: str = "cuda", detector=None, io_threads: int = 4) -> int:
    """reference inference.py:19-70, batched.  Returns the number of pages written."""
    from .detector import TextDetector
    if isinstance(img_dir_list, str):
        img_dir_list = [img_dir_list]
    det = detector if detector is not None else TextDetector(model_path, input_size=1024, device=device, act="leaky")
    imglist: List[str] = []
    for img_dir in img_dir_list:
        imglist += find_all_imgs(img_dir, abs_path=True)
    os.makedirs(save_dir, exist_ok=True)
    batches = [imglist[i: i + batch_size] for i in range(0, len(imglist), batch_size)]
    with ThreadPoolExecutor(max_workers=io_threads) as pool:
        decode = lambda paths: [pool.submit(imread, p) for p in paths]      # noqa: E731
        pending_writes = []
        nxt = decode(batches[0]) if batches else []
        for bi, paths in enumerate(batches):
            imgs = [f.result() for f in nxt]
            nxt = decode(batches[bi + 1]) if bi + 1 < len(batches) else []   # decode k+1 under detection k
            results = det.detect_batch(imgs, refine_mode=REFINEMASK_ANNOTATION, keep_undetected_mask=True)
            for path, img, (mask, mask_refined, blk_list) in zip(paths, imgs, results):
                pending_writes.append(pool.submit(write_page_annotations, save_dir, osp.basename(path), img,
                                                  mask_refined, blk_list, save_json))
        for f in pending_writes:
            f.result()
    return len(imglist)
