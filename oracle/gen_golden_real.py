#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference's ONE real fixture: the example page
`data/examples/AisazuNihaIrarenai-003.jpg` (1654x1170, a double-page spread) and its published refined mask
`data/doc/AisazuNihaIrarenai-003-mask.png`.  Release weights are not available offline, so the network is
replaced by outputs derived from the PUBLISHED mask (letterboxed to the 1024x1024 network input: mask = the
published mask at 0.9, DB shrink map = the mask eroded towards line cores, yolo blocks = boxes of the dilated
mask's components); everything after the network is the reference's OWN `TextDetector.__call__`
(inference.py:141-178, run with the functional third-party stand-ins of oracle/ref_post_import.py).

Writes tests/golden/real_page.npz: the page and mask as their original JPEG / PNG bytes, the synthetic network
outputs, and the reference code's results (mask, refined mask, block records).  Build container only.

    python oracle/gen_golden_real.py
"""
import io
import json
import os
import sys

import numpy as np
import torch
from PIL import Image
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import annot_ref as A              # noqa: E402
from oracle import cv_ref as cv                # noqa: E402
from oracle import ref_post_import as RP       # noqa: E402

PAGE = "/root/reference/data/examples/AisazuNihaIrarenai-003.jpg"
MASK = "/root/reference/data/doc/AisazuNihaIrarenai-003-mask.png"
SIZE = 1024


def load_page(jpeg_bytes: bytes) -> np.ndarray:
    """BGR uint8 like cv2.imread."""
    return np.ascontiguousarray(np.array(Image.open(io.BytesIO(jpeg_bytes)).convert("RGB"))[:, :, ::-1])


def network_outputs(page: np.ndarray, published: np.ndarray):
    """Plausible (blks, mask, lines_map) of a trained detector for this page, from its published mask."""
    lb, _, (dw, dh) = cv.letterbox(np.repeat(published[:, :, None], 3, 2), (SIZE, SIZE))
    m = lb[:, :, 0] > 127
    mask = (m * 0.9).astype(np.float32)
    # DB shrink map: text-line cores = the mask closed along lines, at 0.85 / 0.05
    core = ndimage.binary_closing(m, structure=np.ones((5, 5)), iterations=2)
    core = ndimage.binary_opening(core, structure=np.ones((3, 3)))
    prob = np.where(core, 0.85, 0.05).astype(np.float32)
    big = ndimage.binary_dilation(m, structure=np.ones((3, 3)), iterations=8)
    lab, n = ndimage.label(big, structure=np.ones((3, 3)))
    rng = np.random.RandomState(3)
    blks = np.zeros((1, 4096, 7), np.float32)
    for i, sl in enumerate(ndimage.find_objects(lab)[:4096]):
        x1, y1, x2, y2 = sl[1].start, sl[0].start, sl[1].stop, sl[0].stop
        c, s = int(rng.randint(0, 2)), float(np.round(rng.uniform(0.5, 1), 3))
        blks[0, i] = [(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1, 0.99, 0.0, 0.0]
        blks[0, i, 5 + c] = s / 0.99
    return blks, mask[None, None], np.stack([prob, np.zeros_like(prob)])[None], (dw, dh)


def load_fixture(path: str):
    """tests/golden/real_page.npz -> (page BGR u8, blks, mask_u8, prob f32, (dw, dh), npz)."""
    g = np.load(path)
    page = load_page(g["jpeg"].tobytes())
    assert int(page.astype(np.int64).sum()) == int(g["page_checksum"][0]), "JPEG decoder differs from the generator's"
    core = np.unpackbits(g["core"])[: SIZE * SIZE].reshape(SIZE, SIZE).astype(bool)
    prob = np.where(core, 0.85, 0.05).astype(np.float32)
    return page, g["blks"], g["mask_u8"], prob, tuple(int(v) for v in g["dwdh"]), g


def main():
    jpeg, png = open(PAGE, "rb").read(), open(MASK, "rb").read()
    page = load_page(jpeg)
    published = np.array(Image.open(io.BytesIO(png)).convert("L"))
    blks, mask, lines_map, (dw, dh) = network_outputs(page, published)
    ref = RP.load_reference_post()
    out = {}
    for keep in (0, 1):
        det = RP.reference_detector(ref, (torch.from_numpy(blks.copy()), torch.from_numpy(mask.copy()),
                                          torch.from_numpy(lines_map.copy())), input_size=(SIZE, SIZE))
        m, r, b = det(page.copy(), refine_mode=keep, keep_undetected_mask=bool(keep))
        rec = json.dumps([t.to_dict() for t in b], ensure_ascii=False, cls=A.NumpyEncoder)
        out[f"mask{keep}"] = m
        out[f"refined{keep}"] = np.packbits(r > 0)
        out[f"records{keep}"] = np.frombuffer(rec.encode("utf8"), np.uint8)
        print("keep", keep, "blocks", len(b), "lines", sum(len(t.lines) for t in b), "refined frac", float((r > 0).mean()),
              "IoU vs the published mask", float(((r > 0) & (published > 0)).sum() / ((r > 0) | (published > 0)).sum()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "real_page.npz"), jpeg=np.frombuffer(jpeg, np.uint8),
                        png=np.frombuffer(png, np.uint8), blks=blks, mask_u8=(mask[0, 0] * 255).astype(np.uint8),
                        core=np.packbits(lines_map[0, 0] > 0.5), dwdh=np.array([dw, dh]), shape=np.array(page.shape[:2]),
                        page_checksum=np.array([int(page.astype(np.int64).sum())]), **out)


if __name__ == "__main__":
    main()
