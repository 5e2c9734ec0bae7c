"""Calibration of `synth.make_blob_checkpoint(sparse_det=True)` (run in the build container; uses the oracle network).

The blob checkpoint's Detect head fires on (almost) every cell of ONE anchor (level 2, anchor 2: the confidences of
random weights sit in a 0.02-wide band), so NMS packs the page with ~65 boxes of ~160 px = 1.5 page areas of block
windows -- 4x the block count of the reference's only real fixture (data/examples/AisazuNihaIrarenai-003.jpg: 16
blocks, 29 lines; tests/golden/real_page.npz).  `sparse_det` amplifies that anchor's objectness logit around a
quantile q of its distribution, z' = G (z - q), so that only the cells above q fire.  This script prints, for a
range of quantiles, the boxes NMS keeps on text-like pages at 1024x1024; the constants chosen are in synth.py."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("comic-text-detector_amd")
from oracle.net_ref import OracleNet          # noqa: E402
from oracle import postproc_ref as R          # noqa: E402

if __name__ == "__main__":
    ck = pkg.synth.make_blob_checkpoint(0)
    torch.set_num_threads(8)
    seeds = (0, 1, 131, 262, 393)
    pages = [pkg.synth.text_like_page((1024, 1024), s) for s in seeds]
    net = OracleNet(ck)
    outs = [net(torch.from_numpy(np.ascontiguousarray(p.transpose(2, 0, 1)[None])).float() / 255)[0] for p in pages]
    lo = 3 * (128 * 128 + 64 * 64) + 2 * 32 * 32           # rows of level 2, anchor 2
    zs = []
    for b in outs:
        p = b[0, lo: lo + 1024, 4].double().clamp(1e-9, 1 - 1e-9)
        zs.append(torch.log(p / (1 - p)))
    z = torch.cat(zs)
    print("objectness logit of level 2 / anchor 2 over", len(seeds), "pages: mean %.4f std %.4f min %.4f max %.4f" %
          (z.mean(), z.std(), z.min(), z.max()))
    G = 40.0
    for frac in (0.5, 0.2, 0.1, 0.05, 0.035, 0.03, 0.025, 0.02, 0.015):
        q = float(torch.quantile(z, 1 - frac))
        res = []
        for b, zz in zip(outs, zs):
            bb = b.clone()
            bb[0, :, 4] = 0
            bb[0, lo: lo + 1024, 4] = torch.sigmoid(G * (zz - q)).float()
            dets = R.non_max_suppression(bb.numpy(), 0.4, 0.35)[0]
            area = sum((d[2] - d[0]) * (d[3] - d[1]) for d in dets) / 1024 ** 2
            res.append((len(dets), round(float(area), 2)))
        print(f"top {frac:.2f} of the cells (q = {q:.5f}): (boxes, summed box area / page) per page", res)
