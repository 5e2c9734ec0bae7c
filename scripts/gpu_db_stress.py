"""Randomised stress of the DB text-line stage (`postproc.SegRepresenter`: two GPU labelling passes + contour tables, hull /
min-area rectangle / Clipper unclip on the host) against the oracle's `boxes_from_bitmap` (contour walk + polygon fill;
reference utils/db_utils.py:123-211) on maps that are NOT text-like: smoothed noise at several scales (speckle, holes, islands
in holes), thresholded gradients, rotated bars, maps touching every border.  DB_STRESS_N cases (default 120), DB_STRESS_SEED."""
import os
import sys

import numpy as np
import torch
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg          # noqa: E402
from oracle import postproc_ref as R   # noqa: E402

p = pkg()
rep = p.postproc.SegRepresenter()
n_cases = int(os.environ.get("DB_STRESS_N", "120"))
rng = np.random.RandomState(int(os.environ.get("DB_STRESS_SEED", "1")))
bad = 0
for case in range(n_cases):
    H, W = int(rng.randint(24, 300)), int(rng.randint(24, 400))
    kind = case % 4
    if kind == 0:                                        # smoothed noise
        pr = ndimage.uniform_filter(rng.rand(H, W), int(rng.randint(1, 9)))
        pr = (pr - pr.min()) / max(pr.max() - pr.min(), 1e-9) * rng.uniform(0.4, 0.9)
    elif kind == 1:                                      # rotated bars
        pr = np.full((H, W), 0.05)
        yy, xx = np.mgrid[0:H, 0:W]
        for _ in range(rng.randint(1, 8)):
            ang = rng.uniform(0, np.pi)
            cx, cy, L, T = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(5, 120), rng.uniform(1, 14)
            u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
            v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
            pr[(np.abs(u) < L) & (np.abs(v) < T)] = rng.uniform(0.35, 0.99)
    elif kind == 2:                                      # blobs with holes and islands
        pr = np.full((H, W), 0.1)
        for _ in range(rng.randint(1, 6)):
            y, x, h, w = rng.randint(0, H), rng.randint(0, W), rng.randint(4, 80), rng.randint(4, 120)
            pr[y: y + h, x: x + w] = rng.uniform(0.5, 0.95)
            if h > 8 and w > 8:
                pr[y + 2: y + h - 2, x + 2: x + w - 2] = 0.1
                if h > 14 and w > 14:
                    pr[y + 5: y + h - 5, x + 5: x + w - 5] = rng.uniform(0.5, 0.95)
    else:                                                # gradient times noise
        pr = np.linspace(0, 1, W)[None, :] * np.linspace(0.2, 1, H)[:, None] * (0.6 + 0.4 * rng.rand(H, W))
    pr = pr.astype(np.float32)
    t = torch.from_numpy(pr)[None].cuda()
    boxes, scores = rep(t, (t > 0.3).to(torch.uint8))
    rb, rs = R.boxes_from_bitmap(pr, pr > 0.3, W, H)
    if len(boxes[0]) != len(rb) or not np.array_equal(boxes[0], rb) or not np.allclose(scores[0], rs, rtol=0, atol=1e-6):
        bad += 1
        nd = int((np.asarray(boxes[0]).reshape(len(rb), -1) != np.asarray(rb).reshape(len(rb), -1)).any(1).sum()) if len(boxes[0]) == len(rb) else -1
        print(f"case {case} kind {kind} {H}x{W}: {len(boxes[0])} vs {len(rb)} boxes, {nd} differ", flush=True)
print(f"db stress: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
