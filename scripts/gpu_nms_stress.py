"""Randomised stress of the GPU NMS (`backend.nms`: ctd launch_nms) against the oracle's restatement of
`non_max_suppression` (reference utils/yolov5_utils.py:124-218): row counts 50 .. 70 000, candidate fractions 0 .. 1, box sizes
from dense overlap to sparse, duplicated boxes with equal scores.  NMS_STRESS_N cases (default 200), NMS_STRESS_SEED."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg          # noqa: E402
from oracle import postproc_ref as R   # noqa: E402

p = pkg()
n_cases = int(os.environ.get("NMS_STRESS_N", "200"))
rng = np.random.RandomState(int(os.environ.get("NMS_STRESS_SEED", "1")))
bad = 0
for case in range(n_cases):
    B = int(rng.randint(1, 4))
    rows = int(rng.choice([50, 300, 1008, 4032, 16128, 64512, int(rng.randint(50, 70000))]))
    frac = float(rng.choice([0.0, 0.01, 0.05, 0.3, 1.0, rng.uniform(0, 1)]))
    no = int(rng.choice([6, 7, 8]))
    size = int(rng.choice([256, 1024, 2048]))
    b = np.zeros((B, rows, no), np.float32)
    b[..., 0:2] = rng.uniform(0, size, (B, rows, 2))
    b[..., 2:4] = rng.uniform(2, rng.choice([20, 300, 900]), (B, rows, 2))
    b[..., 4] = np.where(rng.uniform(size=(B, rows)) < frac, rng.uniform(0.4, 1.0, (B, rows)), rng.uniform(0, 0.4, (B, rows)))
    b[..., 5:] = rng.uniform(0, 1, (B, rows, no - 5))
    if case % 5 == 0 and rows >= 64:                     # exact duplicates with equal scores: tie-break = lower row first
        k = rows // 8
        b[:, k: 2 * k] = b[:, :k]
    if case % 7 == 0:                                    # scores on a coarse grid: many equal confidences
        b[..., 4] = np.round(b[..., 4] * 20) / 20
    dets, counts = p.backend.nms(torch.from_numpy(b).cuda(), 0.4, 0.35)
    torch.cuda.synchronize()
    ref = R.non_max_suppression(b, 0.4, 0.35)
    for i in range(B):
        n = int(counts[i])
        if n != len(ref[i]) or not np.array_equal(dets[i, :n].cpu().numpy(), ref[i]):
            bad += 1
            print(f"case {case} page {i}: rows {rows} frac {frac:.3f} no {no}: {n} vs {len(ref[i])} detections", flush=True)
            break
print(f"nms stress: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
