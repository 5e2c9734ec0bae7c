"""Randomised checks without a test file of their own (round 6): (1) `backend.resize_linear_u8` (ctd_resize_linear_u8) against the
oracle's restatement of cv2.resize(INTER_LINEAR) on random source / destination shapes, 1 and 3 channels, with and without a
canvas; (2) `TextDetector.detect_batch` on pages of DIFFERENT sizes in one batch against the same pages one at a time (letterbox
per page, one forward, per-page inverse mapping) -- masks, refined masks and blocks identical.  RB_STRESS_N, RB_STRESS_SEED."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg, checkpoint          # noqa: E402
from oracle import cv_ref as cv             # noqa: E402
from test_post_host import blocks_equal     # noqa: E402

p = pkg()
n_cases = int(os.environ.get("RB_STRESS_N", "150"))
rng = np.random.RandomState(int(os.environ.get("RB_STRESS_SEED", "1")))
bad = 0
for case in range(n_cases):
    sh, sw = int(rng.randint(1, 700)), int(rng.randint(1, 700))
    dh, dw = int(rng.randint(1, 900)), int(rng.randint(1, 900))
    ch = 3 if case % 2 else 1
    img = rng.randint(0, 256, (sh, sw, 3) if ch == 3 else (sh, sw)).astype(np.uint8)
    got = p.backend.resize_linear_u8(torch.from_numpy(img).cuda(), (dh, dw)).cpu().numpy()
    ref = cv.resize_linear_u8(img, (dw, dh))
    if not np.array_equal(got, ref):
        bad += 1
        print(f"resize case {case}: {sh}x{sw}x{ch} -> {dh}x{dw}: {int((got != ref).sum())} differing values", flush=True)
print(f"resize stress: {n_cases} cases, {bad} mismatches")
size = 256
det = p.detector.TextDetector(checkpoint(), input_size=size, device="cuda", precision="fp32s")
bad2 = 0
for case in range(max(4, n_cases // 10)):
    k = int(rng.randint(2, 7))
    pages = [p.synth.text_like_page((int(rng.randint(160, 700)), int(rng.randint(160, 700))), 500 + 10 * case + i, n_blocks=int(rng.randint(2, 7)))
             for i in range(k)]
    mode, keep = int(rng.randint(0, 2)), bool(rng.randint(0, 2))
    batch = det.detect_batch(pages, mode, keep)
    for i, (pg, (m, r, bl)) in enumerate(zip(pages, batch)):
        m1, r1, bl1 = det(pg, mode, keep)
        try:
            np.testing.assert_array_equal(m, m1)
            np.testing.assert_array_equal(r, r1)
            blocks_equal(bl, bl1)
        except AssertionError as e:
            bad2 += 1
            print(f"batch case {case} page {i} {pg.shape}: {str(e)[:120]}", flush=True)
print(f"mixed-size batch against single calls: {max(4, n_cases // 10)} batches, {bad2} mismatching pages")
sys.exit(1 if bad or bad2 else 0)
