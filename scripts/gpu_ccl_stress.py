#!/usr/bin/env python3
"""Randomised stress of the labelling kernels against scipy (run-pruned border links are the delicate part:
tile corners, thin diagonal chains, dense noise).  Not part of the test-suite; `CCL_STRESS_N` cases."""
import importlib
import os
import sys

import numpy as np
import torch
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("comic-text-detector_amd")
BK = pkg.backend
rng = np.random.RandomState(int(os.environ.get("CCL_STRESS_SEED", "0")))
N = int(os.environ.get("CCL_STRESS_N", "120"))
S8, S4 = np.ones((3, 3), int), np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
bad = 0
for case in range(N):
    h, w = int(rng.randint(1, 300)), int(rng.randint(1, 300))
    kind = case % 4
    if kind == 0:
        img = rng.uniform(size=(h, w)) < rng.uniform(0.02, 0.98)
    elif kind == 1:                                   # diagonal / anti-diagonal chains crossing tile corners
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((yy + xx) % int(rng.randint(2, 7)) == 0) | ((yy - xx) % int(rng.randint(2, 9)) == 0)
        img &= rng.uniform(size=(h, w)) < 0.9
    elif kind == 2:                                   # blocks aligned to the 32-pixel tiles with random gaps
        img = np.ones((h, w), bool)
        img[::32] = rng.uniform(size=img[::32].shape) < 0.5
        img[:, ::32] = rng.uniform(size=img[:, ::32].shape) < 0.5
        img[31::32] = rng.uniform(size=img[31::32].shape) < 0.5
        img[:, 31::32] = rng.uniform(size=img[:, 31::32].shape) < 0.5
    else:                                             # smooth blobs
        img = ndimage.gaussian_filter(rng.uniform(size=(h, w)), rng.uniform(0.5, 3)) > 0.5
    u8 = torch.from_numpy(img.astype(np.uint8) * 255).cuda()
    for conn, st in ((8, S8), (4, S4)):
        lab, n, stats = BK.connected_components(u8, 0, conn, max_labels=1 << 17)
        ref, nref = ndimage.label(img, structure=st)
        if int(n[0]) != nref or not np.array_equal(lab[0].cpu().numpy(), ref):
            bad += 1
            print("MISMATCH ccl", case, kind, (h, w), conn, int(n[0]), nref)
    lab, (nf, nb), _, _ = BK.connected_components_dual(u8, 0, max_labels=1 << 17)
    l = lab[0].cpu().numpy()
    rf, nrf = ndimage.label(img, structure=S8)
    rb, nrb = ndimage.label(~img, structure=S4)
    if int(nf[0]) != nrf or int(nb[0]) != nrb or not np.array_equal(np.maximum(l, 0), rf) or not np.array_equal(np.maximum(-l, 0), rb):
        bad += 1
        print("MISMATCH dual", case, kind, (h, w), int(nf[0]), nrf, int(nb[0]), nrb)
print("ccl stress:", N, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
