#!/usr/bin/env python3
"""Supplementary measurements for DESIGN.md / profiles (not the headline bench):
  1. bs=1 forward latency, eager launches vs hipGraph replay
  2. host+GPU time of the detector tail per 1024x1024 page on text-like outputs
  3. config 2 (bs=8, exact-fp32 engine) forward time
  4. latency of one `TextDetector.__call__` on a host page (bs=1)"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("comic-text-detector_amd")
out = {}
ck = pkg.synth.make_checkpoint(0)
be = pkg.backend.HipTextDetBackend(ck, precision="fp16")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in (1, 4):
    x = torch.randint(0, 256, (B, 1024, 1024, 3), dtype=torch.uint8).cuda()
    out[f"fwd_ms_b{B}_eager"] = round(timeit(lambda: be.forward_u8(x)), 3)
    static_in, replay = be.capture(B, 1024, 1024, "u8")
    static_in.copy_(x)
    out[f"fwd_ms_b{B}_graph"] = round(timeit(replay), 3)

from test_post_host import fake_outputs          # noqa: E402
from test_gpu_e2e import blks_tensor             # noqa: E402
det = pkg.detector.TextDetector(ck, input_size=1024, device="cuda", half=True)
NP = int(os.environ.get("EXTRA_PAGES", "32"))
pages, args = [], []
for s in range(NP):
    page, mask_u8, prob, blks = fake_outputs(s % 8, 1024)
    pages.append(torch.from_numpy(page).cuda())
    args.append((blks_tensor(blks), mask_u8, prob, (prob > 0.3).astype(np.uint8)))
bt = torch.from_numpy(np.concatenate([a[0] for a in args])).cuda()
mu = torch.from_numpy(np.stack([a[1] for a in args])).cuda()
pr = torch.from_numpy(np.stack([a[2] for a in args])).cuda()
bm = torch.from_numpy(np.stack([a[3] for a in args])).cuda()
for keep in (False, True):
    det.tail_batch(pages, bt, mu, pr, bm, keep_undetected_mask=keep)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        res = det.tail_batch(pages, bt, mu, pr, bm, keep_undetected_mask=keep)
    out[f"tail_ms_per_page_textlike_1024_b{NP}_keep{int(keep)}"] = round((time.perf_counter() - t0) / 3 / NP * 1e3, 3)
out["tail_blocks_per_page"] = [len(r[2]) for r in res][:8]

# 4. latency mode: one numpy page in, `TextDetector.__call__` as the reference's callers use it (bs=1, host page,
#    blob checkpoint so that the maps have contours -> the tail does real work)
ckb = pkg.synth.make_blob_checkpoint(0)
page1 = pkg.synth.text_like_page((1024, 1024), 3, n_blocks=8)
import gc                                          # noqa: E402
for half in (True, False):
    d1 = pkg.detector.TextDetector(ckb, input_size=1024, device="cuda", half=half)
    key = f"call_ms_b1_host_page_{'fp16' if half else 'fp32'}"
    out[key + "_no_gc_freeze"] = round(timeit(lambda: d1(page1), n=10, warm=3), 3)
    gc.collect()
    gc.freeze()                                    # what a serving process does once after start-up
    out[key] = round(timeit(lambda: d1(page1), n=10, warm=3), 3)
    gc.unfreeze()
    del d1

be32 = pkg.backend.HipTextDetBackend(ck, precision="fp32")
x = torch.rand(8, 3, 1024, 1024).cuda()
out["config2_fp32_direct_ms_per_8_pages"] = round(timeit(lambda: be32(x), n=2, warm=1), 1)
print(json.dumps(out))
