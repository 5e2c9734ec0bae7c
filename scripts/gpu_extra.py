#!/usr/bin/env python3
"""Supplementary measurements for DESIGN.md / profiles (not the headline bench):
  1. bs=1 forward latency, eager launches vs hipGraph replay
  2. host+GPU time of the detector tail per 1024x1024 page on text-like outputs
  3. config 2 (bs=8, exact-fp32 direct kernels) forward time"""
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
det = pkg.detector.TextDetector(ck, input_size=1024, device="cuda")
pages, args = [], []
for s in range(4):
    page, mask_u8, prob, blks = fake_outputs(s, 1024)
    pages.append(page)
    args.append((blks_tensor(blks), mask_u8, prob, (prob > 0.3).astype(np.uint8)))
bt = torch.from_numpy(np.concatenate([a[0] for a in args])).cuda()
mu = torch.from_numpy(np.stack([a[1] for a in args])).cuda()
pr = torch.from_numpy(np.stack([a[2] for a in args])).cuda()
bm = torch.from_numpy(np.stack([a[3] for a in args])).cuda()
det.tail_batch(pages, bt, mu, pr, bm, keep_undetected_mask=True)
t0 = time.perf_counter()
res = det.tail_batch(pages, bt, mu, pr, bm, keep_undetected_mask=True)
out["tail_ms_per_page_textlike_1024"] = round((time.perf_counter() - t0) / 4 * 1e3, 1)
out["tail_blocks_per_page"] = [len(r[2]) for r in res]

# per-stage split of the same tail (host wall clock, device synchronised at the stage ends)
from importlib import import_module                  # noqa: E402
PP = pkg.postproc
stage = {}


def clock(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    stage[name] = stage.get(name, 0.0) + (time.perf_counter() - t) * 1e3 / 4
    return r


ratios = [(1.0, 1.0)] * 4
yolo = clock("nms+unpack", lambda: PP.postprocess_yolo(bt, det.conf_thresh, det.nms_thresh, ratios))
boxes, scores = clock("db_boxes(ccl x2 + download + ctd_db_boxes)", lambda: det.seg_rep(pr, bm))
for b in range(4):
    lines = boxes[b][scores[b] > 0.6].astype(np.int32)
    m = clock("mask download", lambda: mu[b].cpu().numpy().copy())
    blk = clock("group_output", lambda: pkg.textblock.group_output(yolo[b], lines, 1024, 1024, m))
    ref = clock("refine_mask", lambda: pkg.textmask.refine_mask(pages[b], m, blk, 0, "cuda"))
    clock("refine_undetected_mask", lambda: pkg.textmask.refine_undetected_mask(pages[b], m, ref, blk, 0, "cuda"))
out["tail_stage_ms_per_page"] = {k: round(v, 2) for k, v in stage.items()}

be32 = pkg.backend.HipTextDetBackend(ck, precision="fp32")
x = torch.rand(8, 3, 1024, 1024).cuda()
out["config2_fp32_direct_ms_per_8_pages"] = round(timeit(lambda: be32(x), n=2, warm=1), 1)
print(json.dumps(out))
