#!/usr/bin/env python3
"""The native tail ALONE on the real chain's inputs (blob checkpoint at the bench's block density, 32 text-like 1024x1024 pages: the
outputs of one fp16 forward), as the three page-range work items of the pipeline run one after the other on one thread -- for
`rocprofv3 --kernel-trace --stats` (kernel times without the forward next to them) and the host stage timings."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("comic-text-detector_amd")
dev = torch.device("cuda", 0)
ck = pkg.synth.make_blob_checkpoint(0, sparse_det=not os.environ.get("DENSE"))
det = pkg.detector.TextDetector(ck, input_size=1024, device=dev, precision="fp16")
pages = [pkg.synth.text_like_page((1024, 1024), i) for i in range(32)]
x = torch.from_numpy(np.stack(pages)).to(dev)
job = det._forward([x[i] for i in range(32)])
torch.cuda.synchronize()
parts = det._split(32, int(os.environ.get("SPLIT", "3")))
for _ in range(3):
    for lo, hi in parts:
        det._tail(job, 0, False, lo, hi)
n = int(os.environ.get("TAIL_ITERS", "10"))
t0 = time.perf_counter()
for _ in range(n):
    for lo, hi in parts:
        res = det._tail(job, 0, False, lo, hi)
dt = (time.perf_counter() - t0) / n
tl = importlib.import_module("comic-text-detector_amd.tail").thread_tail(dev)
print(json.dumps({"tail_ms_per_batch_serial": round(dt * 1e3, 3), "iters": n, "work_items": len(parts),
                  "last_item_stages_ms": tl.timings()}))
