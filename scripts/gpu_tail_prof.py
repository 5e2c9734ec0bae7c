#!/usr/bin/env python3
"""The native tail alone on 32 text-like 1024x1024 pages (for rocprofv3 --kernel-trace --stats and stage timings)."""
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
T = importlib.import_module("comic-text-detector_amd.tail")
NP, NS, S = int(os.environ.get("TAIL_PAGES", "32")), 8, 1024
samples = [pkg.synth.text_like_outputs(s, S) for s in range(NS)]
dev = torch.device("cuda", 0)
pages = [torch.from_numpy(samples[i % NS][0]).to(dev) for i in range(NP)]
blks = torch.from_numpy(np.concatenate([samples[i % NS][1] for i in range(NP)])).to(dev)
mask = torch.from_numpy(np.stack([samples[i % NS][2] for i in range(NP)])).to(dev)
prob = torch.from_numpy(np.stack([samples[i % NS][3] for i in range(NP)])).to(dev)
bitm = torch.from_numpy(np.stack([samples[i % NS][4] for i in range(NP)])).to(dev)
metas = [(S, S, 0, 0)] * NP
tail = T.thread_tail(dev)
torch.cuda.synchronize()
keep = bool(int(os.environ.get("TAIL_KEEP", "0")))
for _ in range(2):
    tail.run(pages, metas, blks, mask, prob, bitm, keep_undetected_mask=keep)
if os.environ.get("GC_FREEZE"):
    import gc
    gc.collect()
    gc.freeze()          # startup objects (torch, numpy modules) leave the collector's generations
n = int(os.environ.get("TAIL_ITERS", "5"))
t0 = time.perf_counter()
for _ in range(n):
    res = tail.run(pages, metas, blks, mask, prob, bitm, keep_undetected_mask=keep)
dt = (time.perf_counter() - t0) / n
if os.environ.get("PROFILE"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        tail.run(pages, metas, blks, mask, prob, bitm, keep_undetected_mask=keep)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
area = sum(max(0, b.xyxy[2] - b.xyxy[0]) * max(0, b.xyxy[3] - b.xyxy[1]) for r in res for b in r[2])
print(json.dumps({"block_area_over_page_area": round(float(area) / (NP * S * S), 3),
                  "tail_ms_per_batch": round(dt * 1e3, 3), "ms_per_page": round(dt * 1e3 / NP, 4), "stages": tail.timings(),
                  "blocks": sum(len(r[2]) for r in res), "lines": sum(len(b.lines) for r in res for b in r[2])}))
