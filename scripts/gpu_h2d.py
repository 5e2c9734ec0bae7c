#!/usr/bin/env python3
"""Does a pinned-host -> HBM copy of a batch of pages (100 MB) overlap the forward of another batch?"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("comic-text-detector_amd")
dev = torch.device("cuda", 0)
be = pkg.backend.HipTextDetBackend(pkg.synth.make_checkpoint(0), dev, precision="fp16")
pages = torch.randint(0, 256, (32, 1024, 1024, 3), dtype=torch.uint8, device=dev)
host = torch.empty((32 * 1024 * 1024 * 3,), dtype=torch.uint8).pin_memory()
cs = torch.cuda.Stream(dev)
N = 10


def t(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / N * 1e3, 3)


def copy():
    with torch.cuda.stream(cs):
        return host.to(dev, non_blocking=True)


def fwd():
    be.forward_u8(pages)


def both():
    copy()
    fwd()


out = {"h2d_100MB_ms": t(copy), "forward_ms": t(fwd), "both_ms": t(both)}
out["h2d_GBps"] = round(host.numel() / out["h2d_100MB_ms"] / 1e6, 1)
print(json.dumps(out))
