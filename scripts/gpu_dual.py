#!/usr/bin/env python3
"""Experiment: do two engines on two HIP streams overlap each other's MFMA-bound and HBM-bound layers?
Same total work (pages), one engine on one stream vs two engines on two streams."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("comic-text-detector_amd")
B = int(os.environ.get("DUAL_B", "32"))
N = int(os.environ.get("DUAL_N", "20"))
dev = torch.device("cuda", 0)
ck = pkg.synth.make_checkpoint(0)
be = [pkg.backend.HipTextDetBackend(ck, dev, precision="fp16") for _ in range(2)]
st = [torch.cuda.Stream(dev) for _ in range(2)]
pages = torch.randint(0, 256, (B, 1024, 1024, 3), dtype=torch.uint8, device=dev)
half = pages[: B // 2].contiguous()


def run(n_eng, batch, n):
    for i in range(2 * n_eng):                       # warm-up / planning
        with torch.cuda.stream(st[i % n_eng]):
            be[i % n_eng].forward_u8(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        with torch.cuda.stream(st[i % n_eng]):
            be[i % n_eng].forward_u8(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {"one_engine_bs%d_ms" % B: round(run(1, pages, N), 3),
       "two_engines_bs%d_ms_per_forward" % B: round(run(2, pages, N), 3),
       "one_engine_bs%d_ms" % (B // 2): round(run(1, half, 2 * N), 3),
       "two_engines_bs%d_ms_per_forward" % (B // 2): round(run(2, half, 2 * N), 3)}
print(json.dumps(out))
