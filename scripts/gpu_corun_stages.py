"""Which part of the tail costs the network its speed?  The network-only step (forward + GPU NMS, as `bench.py --mode net`)
is timed alone and next to ONE tail stage looping on a background thread (own `Tail`, own stream) over the same batch:
  db      ctd_tail_db_boxes: dual labelling + contour tables + host geometry (stage 1 without NMS / mask copies)
  refine  ctd_tail_refine: window histograms, xor sums, render, labelling, merge rounds, dilation, labelling, hole filling,
          commit, downloads of the refined pages
  d2h     a 32 MB device -> pinned host copy (hipMemcpyAsync) per iteration
  whole   the whole tail (Tail.run) on the forward's outputs
Prints the forward's ms per step and the background loop's iterations per second: stretch per tail iteration =
(ms with - ms alone) * forward steps per background iteration.   usage: python scripts/gpu_corun_stages.py [steps]"""
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("comic-text-detector_amd")
DET = importlib.import_module("comic-text-detector_amd.detector")
BK = importlib.import_module("comic-text-detector_amd.backend")
TL = importlib.import_module("comic-text-detector_amd.tail")

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
S, B = 1024, 32
ck = pkg.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
det = DET.TextDetector(ck, input_size=S, device=dev, precision="fp16")
pages = np.stack([pkg.synth.text_like_page((S, S), i) for i in range(B)])
x = torch.from_numpy(pages).to(dev)
pg = [x[i] for i in range(B)]
job = det._forward(pg)
torch.cuda.synchronize()
res = det._tail(job, 0, False)
boxes = [[b.xyxy for b in r[2]] for r in res]
masks = [r[0].copy() for r in res]
prob = job["lines_map"][:, 0].contiguous()
bitmap = job["bitmap"].clone()
blks, mask_u8, lines_map = job["blks"].clone(), job["mask_u8"].clone(), job["lines_map"].clone()
metas = job["metas"]
pin = torch.empty((32 << 20,), dtype=torch.uint8).pin_memory()
src = torch.empty((32 << 20,), dtype=torch.uint8, device=dev)
cstream = torch.cuda.Stream(dev)


def forward_ms(n):
    for _ in range(5):
        b_, _, _ = det.net.forward_u8(x)
        BK.nms(b_, 0.4, 0.35)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        b_, _, _ = det.net.forward_u8(x)
        BK.nms(b_, 0.4, 0.35)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def background(kind, stop, count, nthreads):
    tail = TL.Tail(dev)
    lo = 0
    while not stop.is_set():
        if kind == "db":
            tail.db_boxes(prob, bitmap)
        elif kind == "refine":
            tail.refine(pg, masks, boxes, 0, False)
        elif kind == "d2h":
            with torch.cuda.stream(cstream):
                pin.copy_(src, non_blocking=True)
            cstream.synchronize()
        elif kind == "whole":
            tail.run(pg, metas, blks, mask_u8, lines_map, bitmap, 0.4, 0.35, 0.6, True, 0, False, None, lazy=True)
        count[0] += 1


alone = forward_ms(steps)
print(f"network + NMS alone: {alone:.3f} ms per step")
for kind, nthr in (("db", 1), ("refine", 1), ("d2h", 1), ("whole", 1), ("db", 3), ("refine", 3), ("whole", 3)):
    stop, counts = threading.Event(), [[0] for _ in range(nthr)]
    th = [threading.Thread(target=background, args=(kind, stop, c, nthr)) for c in counts]
    for t in th:
        t.start()
    time.sleep(0.5)
    c0, t0 = sum(c[0] for c in counts), time.perf_counter()
    ms = forward_ms(steps)
    dt, c1 = time.perf_counter() - t0, sum(c[0] for c in counts)
    stop.set()
    for t in th:
        t.join()
    rate = (c1 - c0) / dt
    per_iter = (ms - alone) * (1e3 / ms) / max(rate, 1e-9)
    print(f"next to {nthr} x {kind:7s}: {ms:.3f} ms per step (+{ms - alone:.3f}); background {rate:.1f} batches of 32 per s "
          f"({1e3 / max(rate, 1e-9):.2f} ms each) -> {per_iter:.3f} ms of network time per background batch")
