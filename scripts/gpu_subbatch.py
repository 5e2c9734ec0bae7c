#!/usr/bin/env python3
"""Does a 32-page forward run faster as 32 / b forwards of b pages (depth first: a sub-batch's intermediates are 1 / (32 / b)
the size -- inside the 256-MB Infinity Cache for small b)?  Prints ms per 32 pages for b = 4, 8, 16, 32 and both engines."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pkg = importlib.import_module("comic-text-detector_amd")
BK = importlib.import_module("comic-text-detector_amd.backend")


def main():
    S, B = 1024, 32
    dev = torch.device("cuda", 0)
    ck = pkg.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
    xs = [torch.from_numpy(np.stack([pkg.synth.text_like_page((S, S), 131 * k + i) for i in range(B)])).to(dev) for k in range(2)]
    for prec in sys.argv[1:] or ["fp16", "fp32s"]:
        be = BK.HipTextDetBackend(ck, device=dev, precision=prec)
        for _ in range(30):
            be.forward_u8(xs[0])
        torch.cuda.synchronize()
        for rep in range(2):
            for b in (32, 16, 8, 4):
                for _ in range(3):
                    for lo in range(0, B, b):
                        be.forward_u8(xs[0][lo: lo + b])
                torch.cuda.synchronize()
                n = 20
                t0 = time.perf_counter()
                for k in range(n):
                    x = xs[k & 1]
                    for lo in range(0, B, b):
                        be.forward_u8(x[lo: lo + b])
                torch.cuda.synchronize()
                print(f"{prec} sub-batch {b:2d}: {(time.perf_counter() - t0) / n * 1e3:7.3f} ms per 32 pages", flush=True)


if __name__ == "__main__":
    main()
