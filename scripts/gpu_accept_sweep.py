"""The acceptance comparison of tests/test_gpu_accept.py over MANY pages (round 6): the product detector (HIP engine -> native tail)
against the oracle (CPU fp32 network -> oracle tail) end to end on text-like pages of seed after seed, benchmark checkpoint.
ACCEPT_N pages (default 40), ACCEPT_SIZE (512), ACCEPT_PREC (fp32s), ACCEPT_SEED0 (100).  Prints, per page, what differs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg          # noqa: E402
from oracle import accept, cv_ref as cv   # noqa: E402
from oracle import postproc_ref as R      # noqa: E402
from oracle.net_ref import OracleNet      # noqa: E402

p = pkg()
n = int(os.environ.get("ACCEPT_N", "40"))
size = int(os.environ.get("ACCEPT_SIZE", "512"))
prec = os.environ.get("ACCEPT_PREC", "fp32s")
seed0 = int(os.environ.get("ACCEPT_SEED0", "100"))
ck = p.synth.make_blob_checkpoint(0)
net = OracleNet(ck)
torch.set_num_threads(16)
det = p.detector.TextDetector(ck, input_size=size, device="cuda", precision=prec)
rng = np.random.RandomState(seed0)
full = lines_ok = blocks_ok = refined_ok = 0
n_lines = n_blocks = 0
worst_mask = 1.0
for i in range(n):
    shape = (size, size) if i % 3 else (int(rng.randint(300, 900)), int(rng.randint(300, 900)))
    page = p.synth.text_like_page(shape, seed0 + i, n_blocks=int(rng.randint(4, 12)))
    mode, keep = (0, False) if i % 2 == 0 else (1, True)
    got = det(page, refine_mode=mode, keep_undetected_mask=keep)
    lb, ratio, (dw, dh) = cv.letterbox(page, (size, size))
    x = torch.from_numpy(np.ascontiguousarray(lb.transpose(2, 0, 1)[None])).float() / 255
    blks, mask, lines_map = net(x)
    ref = R.detector_tail(page, blks.numpy(), mask.numpy(), lines_map.numpy(), input_size=(size, size), dw=dw, dh=dh,
                          refine_mode=mode, keep_undetected_mask=keep)
    rep = accept.compare(got, ref)
    lo = rep["lines"]["identical"] == rep["lines"]["ref"] == rep["lines"]["ours"]
    bo = rep["blocks"]["identical"] == rep["blocks"]["ref"] == rep["blocks"]["ours"]
    ro = rep["refined_mask_equal_frac"] == 1.0
    lines_ok += lo
    blocks_ok += bo
    refined_ok += ro
    full += lo and bo and ro
    n_lines += rep["lines"]["ref"]
    n_blocks += rep["blocks"]["ref"]
    worst_mask = min(worst_mask, rep["mask_u8_equal_frac"])
    if not (lo and bo and ro):
        print(f"page {seed0 + i} {shape} mode {mode} keep {keep}: lines {rep['lines']} blocks {rep['blocks']} refined equal "
              f"{rep['refined_mask_equal_frac']:.6f} mask equal {rep['mask_u8_equal_frac']:.6f}", flush=True)
print(f"acceptance sweep, engine {prec}, input {size}: {n} pages ({n_lines} lines, {n_blocks} blocks in the oracle's results): "
      f"lines identical on {lines_ok}, blocks on {blocks_ok}, refined mask on {refined_ok}, all three on {full}; "
      f"lowest share of equal u8 mask pixels {worst_mask:.6f}")
