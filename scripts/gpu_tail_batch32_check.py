"""One native tail call on 32 DIFFERENT pages (text-like outputs of 32 seeds, tests/test_post_host.py `fake_outputs`) against the oracle's
tail page by page (round 6): the batch-wide GPU stages and the one-thread-per-page host stages at the benchmark's batch size.
TB32_SEED0 (default 7000), TB32_SIZE (512), TB32_ROUNDS (2: one per tail configuration)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg          # noqa: E402
from oracle import postproc_ref as R   # noqa: E402
from test_post_host import blocks_equal, fake_outputs   # noqa: E402
from test_gpu_e2e import blks_tensor, detector          # noqa: E402

p = pkg()
seed0 = int(os.environ.get("TB32_SEED0", "7000"))
size = int(os.environ.get("TB32_SIZE", "512"))
det = detector(size)
bad = 0
for rnd in range(int(os.environ.get("TB32_ROUNDS", "2"))):
    keep = bool(rnd & 1)
    cases = [fake_outputs(seed0 + 32 * rnd + i, size) for i in range(32)]
    rows = max(len(c[3][0]) for c in cases) + 8
    bts = [blks_tensor(c[3], rows=max(rows, 64)) for c in cases]
    blks = torch.from_numpy(np.concatenate(bts)).cuda()
    mask = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    prob = torch.from_numpy(np.stack([c[2] for c in cases])).cuda()
    bitmap = (prob > 0.3).to(torch.uint8)
    got = det.tail_batch([c[0] for c in cases], blks, mask, prob, bitmap, refine_mode=1 if keep else 0, keep_undetected_mask=keep)
    for i, (page, mask_u8, pr, b) in enumerate(cases):
        mask_f = (mask_u8.astype(np.float32) + 0.5) / 255
        ref = R.detector_tail(page, bts[i], mask_f[None, None], np.stack([pr, np.zeros_like(pr)])[None], input_size=(size, size),
                              refine_mode=1 if keep else 0, keep_undetected_mask=keep)
        try:
            np.testing.assert_array_equal(got[i][0], ref[0])
            blocks_equal(got[i][2], ref[2])
            np.testing.assert_array_equal(got[i][1], ref[1])
        except AssertionError as e:
            bad += 1
            print(f"round {rnd} page {i} (seed {seed0 + 32 * rnd + i}): {str(e)[:160]}", flush=True)
print(f"tail on 32 different pages per call, {int(os.environ.get('TB32_ROUNDS', '2'))} calls at {size}: {bad} mismatching pages")
sys.exit(1 if bad else 0)
