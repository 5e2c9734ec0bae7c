"""-m gpu: a long-lived server process.  `tail._Lease` exists because a process that has owned more compute streams than
the runtime has hardware queues for stays ~20 % slow for good (found with scripts/gpu_inprocess.py by hand).
This test is the guard: a detector is created, streams 50 batches through `detect_stream`, is closed -- five times over,
with a SECOND detector alive in the process and used in between -- and the last pass must run as fast as the first; the
process must not accumulate native tails (streams) either."""
import gc
import time

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu

B, SIZE, STEPS, WORKERS = 32, 1024, 50, 3


def one_pass(p, ck, batches, other):
    det = p.detector.TextDetector(ck, input_size=SIZE, device="cuda", precision="fp16")
    it = det.detect_stream((batches[k % len(batches)] for k in range(STEPS + 10)), workers=WORKERS, depth=4)
    n_blocks = 0
    for _ in range(10):                                   # warm-up: pools, arenas, tails' buffers
        n_blocks += sum(len(r[2]) for r in next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for res in it:
        n_blocks += sum(len(r[2]) for r in res)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    other.detect_batch(batches[0][:2])                    # the second detector stays in use between the passes
    det.close()
    del det
    gc.collect()
    assert n_blocks > 0
    return STEPS * B / dt, p.tail.live_tails()


def test_rate_survives_repeated_create_stream_close_with_two_detectors():
    p = pkg()
    ck = p.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
    dev = torch.device("cuda", 0)
    batches = []
    for k in range(2):
        x = torch.from_numpy(np.stack([p.synth.text_like_page((SIZE, SIZE), 131 * k + i) for i in range(B)])).to(dev)
        batches.append([x[i] for i in range(B)])
    other = p.detector.TextDetector(ck, input_size=SIZE, device=dev, precision="fp16")
    other.detect_batch(batches[0][:2])
    one_pass(p, ck, batches, other)                       # untimed: clocks, allocators, the first pools
    # what a serving process does once after start-up (bench.py `timed`): without it the cyclic collector re-scans the
    # ~1M long-lived objects of torch / numpy whenever the per-page results trigger a full collection (~10 ms a batch)
    gc.collect()
    gc.freeze()
    try:
        passes = [one_pass(p, ck, batches, other) for _ in range(5)]
    finally:
        gc.unfreeze()
    rates, tails = [r for r, _ in passes], [n for _, n in passes]
    print(f"\nsoak: pages/s per pass {[round(r) for r in rates]}; live native tails after each pass {tails}")
    # one-sided: the failure mode is a process that got slower -- the hardware-queue cliff this guards against is -20 % and
    # permanent.  Passes of 0.6 s each scatter by up to 5 % on these boxes (measured: 2158 / 2055 / 2219 / 2199 / 2133), so the
    # bars sit between the noise and the cliff
    assert rates[-1] >= 0.90 * rates[0], rates
    assert min(rates[1:]) >= 0.88 * rates[0], rates
    # reused, not accumulated: the count stops growing once a pool's worth exists (a pass may start before the previous
    # pool's thread-local leases have been returned, so the plateau can be up to two pools + the main thread's)
    assert tails[-1] <= tails[1] and tails[-1] <= 2 * WORKERS + 2, tails
    other.close(drain=True)
