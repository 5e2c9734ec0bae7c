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
        rates, tails = [r for r, _ in passes], [n for _, n in passes]
        last = rates[-1]
        if last < 0.90 * rates[0]:
            # the cliff is permanent, a busy host is not (the boxes' 256-CPU hosts are shared: one run in this round read the
            # host stages 1.3-2x slower for a few hundred ms, profiles/r06_cpu_quota.txt): one more pass tells them apart
            again = one_pass(p, ck, batches, other)
            print(f"\nsoak: last pass {round(last)} against {round(rates[0])}: one more pass {round(again[0])}")
            last = max(last, again[0])
    finally:
        gc.unfreeze()
    print(f"\nsoak: pages/s per pass {[round(r) for r in rates]}; live native tails after each pass {tails}")
    # one-sided: the failure mode is a process that got slower -- the hardware-queue cliff this guards against is -20 % and
    # permanent.  Passes of 0.6 s each scatter by up to 5 % on these boxes (measured: 2158 / 2055 / 2219 / 2199 / 2133), so the
    # bars sit between the noise and the cliff
    assert last >= 0.90 * rates[0], rates
    assert sorted(rates[1:])[1] >= 0.88 * rates[0], rates          # all but one of the later passes (one may meet a busy host)
    # reused, not accumulated: the count stops growing once a pool's worth exists (a pass may start before the previous
    # pool's thread-local leases have been returned, so the plateau can be up to two pools + the main thread's)
    assert tails[-1] <= tails[1] and tails[-1] <= 2 * WORKERS + 2, tails
    other.close(drain=True)


def test_bare_api_consumer_runs_at_the_bench_pipeline_rate():
    """the headline must be an API number.  `bench.Pipeline` now only feeds `TextDetector.detect_stream` and
    counts what comes back; this test is the other half -- a caller that knows nothing of bench.py (`for res in
    det.detect_stream(batches)`, every argument at its default: workers from the product's thread budget, `serve_tuning`
    applied by the generator itself) gets the rate the bench's pipeline gets (medians of five interleaved runs)."""
    import importlib
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    p = pkg()
    ck = p.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
    dev = torch.device("cuda", 0)
    xs = [torch.from_numpy(np.stack([p.synth.text_like_page((SIZE, SIZE), 131 * k + i) for i in range(B)])).to(dev) for k in range(2)]
    det = p.detector.TextDetector(ck, input_size=SIZE, device=dev, precision="fp16")
    tb = p.detector.thread_budget()
    pipe = bench.Pipeline(det, xs, None, dev, 1, 0, B, p.dist, tb["tail_workers"], 4, tb["tail_workers"])
    assert pipe.api

    def bare(n):
        k = 0
        for res in det.detect_stream([x[j] for j in range(B)] for x in (xs[i & 1] for i in range(n))):
            k += sum(len(r[2]) for r in res)
        return k

    def rate(fn, n=40):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return n * B / (time.perf_counter() - t0)

    bare(30), pipe.run(30)                                # clocks, pools, tails' buffers, serve_tuning (inside detect_stream)
    api, ref = [], []
    for _ in range(5):
        api.append(rate(bare))
        ref.append(rate(pipe.run))
    print(f"\nbare detect_stream consumer {[round(r) for r in api]} pages/s; bench.Pipeline {[round(r) for r in ref]}")
    assert pipe.stats["blocks"] > 0
    # Both loops ARE `detect_stream` (asserted above), so what is compared is two samples of one distribution: alone in a
    # process 0.5-s passes agree within 1 % (2936 / 2920 / 2922 vs 2931 / 2915 / 2922, profiles/r06_pytest_rccl_and_api_rate.txt),
    # late in the full suite (other detectors alive, warm allocator) they scatter by +-5 % -- medians of five, and the bar
    # of the soak test above (10 %: between that noise and the -20 % cliff of a mis-tuned process)
    assert float(np.median(api)) >= 0.90 * float(np.median(ref)), (api, ref)
    det.close(drain=True)
    gc.unfreeze()
