#!/usr/bin/env python3
"""Headline benchmark: END-TO-END detector pages/sec at 1024x1024, bs=32 per GPU (BASELINE.json configs[2]: fp16
operands, fp32 accumulate), one process per GPU.

A "step" = one pass of the hot path (reference inference.py:141-178, `TextDetector.__call__`, batched) over one batch
of 32 synthetic pages already resident in HBM -- the REAL chain, every stage consuming the previous one's output:

    fused CNN forward (u8 pages -> backbone + Detect + UNet + DB heads, fused sigmoid / u8-mask / DB-binarize epilogues)
    -> native tail (`ctd_tail_run`) ON THAT FORWARD'S OWN blks / mask_u8 / lines_map / bitmap: GPU NMS, 2x GPU labelling
       + contour tables, host hull / min-area rectangle / unclip, mask crop, group_output, refine_mask (GPU candidates,
       labelling, merge rounds, hole filling), masks and TextBlock records back on the host
    -> (N>1) RCCL all-gather of the fixed-capacity per-page block records.

Release weights are not available offline; the default workload uses `synth.make_blob_checkpoint` (random weights in the
reference's checkpoint format whose maps have contours, boxes above the score threshold, lines and blocks) on
`synth.text_like_page` pages, `--batches` (default 4) DISTINCT batches rotated so the input is not Infinity-Cache
resident.  `--tail-input canned` restores round 2's workload (random checkpoint, tail fed text-like maps).

The tail of step k runs on worker threads (own HIP streams; one page-range work item per worker) under the forward of
step k+1.  Before the W warm-up steps the process runs `--spinup` untimed steps (board out of its low-power state, host
buffers settled; `config.spinup_steps`); the timed region is exactly K steps between barriers + synchronisations.

The same JSON line carries, from short sub-runs on rank 0 at N=1 (skipped with --no-extras; in this process, reusing the
headline's native tails -- see `inproc_bench` -- except the mixed-size stream and the MIOpen baseline, which are children):
  parity_exact   the EXACT engine (fp32s: fp32 tensors, split-operand products; identical lines / blocks / refined mask
                 to the oracle) end to end at the same batch size -- the rate the parity claim refers to
  extra_configs  BASELINE configs[1] (fp32 bs=8, end to end) and configs[4] (mixed 640/1024/1536 stream, hipGraph per
                 bucket, native tail); the headline on round 3's pages (16 lines per page), on the dense-block pages, on
                 round 2's canned tail inputs, with eagerly built TextBlocks, and with the pages starting in host memory
  rocm_baseline  the reference's own torch network on this GPU through PyTorch-ROCm / MIOpen
  cpu_baseline   the oracle (CPU fp32 port of the reference forward + the oracle tail) on the host cores

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8            (re-executes itself under torch.distributed.run when WORLD_SIZE is unset)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is measured live with hipEvents around every op of the engine
(ctd_engine_profile, on the stream the kernels run on).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32
# fp32s: every product costs three fp16 MFMAs -> a third of the fp16 peak for the ALGORITHMIC flops
PEAK_TF = {"fp16": MFMA_F16_PEAK_TFLOPS, "fp32": MFMA_F32_PEAK_TFLOPS, "fp32s": MFMA_F16_PEAK_TFLOPS / 3}
DTYPE = {"fp16": "f16", "fp32": "f32", "fp32s": "f32"}
FAMILY = {"fp16": "conv_halo3_kernel + conv_halo_kernel + conv_igemm_kernel + c3_fused_kernel + c3b_kernel (MFMA conv / convT family)",
          "fp32": "conv_f32_mfma_kernel (f32-operand MFMA conv / convT family)",
          "fp32s": "conv_split_kernel + conv_split_halo_kernel + stem_split_kernel (split-operand conv / convT family: 3 fp16 MFMAs "
                   "per product; conv-to-conv tensors split-plane in HBM)"}


def host_info() -> dict:
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the container's CFS quota (a 1-GPU box of this pool: 16 CPUs under an affinity mask of 256): threads beyond it are
    # throttled, not run -- the CPU legs size themselves on it
    quota = importlib.import_module("comic-text-detector_amd.detector").cgroup_cpu_quota()
    out = {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cpus": min(avail, int(quota)) if quota >= 1 else avail}
    if quota > 0:
        out["affinity_cpus"], out["cgroup_cpu_quota"] = avail, round(quota, 2)
    return out


def thread_budget(world: int, pinned: bool = False, host_cpus: int = 0) -> dict:
    """The product's own budget (`detector.thread_budget`: what `detect_stream(workers=0)` uses)."""
    DET = importlib.import_module("comic-text-detector_amd.detector")
    return DET.thread_budget(world, pinned, host_cpus)


# =====================================================================================================================
# workload
# =====================================================================================================================

def blob_checkpoint(pkg, args):
    """The benchmark's checkpoint: random weights whose maps have contours (synth.make_blob_checkpoint).  Default (round 4):
    the LINE density of the reference's one real page (~29 lines per page; ~27 blocks, see synth._FIXTURE_DET_Q);
    `--line-density r3`: round 3's pages (16 blocks / 16 lines); `--dense-blocks`: every cell of one Detect anchor fires."""
    if args.dense_blocks:
        return pkg.synth.make_blob_checkpoint(0)
    return pkg.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture" if args.line_density == "fixture" else None)


def make_workload(pkg, args, rank: int, nloc: int, dev):
    """Returns (checkpoint, batches, canned, canned_sample): `batches` = list of (B,H,W,3) u8 tensors resident in HBM
    (distinct pages), `canned` = None or the text-like network outputs of batch 0's pages (`--tail-input canned`)."""
    S = args.size
    if args.tail_input == "canned":
        ckpt = pkg.synth.make_checkpoint(0)
        NS = 8
        samples = [pkg.synth.text_like_outputs(100 * rank + s, S) for s in range(NS)]
        x = torch.from_numpy(np.stack([samples[i % NS][0] for i in range(nloc)])).to(dev)
        canned = dict(
            blks=torch.from_numpy(np.concatenate([samples[i % NS][1] for i in range(nloc)])).to(dev),
            mask_u8=torch.from_numpy(np.stack([samples[i % NS][2] for i in range(nloc)])).to(dev),
            lines_map=torch.from_numpy(np.stack([samples[i % NS][3] for i in range(nloc)])).to(dev),
            bitmap=torch.from_numpy(np.stack([samples[i % NS][4] for i in range(nloc)])).to(dev))
        return ckpt, [x], canned, samples[0]
    ckpt = blob_checkpoint(pkg, args)
    nb = max(1, args.batches)
    batches = []
    for k in range(nb):
        pages = [pkg.synth.text_like_page((S, S), 10007 * rank + 131 * k + i) for i in range(nloc)]
        batches.append(torch.from_numpy(np.stack(pages)).to(dev))
    return ckpt, batches, None, None


class Pipeline:
    """`TextDetector.detect_stream`'s pipeline with the benchmark's bookkeeping: forward of step k+1 on the main thread
    while worker threads run the tail work items of step k; (N>1) the record gather on its own stream."""

    def __init__(self, det, batches, canned, dev, world, rank, total_pages, D, workers, depth, tail_split,
                 host_input=False, loaders=2, engines=1, keep_undetected=False, lazy=False, refine_mode=0, force_dist=False):
        self.det, self.batches, self.canned, self.dev = det, batches, canned, dev
        self.world, self.rank, self.total_pages, self.D = world, rank, total_pages, D
        self.workers, self.depth, self.tail_split = max(1, workers), max(1, depth), max(1, tail_split)
        self.engines, self.keep_undetected, self.lazy = max(1, engines), keep_undetected, lazy
        self.refine_mode = int(refine_mode)
        # the record gather runs for N > 1 -- and for `--force-dist` at N = 1 (a world-size-1 group: the same code on RCCL)
        self.gather = world > 1 or bool(force_dist)
        self.nloc = batches[0].shape[0]
        # the detector's own pools (`detect_stream` keeps them between calls): the tails' streams before any loader stream
        self.pool = det._pool("tail", self.workers)
        det.warm_tails(self.pool, self.workers)
        det._warmed = self.pool
        self.loaders = max(1, loaders)
        self.host_batches = [[p for p in b.cpu().numpy()] for b in batches] if host_input else None
        self.lpool = det._pool("load", self.loaders) if host_input else None
        self.comm_stream = torch.cuda.Stream(dev) if self.gather else None      # the record gather's stream
        self.stats = {"blocks": 0, "lines": 0, "pages": 0, "cpu_cores": 0.0, "gather_s": 0.0}
        self.tfin = deque(maxlen=4096)                       # when each batch's results came back (perf_counter): the line's jitter evidence
        self.k = 0                                           # batches rotate across calls too
        self.fixed_job = None
        self.fwd_stream = None
        # N > 1: the tail builds every page's gather record natively (dist.pack_results then only stacks them)
        self.records = (D.CAP_BLK, D.CAP_LINE) if self.gather else None
        self.gathered = None                                 # the last gathered record tensor (checked by the caller)
        self.gathers = deque()                               # record gathers handed to the communication thread (futures)
        self.handles = deque()                               # ... in flight on the device (dist.GatherHandle), that thread's
        self.cpool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="ctd-comm") if self.gather else None

    def close(self):
        self.det.close()                                     # ends the detector's worker / loader pools (idempotent)
        if self.cpool is not None:
            self.cpool.shutdown(wait=True)
            self.cpool = None

    @property
    def api(self):
        """True when the step is exactly `TextDetector.detect_stream` (everything but the measurement variants that
        replace a stage: canned tail inputs, `--tail-only`)."""
        return self.canned is None and self.fixed_job is None

    def forward_job(self, i, pg=None):
        if self.fixed_job is not None:                       # --tail-only: no forward; every step's tail reads the same outputs
            return dict(self.fixed_job)
        if pg is None:
            x = self.batches[i % len(self.batches)]
            pg = [x[j] for j in range(x.shape[0])]           # slices of one batch tensor: no torch.stack in the detector
        if self.engines > 1:
            net, st = self.det._lane(i % self.engines)
            with torch.cuda.stream(st):
                job = self.det._forward(pg, net)
        else:
            job = self.det._forward(pg)
        if self.canned is not None:
            job.update(self.canned)                          # --tail-input canned: text-like maps instead of the forward's
        return job

    def finish(self, res):
        if self.gather:
            # on its own stream: the default stream holds the queued forwards of the next batches, and an upload or a
            # collective enqueued behind them would stall this thread until they have run
            # enqueued, not waited for: the counts that decide about the (rare) re-gather at the worst-case capacities come
            # back behind the collective; handles are resolved once their event has fired (and all of them when a run ends)
            # ... and off the launching thread: ONE communication thread enqueues the collectives in step order (every rank in
            # the same order) and resolves them; this thread only hands the batch over
            self.gathers.append(self.cpool.submit(self._gather, res))
        self.tfin.append(time.perf_counter())
        self.stats["pages"] += len(res)
        self.stats["blocks"] += sum(len(r[2]) for r in res)
        # lazy results: the counts come from the native records (no TextBlock is built for the bookkeeping)
        self.stats["lines"] += sum(r[2].n_lines if hasattr(r[2], "n_lines") else sum(len(b.lines) for b in r[2]) for r in res)

    def _gather(self, res):
        """On the communication thread: this batch's record gather, enqueued on the communication stream; earlier gathers
        whose event has fired are resolved (the re-gather at the worst-case capacities happens there, rarely)."""
        tg = time.perf_counter()
        torch.cuda.set_device(self.dev)
        with torch.cuda.stream(self.comm_stream):
            self.handles.append(self.D.gather_results_async(res, self.total_pages, self.rank, self.world, device=self.dev, pin=True,
                                                            force=self.world == 1))
            # resolved at a FIXED lag, not "when done": a page that does not fit the compact record makes `result()` issue a second
            # collective, and every rank must issue it at the same place of its sequence of collectives
            while len(self.handles) > self.depth:
                self.gathered = self.handles.popleft().result()
        self.stats["gather_s"] = self.stats.get("gather_s", 0.0) + (time.perf_counter() - tg)

    def _drain_gathers(self):
        while self.gathers:
            self.gathers.popleft().result()
        def rest():
            with torch.cuda.stream(self.comm_stream):
                while self.handles:
                    self.gathered = self.handles.popleft().result()
        if self.cpool is not None:
            self.cpool.submit(rest).result()

    def run(self, n):
        if self.fwd_stream is not None:                      # --fwd-stream high: the forwards on a high-priority stream
            with torch.cuda.stream(self.fwd_stream):
                self._run(n)
        else:
            self._run(n)
        if self.gather:                                      # a run ends with every gather it started complete
            self._drain_gathers()

    def _batches(self, n):
        for _ in range(n):
            i = self.k
            self.k += 1
            if self.host_batches is not None:
                yield self.host_batches[i % len(self.host_batches)]
            else:
                x = self.batches[i % len(self.batches)]
                yield [x[j] for j in range(x.shape[0])]      # slices of one batch tensor: no torch.stack in the detector

    def _run(self, n):
        if self.api:
            # THE PUBLIC API: the headline's step is `TextDetector.detect_stream` -- this class only hands it the resident
            # batches and counts what comes back (and, N > 1, gathers the page records it asked for)
            for res in self.det.detect_stream(self._batches(n), self.refine_mode, self.keep_undetected, workers=self.workers,
                                              depth=self.depth, engines=self.engines, loaders=self.loaders,
                                              tail_split=self.tail_split, lazy=self.lazy, records=self.records):
                self.finish(res)
            return
        det, pending, ahead, issued = self.det, deque(), deque(), 0
        main = torch.cuda.current_stream(self.dev)
        collect = lambda futs: [r for f in futs for r in f.result()]          # noqa: E731
        for _ in range(n):
            i = self.k
            self.k += 1
            pg = None
            if self.host_batches is not None:                # loader threads stage up to `depth` batches ahead
                while issued < n and len(ahead) < self.depth:
                    ahead.append(self.lpool.submit(det._stage, self.host_batches[(i + len(ahead)) % len(self.host_batches)]))
                    issued += 1
                pg, ev = ahead.popleft().result()
                main.wait_event(ev)
            job = self.forward_job(i, pg)
            pending.append([self.pool.submit(det._tail, job, self.refine_mode, self.keep_undetected, lo, hi, self.records, self.lazy)
                            for lo, hi in det._split(self.nloc, self.tail_split)])
            while len(pending) >= self.depth:
                self.finish(collect(pending.popleft()))
        while pending:
            self.finish(collect(pending.popleft()))


def timed(run, steps, warmup, spinup, world, dev, stats=None):
    """W warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
    if spinup > 0:
        run(spinup)
    run(warmup)
    # what a serving process does once after start-up -- the product's own `serve_tuning` (gc.freeze + the youngest
    # generation's threshold), which `detect_stream` applies for any caller; here again AFTER the warm-up steps so that
    # every sub-run of this process freezes what it built
    importlib.import_module("comic-text-detector_amd.detector").serve_tuning(refreeze=True)
    grouped = world > 1 or dist.is_initialized()               # `--force-dist` at N = 1: barrier + reduction on the real backend
    if grouped:
        dist.barrier()
    torch.cuda.synchronize()
    if stats is not None:
        for k in stats:
            stats[k] = 0
    t0, c0 = time.perf_counter(), time.process_time()
    run(steps)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    dt = time.perf_counter() - t0
    if stats is not None:                        # CPU time of this process (all its threads) per second of wall clock
        stats["cpu_cores"] = (time.process_time() - c0) / max(dt, 1e-9)
    if grouped:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# =====================================================================================================================
# baselines and parity legs (rank 0, N = 1)
# =====================================================================================================================

def cpu_baseline(pkg, ckpt, size: int, pages, canned_sample=None, budget_s: float = 14.0, max_pages: int = 16):
    """The reference's CPU path restated, both legs at the SAME core budget: the oracle forward (CPU fp32, bit-exact with
    the reference's torch modules) at bs=1 like `TextDetector.__call__`, on the thread count that is fastest on this host;
    and the oracle tail (the reference's post-processing restated in numpy, single-threaded per page like the reference's
    cv2 / numpy code) for those pages' own forward outputs on a pool of as many PROCESSES (oracle/tail_pool.py) -- what a
    host-only deployment with that many cores would do.  value = pages / (forward time + pooled tail time)."""
    import tempfile
    from oracle.net_ref import OracleNet
    from oracle import postproc_ref as R
    hi = host_info()
    avail = hi["usable_cpus"]
    net = OracleNet(ckpt)
    g = torch.Generator().manual_seed(123)
    # pick the thread count that is fastest on this host (a 256-thread oneDNN run at bs=1 is ~100x slower than 32)
    xs = torch.rand(1, 3, 256, 256, generator=g)
    best = (1e30, 1)
    for nt in sorted({min(avail, c) for c in (8, 16, 32, 64, 128, avail)}):
        torch.set_num_threads(nt)
        net(xs)
        t0 = time.perf_counter()
        net(xs)
        dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    pages = list(pages)[:max_pages]
    x0 = torch.from_numpy(np.ascontiguousarray(pages[0].transpose(2, 0, 1)[None])).float() / 255
    net(x0)                                         # warm-up (allocator, oneDNN primitives)
    outs, t_net = [], 0.0
    t0 = time.perf_counter()
    for pg in pages:
        if outs and (time.perf_counter() - t0) > budget_s:
            break
        x = torch.from_numpy(np.ascontiguousarray(pg.transpose(2, 0, 1)[None])).float() / 255
        ta = time.perf_counter()
        ob, om, ol = net(x)
        t_net += time.perf_counter() - ta
        if canned_sample is not None:
            _, blks, mask_u8, prob, _ = canned_sample
            outs.append((blks[0], ((mask_u8.astype(np.float32) + 0.5) / 255)[None], np.stack([prob, np.zeros_like(prob)])))
        else:
            outs.append((ob.numpy()[0], om.numpy()[0], ol.numpy()[0]))
    n = len(outs)
    pages = pages[:n]
    # tail leg: a pool of `cores` processes, in a child interpreter without torch / HIP (fork-safe)
    pool = None
    try:
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "in.npz")
            np.savez(path, pages=np.stack(pages), blks=np.stack([o[0] for o in outs]), mask=np.stack([o[1] for o in outs]),
                     lines=np.stack([o[2] for o in outs]))
            env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
            p = subprocess.run([sys.executable, "-m", "oracle.tail_pool", path, str(cores), str(size)], capture_output=True,
                               text=True, timeout=120, cwd=ROOT, env=env)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            pool = json.loads(line[-1]) if line else {"error": (p.stderr or "no output")[-300:]}
    except Exception as e:                          # fall back to one page in this process
        pool = {"error": repr(e)[:300]}
    if "wall_s" in pool:
        t_tail_pooled = pool["wall_s"]
        t_tail_single = float(np.mean(pool["per_page_single_s"]))
        procs = pool["processes"]
    else:
        ta = time.perf_counter()
        R.detector_tail(pages[0], outs[0][0][None], outs[0][1][None], outs[0][2][None], input_size=(size, size), refine_mode=0,
                        keep_undetected_mask=False)
        t_tail_single = time.perf_counter() - ta
        t_tail_pooled, procs = t_tail_single * n, 1
    value = n / (t_net + t_tail_pooled)
    return {"value": round(value, 4), "unit": "pages/s", "cores": cores, "kind": "port", "host": hi,
            "forward_threads": cores, "tail_processes": procs,
            "forward_ms_per_page": round(t_net / n * 1e3, 1), "tail_ms_per_page_one_process": round(t_tail_single * 1e3, 1),
            "tail_ms_per_page_pooled": round(t_tail_pooled / n * 1e3, 1),
            "serial_one_page_pages_per_s": round(1.0 / (t_net / n + t_tail_single), 4),
            "pool": {k: v for k, v in pool.items() if k in ("error", "blocks", "lines")},
            "sample": f"{n} pages of {size}x{size}: torch CPU fp32 oracle forward at bs=1 ({t_net / n * 1e3:.0f} ms/page on "
                      f"{cores} threads, the fastest count on this {avail}-CPU host) + oracle tail in numpy on "
                      + ("text-like maps of the same pages" if canned_sample is not None else "those forwards' own outputs")
                      + f" ({t_tail_single * 1e3:.0f} ms/page in one process; {t_tail_pooled / n * 1e3:.0f} ms/page on a pool of "
                      f"{procs} processes = the same core budget); value = pages / (forward + pooled tail); a single "
                      f"`TextDetector.__call__` at a time (forward, then a one-thread tail) gives "
                      f"{1.0 / (t_net / n + t_tail_single):.2f} pages/s"}


FP16_BAND_EPS = 4e-3            # = tests/test_gpu_accept.py EPS_FP16


def parity_block(pkg, ckpt, det, batch: torch.Tensor, idx: int, size: int) -> dict:
    """The second half of BASELINE's metric on ONE page of the benchmark input (checkpoint and page as benchmarked), taken
    from a forward of the WHOLE benchmark batch -- the dispatch that is timed (at B = 32 the grid thresholds select
    `conv_halo3_kernel`, `c3_fused_kernel`, `c3b_kernel`; a B = 1 re-run of the page would go through other kernels):
    (1) every engine's maps against the oracle forward; (2) for the fp16 engine the BOUND on its deviation (every
    thresholded pixel that differs lies within eps of the threshold in the oracle's map; every differing line / block
    touches such a pixel); (3) end to end -- lines / blocks / masks of every engine (`detect_batch` of the batch) against
    oracle forward + oracle tail (= the reference's TextDetector.__call__ restated)."""
    from oracle import accept
    from oracle import postproc_ref as R
    from oracle.net_ref import OracleNet
    DET = importlib.import_module("comic-text-detector_amd.detector")
    page = batch[idx].cpu().numpy()
    x = torch.from_numpy(np.ascontiguousarray(page.transpose(2, 0, 1)[None])).float() / 255
    torch.set_num_threads(min(32, host_info()["usable_cpus"]))
    ob, om, ol = OracleNet(ckpt)(x)
    ref = R.detector_tail(page, ob.numpy(), om.numpy(), ol.numpy(), input_size=(size, size), refine_mode=0,
                          keep_undetected_mask=False)
    ref_dets = np.asarray(R.non_max_suppression(ob.numpy(), 0.4, 0.35)[0])
    sbb = accept.score_band_boxes(ol.numpy(), (size, size), FP16_BAND_EPS)
    ref_cand = np.asarray(R.seg_rep((size, size), ol.numpy())[0][0])
    BKm = importlib.import_module("comic-text-detector_amd.backend")
    nB = int(batch.shape[0])
    out = {"page": f"page {idx} of the benchmark's first batch ({size}x{size}), benchmark checkpoint; oracle = CPU fp32 restatement "
                   "of the reference net + restated tail",
           "dispatch": f"as timed: every engine's outputs come from ONE forward / `detect_batch` of the {nB}-page benchmark batch "
                       "at the default grid thresholds",
           "engines": {}}
    dev = det.net.device
    for prec in ("fp16", "fp32s", "fp32"):
        d = det if det.precision == prec else DET.TextDetector(ckpt, input_size=size, device=dev, precision=prec)
        # the fp32 engine is benchmarked at bs = 8 (BASELINE configs[1]): its batch here is the first 8 pages
        xb = batch[: max(8, idx + 1)] if prec == "fp32" else batch
        got = d.detect_batch([xb[i] for i in range(int(xb.shape[0]))], refine_mode=0, keep_undetected_mask=False)[idx]
        rep = accept.compare(got, ref)
        rep["batch"] = int(xb.shape[0])
        blks, mask, lines = d.net.forward_u8(xb)
        torch.cuda.synchronize()
        kern = d.net.op_kernels()
        rep["kernels"] = sorted({k for _, k in kern if k != "(fused)"})
        blks, mask, lines = blks[idx: idx + 1], mask[idx: idx + 1], lines[idx: idx + 1]
        mu8, bmp = d.net.mask_u8[idx: idx + 1], d.net.bitmap[idx: idx + 1]
        band = accept.band_report(ol[0, 0].numpy(), om[0, 0].numpy(), bmp[0].cpu().numpy(),
                                  mu8[0].cpu().numpy(), FP16_BAND_EPS, prob=lines[0, 0].cpu().numpy(),
                                  mask=mask[0, 0].cpu().numpy())
        flips = band.pop("_flips")
        dets, counts = BKm.nms(blks.contiguous(), 0.4, 0.35)
        extras = d.tail_batch([page], blks.contiguous(), mu8.contiguous(), lines[:, 0].contiguous(), bmp.contiguous(),
                              want_extras=True)[0][3]
        band.update(accept.explain_geometry(got, ref, flips, dets=dets[0, : int(counts[0])].cpu().numpy(), ref_dets=ref_dets,
                                            score_band_boxes=sbb, candidates=(extras["db_boxes"], ref_cand, 1000)))
        rep["band"] = band
        out["engines"][prec] = rep
        if d is not det:
            del d
            torch.cuda.empty_cache()
    b16 = out["engines"]["fp16"]["band"]
    out["fp16_band"] = {"eps": FP16_BAND_EPS,
                        "claim": "the fp16 engine's maps stay within eps of the oracle's; every DB-bitmap (0.3) / mask@127 pixel "
                                 "that differs lies within eps of the threshold in the ORACLE's map; every differing line / "
                                 "block is attributed to such a pixel, to an int32 truncation of coordinates < 1 px apart, or "
                                 "to a detection NMS kept differently (tests/test_gpu_accept.py, tests/test_gpu_dispatch.py, "
                                 "oracle/accept.py)",
                        "holds": bool(b16["bitmap_flips_out_of_band"] == 0 and b16["mask127_flips_out_of_band"] == 0 and
                                      b16["prob_max_abs_delta"] < FP16_BAND_EPS and b16["mask_max_abs_delta"] < FP16_BAND_EPS and
                                      b16["lines_unexplained"] == 0 and b16["blocks_unexplained"] == 0),
                        "in_band_pixel_frac": {"bitmap": b16["bitmap_in_band_frac"], "mask127": b16["mask127_in_band_frac"]}}
    out["tail"] = "bit-exact vs the oracle tail on identical network outputs (tests/test_gpu_e2e.py)"
    return out


def rocm_baseline_main(args) -> None:
    """`--mode rocm-baseline` (normally run as a time-boxed subprocess of the default run): the reference's network as
    stock PyTorch-ROCm executes it (oracle/net_torch_device.py: ATen + MIOpen, NCHW), forward only, same pages."""
    from oracle.net_torch_device import TorchDeviceNet
    pkg = importlib.import_module("comic-text-detector_amd")
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    ckpt = pkg.synth.make_blob_checkpoint(0)
    S = args.size
    out = {"what": "oracle/net_torch_device.py: the reference's torch network (yolo BN folded like Model.fuse, head BNs "
                   "separate, NCHW) on this GPU through PyTorch-ROCm ATen / MIOpen (cudnn.benchmark off), forward only, "
                   "f32 input already on the device; first call (MIOpen kernel compilation) excluded",
           "torch": torch.__version__}
    t_start = time.perf_counter()
    for name, dtype, B in (("fp16_bs32", torch.float16, args.batch), ("fp32_bs32", torch.float32, args.batch),
                           ("fp32_bs8", torch.float32, 8)):
        if time.perf_counter() - t_start > args.rocm_budget:
            out[name] = {"skipped": "time budget"}
            continue
        try:
            net = TorchDeviceNet(ckpt, dev, dtype)
            pages = np.stack([pkg.synth.text_like_page((S, S), i) for i in range(B)])
            x = (torch.from_numpy(pages).to(dev).permute(0, 3, 1, 2).float() / 255).to(dtype).contiguous()
            t0 = time.perf_counter()
            net(x)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 5
            e0.record()
            for _ in range(n):
                net(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            out[name] = {"ms_per_forward": round(ms, 3), "pages_per_s": round(B / ms * 1e3, 1), "batch": B,
                         "first_call_s": round(first, 1)}
            del net, x
            torch.cuda.empty_cache()
        except Exception as e:                           # a baseline must never take the bench line down
            out[name] = {"error": repr(e)[:300]}
        print(json.dumps({"rocm_baseline_partial": out}), flush=True)
    print(json.dumps({"rocm_baseline": out}), flush=True)


def rocm_baseline_subprocess(args, timeout_s: float):
    """Runs `--mode rocm-baseline` in a child process with a wall-clock limit (MIOpen compiles its kernels at first use on
    a fresh box: minutes).  When the child measures nothing in time the entries are null: the line carries numbers of
    this run only."""
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "rocm-baseline", "--batch", str(args.batch), "--size",
           str(args.size), "--rocm-budget", str(max(20.0, timeout_s - 30))]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("MIOPEN_FIND_MODE", "FAST")
    res, note, last = None, None, None
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        lines = p.stdout.splitlines()
        if p.returncode != 0 and not lines:
            note = "child failed: " + p.stderr[-300:]
    except subprocess.TimeoutExpired as e:
        so = e.stdout
        lines = (so.decode() if isinstance(so, bytes) else (so or "")).splitlines()
        note = f"child stopped after {timeout_s:.0f} s (MIOpen kernel compilation); partial results kept"
    for ln in lines:
        try:
            d = json.loads(ln)
        except Exception:
            continue
        if "rocm_baseline" in d:
            res = d["rocm_baseline"]
        elif "rocm_baseline_partial" in d:
            last = d["rocm_baseline_partial"]
    res = res or last
    have = isinstance(res, dict) and any(isinstance(v, dict) and "ms_per_forward" in v for v in res.values())
    if not have:                                     # nothing measured in THIS run: say so, never a committed figure
        return {"fp16_bs32": None, "fp32_bs32": None, "fp32_bs8": None,
                "note": (note or "child produced nothing") + "; no number from this run (an earlier measurement of the same "
                        "command is in profiles/r03_rocm_baseline.json)"}
    if isinstance(res, dict) and note:
        res["note"] = note
    return res if isinstance(res, dict) else {"error": note or "no output"}


def sub_bench(argv, steps: int, warmup: int = 3, spinup: int = 40, timeout: float = 150.0, whole: bool = False) -> dict:
    """One configuration of this script in a child process (`--no-cpu-baseline --no-extras`): value / ms_per_step /
    blocks and lines per page / network ms of its JSON line (`whole`: the line itself)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--spinup", str(spinup),
           "--no-cpu-baseline", "--no-extras"] + list(argv)
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not line:
            return {"error": (p.stderr or "no output")[-300:], "command": " ".join(cmd[1:])}
        d = json.loads(line[-1])
    except Exception as e:                                   # a sub-run must never take the bench line down
        return {"error": repr(e)[:300], "command": " ".join(cmd[1:])}
    if whole:
        for k in ("cpu_baseline", "parity", "parity_exact", "extra_configs", "rocm_baseline", "serial_step"):
            d.pop(k, None)
        d["command"] = "python bench.py " + " ".join(cmd[2:])
        return d
    out = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
           "blocks_per_page": d["config"].get("blocks_per_page"), "lines_per_page": d["config"].get("lines_per_page"),
           "command": "python bench.py " + " ".join(cmd[2:])}
    if isinstance(d.get("roofline"), dict):
        out["net_ms_per_step"] = d["roofline"].get("net_ms_per_step")
    return out


def inproc_bench(pkg, D, DET, TL, base_args, dev, steps: int, warmup: int = 3, spinup: int = 40, **over) -> dict:
    """One more configuration of the end-to-end pipeline IN THIS PROCESS (rank 0, N = 1): its own checkpoint / pages /
    detector / worker pool, closed again afterwards.  Worker threads take over the native tails (streams, buffers) of the
    pools before them (tail._Lease) -- with FRESH tails per pool a later pipeline measured 12-18 % low once the process had
    owned more tail streams than hardware queues (scripts/gpu_inprocess.py), which is why round 3 ran these as children."""
    import copy
    import gc
    a = copy.copy(base_args)
    for k, v in over.items():
        setattr(a, k, v)
    try:
        ckpt, batches, canned, _ = make_workload(pkg, a, 0, a.batch, dev)
        det = DET.TextDetector(ckpt, input_size=a.size, device=dev, precision=a.precision)
        pipe = Pipeline(det, batches, canned, dev, 1, 0, a.batch, D, a.workers, a.depth, a.tail_split,
                        host_input=a.host_input, loaders=a.loaders, engines=a.engines, keep_undetected=a.keep_undetected,
                        lazy=bool(a.lazy_blocks), refine_mode=getattr(a, "refine_mode", 0))
        gc.unfreeze()
        dt = timed(pipe.run, steps, warmup, spinup, 1, dev, pipe.stats)
        st = dict(pipe.stats)
        prof = det.net.profile(batches[0])
        pipe.close()
        out = {"value": round(a.batch * steps / dt, 2), "unit": "pages/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
               "blocks_per_page": round(st["blocks"] / max(st["pages"], 1), 2), "lines_per_page": round(st["lines"] / max(st["pages"], 1), 2),
               "net_ms_per_step": round(float(prof["ms"].sum()), 3), "host_cpu_cores_used": round(float(st.get("cpu_cores", 0.0)), 2),
               "in_process": True, "overrides": {k: (v if isinstance(v, (int, float, str, bool)) else str(v)) for k, v in over.items()}}
        del pipe, det, batches, canned, ckpt
        gc.collect()
        torch.cuda.empty_cache()
        return out
    except Exception as e:                                   # a sub-run must never take the bench line down
        return {"error": repr(e)[:300], "overrides": {k: str(v) for k, v in over.items()}}


# =====================================================================================================================
# BASELINE configs[4]: mixed-size stream
# =====================================================================================================================

def mixed_stream(pkg, D, BK, det, rank, world, dev, steps, warmup, with_tail=True, n_per_gpu=512) -> dict:
    """A seeded stream of pages of three sizes, batched dynamically per size bucket under a fixed pixel budget, every
    bucket's forward captured ONCE into a hipGraph (largest bucket first, so the arena never moves; two instances per
    bucket so that a replay never overwrites outputs a tail is still reading) and replayed in steady state; the native
    tail of every batch (per-page metas) on worker threads under the next replays.  A step = one pass over this rank's
    shard of the stream."""
    be = det.net
    sizes = (640, 1024, 1536)
    budget = 32 * 1024 * 1024                     # pixels per batch: 81 -> 64 @ 640, 32 @ 1024, 14 @ 1536
    cap = {s: max(1, min(64, budget // (s * s))) for s in sizes}
    n_stream = n_per_gpu * world
    rng = np.random.RandomState(2024)
    stream = rng.choice(sizes, size=n_stream, p=[0.3, 0.5, 0.2])
    lo, hi = D.shard_range(n_stream, rank, world)
    mine = stream[lo:hi]
    # dynamic batching: pages are taken in stream order, a bucket is flushed when it is full (and at the end)
    batches, open_b = [], {s: 0 for s in sizes}
    for s in mine:
        open_b[s] += 1
        if open_b[s] == cap[s]:
            batches.append((int(s), cap[s]))
            open_b[s] = 0
    for s in sizes:
        if open_b[s]:
            batches.append((int(s), open_b[s]))
    shapes = sorted({b for b in batches}, key=lambda t: -t[0] * t[0] * t[1])
    # page pool per size (distinct text-like pages; a batch takes a rotating window of its size's pool)
    pool_pages = {s: torch.from_numpy(np.stack([pkg.synth.text_like_page((s, s), 7000 + s + i)
                                                for i in range(2 * cap[s])])).to(dev) for s in sizes}
    graphs, setup = {}, {}
    for s, n in shapes:                            # one eager pass over every bucket: the arena reaches its final size
        be.forward_u8(torch.zeros((n, s, s, 3), dtype=torch.uint8, device=dev))
    for s, n in shapes:                            # ... so no capture below can move it (stale-graph guard)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        inst = []
        for _ in range(2 if with_tail else 1):
            static_in, replay = be.capture(n, s, s, "u8")
            outs = replay()
            inst.append(dict(static_in=static_in, replay=replay, outs=outs, mask_u8=be.mask_u8, bitmap=be.bitmap, busy=None))
        torch.cuda.synchronize()
        setup[f"{n}x{s}x{s}"] = round((time.perf_counter() - t0) * 1e3 / len(inst), 2)
        graphs[(s, n)] = inst
    tb = thread_budget(world)
    pool = ThreadPoolExecutor(max_workers=tb["tail_workers"], thread_name_prefix="ctd-tail") if with_tail else None
    stats = {"pages": 0, "blocks": 0, "lines": 0}
    turn = {k: 0 for k in graphs}
    cursor = {s: 0 for s in sizes}

    def drain(fut):
        for r in fut.result():
            stats["pages"] += 1
            stats["blocks"] += len(r[2])
            stats["lines"] += r[2].n_lines if hasattr(r[2], "n_lines") else sum(len(b.lines) for b in r[2])

    def run_steps(k):
        for _ in range(k):
            for key in batches:
                s, n = key
                g = graphs[key][turn[key] % len(graphs[key])]
                turn[key] += 1
                if g["busy"] is not None:                 # the tail that last read this instance's outputs
                    drain(g["busy"])
                    g["busy"] = None
                c = cursor[s] % (pool_pages[s].shape[0] - n + 1)      # a rotating window of n pages of this size's pool
                cursor[s] += n
                x = pool_pages[s][c: c + n]
                g["static_in"].copy_(x)
                blks, _, lines = g["replay"]()
                if not with_tail:
                    BK.nms(blks, 0.4, 0.35)
                    continue
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                job = dict(gpu=[x[j] for j in range(n)], metas=[(s, s, 0, 0)] * n, blks=blks, mask_u8=g["mask_u8"],
                           lines_map=lines, bitmap=g["bitmap"], ev=ev)
                g["busy"] = pool.submit(det._tail, job, 0, False, None, None, None, True)
            for inst in graphs.values():
                for g in inst:
                    if g["busy"] is not None:
                        drain(g["busy"])
                        g["busy"] = None

    dt = timed(run_steps, steps, warmup, 0, world, dev, stats)
    # roofline: algorithmic bytes / flops of the conv family per step from the engine's plan of every bucket, over the
    # family's hipEvent time in one profiled forward per bucket
    fam_ms = fam_bytes = fam_flops = 0.0
    count = {}
    for key in batches:
        count[key] = count.get(key, 0) + 1
    for (s, n), c in count.items():
        prof = be.profile(pool_pages[s][:n])
        cls = prof["cls"]
        fam = (cls == 1) | (cls == 2)
        fam_ms += c * float(prof["ms"][fam].sum())
        fam_bytes += c * float(prof["bytes"][fam].sum())
        fam_flops += c * float(prof["flops"][fam].sum())
    if pool is not None:
        pool.shutdown(wait=True)
    mpix = float(sum(s * s * n for s, n in batches)) / 1e6
    prec = det.precision
    ach_gbs, ach_tf = fam_bytes / (fam_ms * 1e-3) / 1e9, fam_flops / (fam_ms * 1e-3) / 1e12
    return {"metric": "pages/sec, mixed-size stream (640/1024/1536), hipGraph steady state"
                      + (" + native tail" if with_tail else " + GPU NMS"),
            "value": round(n_stream * steps / dt, 2), "unit": "pages/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "dtype": DTYPE[prec],
            "config": {"workload": f"BASELINE configs[4]: seeded stream of {n_per_gpu} pages per GPU, sizes 640/1024/1536 drawn "
                                   "30/50/20 %, dynamic batches under a 32 Mpixel budget per batch, one hipGraph per (size, "
                                   "batch) bucket captured once (largest first, two output instances), text-like pages + blob "
                                   "checkpoint, " + ("the WHOLE native tail per batch on its own forward's outputs (per-page "
                                   "metas), tails on worker threads under the next replays" if with_tail else "forward + GPU NMS"),
                       "pages_per_step": int(n_stream), "batches_per_step_rank0": len(batches),
                       "bucket_caps": {str(k): v for k, v in cap.items()}, "mpixels_per_step_rank0": round(mpix, 1),
                       "mpixels_per_s_rank0": round(mpix * steps / dt, 1), "precision": prec,
                       "blocks_per_page": round(stats["blocks"] / max(stats["pages"], 1), 2) if with_tail else None,
                       "lines_per_page": round(stats["lines"] / max(stats["pages"], 1), 2) if with_tail else None},
            "plan_and_capture_ms": setup,
            "roofline": {"bound": "hbm", "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach_gbs / HBM_PEAK_GBS, 4), "kernel": FAMILY[prec],
                         "family_ms_per_step": round(fam_ms, 3), "alg_bytes_per_step": fam_bytes,
                         "alg_flops_per_step": fam_flops, "tflops": round(ach_tf, 1),
                         "mfma_frac": round(ach_tf / PEAK_TF[prec], 4), "traffic": None}}


# =====================================================================================================================
# roofline of the headline engine
# =====================================================================================================================

def roofline_block(be, x, precision: str, B: int, S: int, dump_ops: str = "") -> dict:
    prof = be.profile(x)
    ms, fl, by, cls = prof["ms"], prof["flops"], prof["bytes"], prof["cls"]
    fam = (cls == 1) | (cls == 2)
    if not fam.any():                          # no MFMA ops in this program: the direct kernels are the family
        fam = cls == 3
    fam_ms, fam_flops, fam_bytes = float(ms[fam].sum()), float(fl[fam].sum()), float(by[fam].sum())
    net_ms = float(ms.sum())
    ach_gbs = fam_bytes / (fam_ms * 1e-3) / 1e9
    ach_tf = fam_flops / (fam_ms * 1e-3) / 1e12
    ai = fam_flops / fam_bytes
    peak_tf = PEAK_TF[precision]
    ridge = peak_tf * 1e12 / (HBM_PEAK_GBS * 1e9)
    if ai < ridge:
        roof = {"bound": "hbm", "achieved": round(ach_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach_gbs / HBM_PEAK_GBS, 4)}
    else:
        roof = {"bound": "mfma", "achieved": round(ach_tf, 1), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                "frac": round(ach_tf / peak_tf, 4)}
    names = prof["names"]

    def is_backbone(nm: str) -> bool:           # yolo.model.0-9: the layers north_star's 60 % HBM target is about
        parts = nm.split(".")
        if parts[0] == "yolo":
            parts = parts[1:]
        return len(parts) > 1 and parts[0] == "model" and parts[1].isdigit() and int(parts[1]) <= 9
    bb = np.array([is_backbone(nm) for nm in names])
    if bb.any():
        bb_ms, bb_bytes = float(ms[bb].sum()), float(by[bb].sum())
        roof["backbone"] = {"ms_per_step": round(bb_ms, 3), "alg_bytes_per_step": bb_bytes,
                            "gbs": round(bb_bytes / (bb_ms * 1e-3) / 1e9, 1),
                            "hbm_frac": round(bb_bytes / (bb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    # HBM traffic of the same kernel family from PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate rocprofv3
    # --pmc passes: scripts/gpu_traffic.sh).  PMC cannot be collected inside this process; the committed measurement of
    # this engine at this shape is attached when it exists (and says which round it is from).
    traffic, tnote = None, None
    for tname in {"fp16": ("r06_traffic_pmc.json", "r05_traffic_pmc.json", "r04_traffic_pmc.json"), "fp32s": ("r03_traffic_pmc_fp32s.json",)}.get(precision, ()):
        tpath = os.path.join(ROOT, "profiles", tname)
        if os.path.isfile(tpath) and (B, S) == (32, 1024):
            try:
                traffic = float(json.load(open(tpath))["hbm_bytes_per_forward_corrected"])
                tnote = f"bytes per forward of this kernel family from profiles/{tname} (rocprofv3 PMC passes of this " \
                        f"engine at this shape, not collected in this run)"
                break
            except Exception:
                traffic = None
    roof.update({"traffic": traffic, "traffic_note": tnote,
                 "kernel": "conv_direct / convt_direct (exact-fp32 VALU kernels)" if not (cls[fam] != 3).any() else FAMILY[precision],
                 "launches_per_step": int((fam & (by > 0)).sum()), "family_ms_per_step": round(fam_ms, 3),
                 "net_ms_per_step": round(net_ms, 3), "alg_bytes_per_step": fam_bytes, "alg_flops_per_step": fam_flops,
                 "tflops": round(ach_tf, 1), "mfma_frac": round(ach_tf / peak_tf, 4), "gbs": round(ach_gbs, 1),
                 "arith_intensity": round(ai, 1)})
    try:
        # per-op roofline: every launch priced at max(its algorithmic bytes / HBM peak, its flops / MFMA peak), summed
        bound_ms = np.maximum(by[fam] / (HBM_PEAK_GBS * 1e9), fl[fam] / (peak_tf * 1e12)) * 1e3
        roof["per_op"] = {"bound_ms_per_step": round(float(bound_ms.sum()), 3),
                          "frac": round(float(bound_ms.sum()) / fam_ms, 4),
                          "hbm_side_launches": int((by[fam] / (HBM_PEAK_GBS * 1e9) >= fl[fam] / (peak_tf * 1e12)).sum())}
    except Exception as e:                          # never lose the bench line to a supplementary number
        roof["per_op"] = {"error": repr(e)}
    if dump_ops:
        with open(dump_ops, "w") as f:
            f.write("op\tclass\tms\tGFLOP\tMB\tTFLOP/s\tGB/s\n")
            for i, nm in enumerate(names):
                t = max(ms[i], 1e-6) * 1e-3
                f.write(f"{nm}\t{cls[i]}\t{ms[i]:.4f}\t{fl[i] / 1e9:.3f}\t{by[i] / 1e6:.2f}\t"
                        f"{fl[i] / t / 1e12:.1f}\t{by[i] / t / 1e9:.1f}\n")
    return roof


# =====================================================================================================================
# multi-GPU launch
# =====================================================================================================================

def relaunch_under_torchrun(args) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start N ranks of this script on this node.  On a box
    with fewer than N devices every rank shares device 0 over gloo (CTD_BENCH_ONE_DEVICE): a REHEARSAL of the N > 1
    code path, flagged as such in the line, not a measurement."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    if ndev < args.gpus:
        env["CTD_BENCH_ONE_DEVICE"] = "1"
        env["CTD_DIST_BACKEND"] = "gloo"
        print(f"bench.py: {ndev} device(s) for --gpus {args.gpus}: one-device rehearsal over gloo", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="pages per GPU per step")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--precision", default="fp16", choices=["fp16", "fp32", "fp32s"])
    ap.add_argument("--mode", default="e2e", choices=["e2e", "net", "mixed", "rocm-baseline"],
                    help="e2e: forward + the whole native tail on the forward's outputs (default); net: forward + GPU NMS only; "
                         "mixed: BASELINE configs[4], a seeded stream of 640 / 1024 / 1536 pages, dynamic batches, one "
                         "captured hipGraph per size bucket, native tail; rocm-baseline: the reference's torch network "
                         "through PyTorch-ROCm / MIOpen (the child process of the default run)")
    ap.add_argument("--tail-input", default="forward", choices=["forward", "canned"],
                    help="forward: the tail consumes the timed forward's own outputs (blob checkpoint, text-like pages); "
                         "canned: round 2's workload (random checkpoint, the tail fed text-like maps of the same pages)")
    ap.add_argument("--batches", type=int, default=4, help="distinct batches rotated through the steps (HBM resident)")
    ap.add_argument("--dense-blocks", action="store_true",
                    help="blob checkpoint without `sparse_det`: ~65 text blocks of ~160 px per page (1.5 page areas of block "
                         "windows) instead of the reference fixture's density (~15 blocks); also a sub-run of the default line")
    ap.add_argument("--line-density", default="fixture", choices=["fixture", "r3"],
                    help="fixture (default): ~29 text lines per page, the count on the reference's real page "
                         "(data/examples/AisazuNihaIrarenai-003.jpg: 16 blocks / 29 lines); r3: round 3's pages (16 / 16)")
    ap.add_argument("--workers", type=int, default=0, help="tail worker threads (e2e); 0 = from the host-thread budget")
    ap.add_argument("--depth", type=int, default=4, help="batches in flight (e2e)")
    ap.add_argument("--tail-split", type=int, default=int(os.environ.get("BENCH_TAIL_SPLIT", "0")),
                    help="work items (page ranges) a batch's tail is cut into (e2e); 0 = one per tail worker")
    ap.add_argument("--host-input", action="store_true",
                    help="e2e: the pages start in HOST memory (numpy, as the reference's callers hand them over): pinned "
                         "staging + one async H2D per batch on loader threads; the PCIe-inclusive rate of DESIGN.md")
    ap.add_argument("--loaders", type=int, default=2, help="loader threads of --host-input")
    ap.add_argument("--engines", type=int, default=1, help="engine copies on their own streams (e2e)")
    ap.add_argument("--keep-undetected", action="store_true", help="also run refine_undetected_mask in the tail")
    ap.add_argument("--fwd-stream", default="default", choices=["default", "high"],
                    help="e2e: the forwards on torch's current stream (default) or on a stream of the highest priority (ONE such "
                         "stream: the class has few hardware queues, DESIGN 4.4)")
    ap.add_argument("--tail-only", action="store_true",
                    help="e2e pipeline WITHOUT the forward: one forward at set-up, then every step runs the native tail on those "
                         "outputs (workers, work items, record gather as in e2e).  With --gpus N on one device (rehearsal) this "
                         "measures what N ranks' HOST sides -- interpreter pipelines, tail workers x geometry threads, pinned "
                         "buffers, the gather -- cost each other; not a detector rate")
    ap.add_argument("--lazy-blocks", action="store_true",
                    help="e2e: the tail workers hand over every page's native records as a lazily materialised BlockList "
                         "(`detect_stream(lazy=True)`) instead of building the reference's return type, a list of Python TextBlock "
                         "objects (the default, and what the headline times since round 5)")
    ap.add_argument("--spinup", type=int, default=int(os.environ.get("BENCH_SPINUP", "100")),
                    help="untimed steps before the warm-up steps (clock / host-side spin-up of a fresh process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip parity_exact / extra_configs / rocm_baseline sub-runs")
    ap.add_argument("--rocm-timeout", type=float, default=float(os.environ.get("BENCH_ROCM_TIMEOUT", "100")),
                    help="wall-clock limit of the rocm_baseline child process (s); 0 skips it")
    ap.add_argument("--rocm-budget", type=float, default=240.0, help="(rocm-baseline mode) stop starting new cases after this many s")
    ap.add_argument("--dump-ops", default="", help="write the per-op profile table to this file")
    ap.add_argument("--cu-split", type=int, default=0,
                    help="e2e experiment: confine the tails' streams to this many CUs (mask bits 0..T-1, the same share of every "
                         "XCD) and run the forwards on a stream masked to the REST of the chip (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--force-dist", default="", choices=["", "nccl", "gloo"],
                    help="N = 1: join a world-size-1 process group of this backend and run the N > 1 step's record gather (on the "
                         "communication stream, from the page-locked record) inside every step -- RCCL exercised on the one GPU a "
                         "1-GPU box has; `config.parallelism` records the backend")
    ap.add_argument("--refine-mode", type=int, default=0, choices=[0, 1],
                    help="0 = REFINEMASK_INPAINT (the API default), 1 = REFINEMASK_ANNOTATION (the reference CLI's, inference.py:35)")
    ap.add_argument("--no-pin", action="store_true",
                    help="N > 1: do NOT bind the rank's threads to the CPUs of its GPU's NUMA node (affinity.py; A/B knob)")
    args = ap.parse_args()

    if args.mode == "rocm-baseline":
        return rocm_baseline_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    pkg = importlib.import_module("comic-text-detector_amd")
    D = importlib.import_module("comic-text-detector_amd.dist")
    DET = importlib.import_module("comic-text-detector_amd.detector")
    BK = importlib.import_module("comic-text-detector_amd.backend")
    TL = importlib.import_module("comic-text-detector_amd.tail")
    # The process group is joined LAST, after the detector's streams exist (the forwards', the tail workers', the upload and
    # communication streams): RCCL creates streams of its own at start-up, and a process whose important streams come after
    # them shares hardware queues badly -- the same pipeline ran at 2480-2590 pages/s with the group initialised first and at
    # 3134 with it initialised here (profiles/r06_rccl_init_order.txt).  Rank and world come from the environment.
    rank, local_rank, world = D.env_world()
    def join():
        # librccl prints its version banner (five lines) to STDOUT when the communicator starts; rank 0's stdout carries ONE
        # JSON line, so the file descriptor points at stderr while the group is joined
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            return D.init(args.force_dist or None, force=bool(args.force_dist))
        finally:
            sys.stdout.flush()
            try:                                             # the banner sits in C stdio's buffer: flush it while fd 1 is stderr
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            os.dup2(saved, 1)
            os.close(saved)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world
    # CTD_BENCH_ONE_DEVICE=1 (with CTD_DIST_BACKEND=gloo): every rank on GPU 0 -- a rehearsal of the N > 1 code path on a
    # 1-GPU box, not a measurement
    one_device = bool(os.environ.get("CTD_BENCH_ONE_DEVICE"))
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    # N > 1: bind this rank's host side (launcher, tail workers and their native threads, loaders -- all created below
    # and inheriting the mask) to CPUs of ITS GPU's NUMA node, disjoint from the other ranks' (affinity.py)
    AFF = importlib.import_module("comic-text-detector_amd.affinity")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    pin = AFF.rank_cpus(local_rank % max(1, local_world), local_world,
                        gpu_of_rank=[0] * local_world if one_device else list(range(local_world)))
    hi_ = host_info()
    host_cpus = hi_.get("affinity_cpus", hi_["usable_cpus"])   # what the host's affinity mask offers (a CPU quota is reported beside it)
    pinned = bool(world > 1 and not args.no_pin and AFF.apply(pin["cpus"]))
    cpu_affinity = {"pinned": pinned, "numa_node": pin["node"], "source": pin["source"], "n_cpus": len(pin["cpus"]),
                    "cpus": f"{pin['cpus'][0]}..{pin['cpus'][-1]}" if pin["cpus"] else ""}
    tb = thread_budget(world, pinned, host_cpus)
    if args.workers <= 0:
        args.workers = tb["tail_workers"]
    if args.tail_split <= 0:
        args.tail_split = max(1, args.workers)
    TL.set_host_threads(tb["native_threads_per_worker"])

    B, S = args.batch, args.size
    total_pages = B * n_gpus                      # weak scaling: fixed per-GPU work
    lo, hi = D.shard_range(total_pages, rank, world)
    nloc = hi - lo

    if args.mode == "mixed":
        ckpt = blob_checkpoint(pkg, args)
        det = DET.TextDetector(ckpt, input_size=1024, device=dev, precision=args.precision)
        join()
        out = mixed_stream(pkg, D, BK, det, rank, world, dev, args.steps, args.warmup, with_tail=True)
        if rank == 0:
            out.update({"n_gpus": n_gpus, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                        "data": "synthetic", "cpu_baseline": None})
            print(json.dumps(out), flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    masked_stream = None
    if args.cu_split > 0:
        # before any tail exists: their streams get the first T CUs; the forwards' stream the others
        Lb = importlib.import_module("comic-text-detector_amd._lib")
        TL.drain_free_tails()
        Lb.check(Lb.lib().ctd_tuning_set(b"tail_cus", int(args.cu_split)), "ctd_tuning_set tail_cus")
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        ncu = torch.cuda.get_device_properties(dev).multi_processor_count
        words = (ncu + 31) // 32
        bits = [0] * words
        for c in range(args.cu_split, ncu):
            bits[c >> 5] |= 1 << (c & 31)
        arr = (ctypes.c_uint32 * words)(*bits)
        sp = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), ctypes.c_uint32(words), arr)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed: {rc}")
        masked_stream = torch.cuda.ExternalStream(sp.value, device=dev)
    ckpt, batches, canned, canned_sample = make_workload(pkg, args, rank, nloc, dev)
    det = DET.TextDetector(ckpt, input_size=S, device=dev, precision=args.precision)
    # every mode times the WHOLE network (the seam's full contract: blks, mask f32, lines_map with both planes), as the
    # reference's `TextDetBase.forward` computes it; `TextDetector(trim_outputs=True)` is not benchmarked
    be = det.net
    e2e = args.mode == "e2e"
    pipe = Pipeline(det, batches, canned, dev, world, rank, total_pages, D, args.workers, args.depth, args.tail_split,
                    host_input=args.host_input and e2e, loaders=args.loaders, engines=args.engines,
                    keep_undetected=args.keep_undetected, lazy=bool(args.lazy_blocks), refine_mode=args.refine_mode,
                    force_dist=bool(args.force_dist))

    join()                                                   # the process group: after the pipeline's streams (see above)
    if args.fwd_stream == "high" and e2e:
        pipe.fwd_stream = torch.cuda.Stream(dev, priority=-1)
    if masked_stream is not None:
        pipe.fwd_stream = masked_stream
    if args.tail_only and e2e:
        fj = pipe.forward_job(0)
        torch.cuda.synchronize()
        fj["ev"] = None
        pipe.fixed_job = fj

    net_k = [0]

    def run_steps_net(n):
        for _ in range(n):
            x = batches[net_k[0] % len(batches)]
            net_k[0] += 1
            blks, _, _ = be.forward_u8(x)
            dets, counts = BK.nms(blks, 0.4, 0.35)
            if world > 1:
                D.gather_records(D.pack_records(dets, counts), total_pages, rank, world)

    run_steps = pipe.run if e2e else run_steps_net
    # Spin-up (untimed, before the W warm-up steps): a process that has just started measures transients, not the detector
    # -- the board leaves its low-power state over roughly the first second of load (rocm-smi: sclk 94 MHz idle) and the
    # host side (tail workers' buffers, pinned arenas, allocator) settles over the first dozens of batches.  A serving
    # process is in the second state; `config.spinup_steps` records it and `--spinup 0` gives the cold number.
    dt = timed(run_steps, args.steps, args.warmup, args.spinup, world, dev, pipe.stats)
    stats = dict(pipe.stats)
    # intervals between the timed steps' result deliveries: a 20-step region is 0.2 s, one late batch (a noisy host) is 5 % of it
    tf = list(pipe.tfin)[-args.steps:] if e2e else []
    gaps = sorted((b - a) * 1e3 for a, b in zip(tf, tf[1:]))
    delivery = ({"median_ms": round(gaps[len(gaps) // 2], 3), "max_ms": round(gaps[-1], 3), "min_ms": round(gaps[0], 3),
                 "steady_pages_per_s": round(nloc * world * 1e3 / gaps[len(gaps) // 2], 1)} if len(gaps) >= 3 else None)

    if rank == 0:
        # ---- one un-pipelined step: where a batch's time goes.  The tail runs on one of the pipeline's own worker threads
        # (no extra native tail / stream for the measurement: see tail._Lease), the second of two runs on that thread (the
        # first one sizes its 32-page buffers: the pipelined steps used page-range work items)
        def serial_tail(job_):                               # twice on ONE worker thread: the first run sizes its tail's buffers
            for _ in range(2):
                t0_ = time.perf_counter()
                det._tail(job_, 0, args.keep_undetected, lazy=True)
                ms_ = (time.perf_counter() - t0_) * 1e3
            return ms_, TL.thread_tail(dev).timings()
        torch.cuda.synchronize()
        ta = time.perf_counter()
        job = pipe.forward_job(0)
        torch.cuda.synchronize()
        tb_ = time.perf_counter()
        tail_ms, stages = pipe.pool.submit(serial_tail, job).result()
        serial = {"forward_ms": round((tb_ - ta) * 1e3, 3), "tail_ms": round(tail_ms, 3),
                  "tail_ms_per_page": round(tail_ms / nloc, 4), "tail_stages_ms": stages}
        roof = roofline_block(be, batches[0], args.precision, B, S, args.dump_ops)
        page0 = batches[0][0].cpu().numpy()
        cpu = parity = exact = extra = rocm = None
        solo = n_gpus == 1
        if solo and not args.no_cpu_baseline:
            cpu = cpu_baseline(pkg, ckpt, S, batches[0][:16].cpu().numpy(), canned_sample)
            try:
                parity = parity_block(pkg, ckpt, det, batches[0], 0, S)
            except Exception as e:                      # never lose the bench line to the extra check
                parity = {"error": repr(e)[:400]}
        real = canned is None
        if solo and e2e and not args.no_extras:
            # Sub-runs IN THIS PROCESS (`inproc_bench`: fresh checkpoint / pages / detector / worker pool each; the native
            # tails are leased from the pool before).  Round 3 ran them as child processes because a second pipeline in one
            # process measured 12-18 % low; the cause -- a second set of tail streams at the highest priority sharing
            # hardware queues with the first -- is gone.  Only the mixed-size stream and MIOpen are children.
            torch.cuda.synchronize()
            pipe.close()                                     # the headline's workers end: their tails go to the next pools
            TL.release_thread_tail()                         # ... and so does the one the parity leg used on this thread
            sub = lambda steps_, **ov: inproc_bench(pkg, D, DET, TL, args, dev, steps_, **ov)      # noqa: E731
            ex = "fp32s" if args.precision != "fp32s" else "fp32"
            c = sub(40, precision=ex)
            exact = dict(c, engine=ex, batch=B, workload="the headline's (same pages, checkpoint, pipeline), in this process",
                         acceptance="lines / blocks / refined mask identical to the oracle on the acceptance pages "
                                    "(tests/test_gpu_accept.py; `parity.engines` here)")
            extra = {}
            c = sub(40, precision="fp32", batch=8, tail_input="forward")
            extra["fp32_bs8_e2e"] = dict(c, config="BASELINE configs[1]: bs=8 1024x1024, fp32 (f32-operand MFMA engine), end to end "
                                                   "with the native tail")
            c = sub_bench(["--mode", "mixed", "--precision", args.precision], 3, warmup=1, spinup=0, timeout=240, whole=True)
            extra["mixed_e2e"] = c
            if real and not args.dense_blocks:
                c = sub(40, dense_blocks=True)
                extra["dense_blocks_e2e"] = dict(c, config="the headline's pages and pipeline on synth.make_blob_checkpoint(0) WITHOUT "
                                                           "sparse_det: every cell of one Detect anchor fires (random weights), NMS "
                                                           "packs the page with boxes (65 blocks / 67 lines per page)")
                c = sub(40, line_density="r3")
                extra["r3_density_e2e"] = dict(c, config="round 3's headline pages (sparse_det without the line-density calibration: 16 "
                                                         "blocks / 16 lines per page; round 3's driver line: 2586 pages/s)")
                c = sub(40, lazy_blocks=True)
                extra["lazy_blocklists_e2e"] = dict(c, config="the headline with the tail workers handing over lazily materialised "
                                                              "BlockLists (`detect_stream(lazy=True)`: the native records, "
                                                              "TextBlock objects built when a consumer looks at them) instead "
                                                              "of the reference's return type, a list of TextBlock objects per "
                                                              "page, which the headline builds (rounds 3-4 timed this variant "
                                                              "as the headline)")
                c = sub(40, refine_mode=1, keep_undetected=True)
                extra["reference_cli_config_e2e"] = dict(c, config="the headline with the reference CLI's arguments (inference.py:35, "
                                                                   "`model2annotations`): refine_mode = REFINEMASK_ANNOTATION, "
                                                                   "keep_undetected_mask = True (refine_undetected_mask runs too)")
                c = sub(40, host_input=True)
                extra["host_input_e2e"] = dict(c, config="the headline with the pages starting in HOST memory (numpy arrays, as the "
                                                         "reference's callers hand them over): PCIe-inclusive, never the headline")
            if real:
                c = sub(40, tail_input="canned")
                extra["canned_tail_inputs_e2e"] = dict(c, config="round 2's workload (`--tail-input canned`): random checkpoint, ONE "
                                                                 "resident batch, the tail fed text-like maps of the same pages "
                                                                 "instead of the forward's outputs (round 2 measured 2502 pages/s)")
            if args.rocm_timeout > 0:
                torch.cuda.synchronize()
                rocm = rocm_baseline_subprocess(args, args.rocm_timeout)
                try:
                    rocm["hip_forward"] = {"engine": args.precision, "ms_per_forward": roof["net_ms_per_step"], "batch": B}
                    if isinstance(exact, dict) and "net_ms_per_step" in exact:
                        rocm["hip_forward_exact"] = {"engine": exact["engine"], "ms_per_forward": exact["net_ms_per_step"], "batch": B}
                    for k, mine in (("fp16_bs32", roof["net_ms_per_step"]),
                                    ("fp32_bs32", exact.get("net_ms_per_step") if isinstance(exact, dict) else None)):
                        if mine and isinstance(rocm.get(k), dict) and "ms_per_forward" in rocm[k] and rocm[k].get("batch") == B:
                            rocm[k]["hip_speedup"] = round(rocm[k]["ms_per_forward"] / mine, 2)
                except Exception:
                    pass
        where = "in HOST memory (H2D inside the timed region)" if (e2e and args.host_input) else "resident in HBM"
        real = canned is None
        workload = (f"BASELINE configs[2]: bs={B}/GPU {S}x{S} u8 pages {where}, {len(batches)} distinct batches rotated; "
                    f"fused HIP forward (YOLOv5s+UNet+DB, "
                    + ("synth.make_blob_checkpoint(" + ("dense" if args.dense_blocks else
                       ("sparse_det + line_density=fixture: ~29 text lines per page, the LINE count of the reference's real "
                        "page (16 blocks / 29 lines); random weights cannot align lines into multi-line blocks, so blocks ~ "
                        "lines (~27 windows per page, more than the fixture's 16)" if args.line_density == "fixture" else
                        "sparse_det: round 3's pages, 16 blocks / 16 lines"))
                       + "): random weights whose maps have contours, text-like pages" if real else
                       "seeded random weights") + ", DB binarize + u8 mask fused)")
        if e2e:
            workload += (" + the WHOLE native tail per page: GPU NMS, DB boxes (2x GPU labelling + contour tables, host "
                         "geometry), mask crop, group_output, refine_mask"
                         + (", refine_undetected_mask" if args.keep_undetected else "")
                         + "; masks + TextBlock records delivered on the host; tail of step k on "
                         f"{args.workers} worker threads ({args.tail_split} page-range work items per batch) under the forward of "
                         "step k+1; "
                         + ("the tail consumes the timed forward's OWN outputs (one dependent chain, reference "
                            "inference.py:146-178)" if real else
                            "CANNED tail inputs: text-like network outputs of the same pages instead of the forward's "
                            "(random weights give noise maps)"))
        else:
            workload += " + GPU NMS (network-only mode)"
        if n_gpus > 1:
            workload += " + RCCL all-gather of the per-page block records"
        out = {
            "metric": f"pages/sec at {S}x{S} bs={B}" + ((" (TAIL ONLY: no forward in the step; host-side rehearsal" if args.tail_only
                                                          else " (end-to-end detector" + ("" if real else ", canned tail inputs"))
                                                         + ", device-resident pages)" if e2e else " (network + NMS)"),
            "value": round(total_pages * args.steps / dt, 2),
            "unit": "pages/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": DTYPE[args.precision],
            "data": "synthetic",
            "config": {"workload": workload, "mode": args.mode, "global_batch": total_pages, "page": [S, S],
                       "driver": ("TextDetector.detect_stream (the public API: bench.Pipeline only feeds it the resident batches "
                                  "and counts the results; workers / gc tuning are the product's thread_budget / serve_tuning)"
                                  if (e2e and pipe.api) else ("bench.Pipeline (a stage replaced: canned inputs / tail only)" if e2e
                                                              else "forward_u8 + nms")),
                       "refine_mode": args.refine_mode, "keep_undetected_mask": bool(args.keep_undetected),
                       "input": "nhwc_u8", "precision": args.precision, "tail_input": args.tail_input if e2e else None,
                       "distinct_batches": len(batches),
                       "tail_workers": args.workers if e2e else 0, "batches_in_flight": args.depth if e2e else 1,
                       "engines": args.engines if e2e else 1,
                       "spinup_steps": args.spinup, "tail_split": args.tail_split if e2e else 1,
                       "host_threads": tb, "cpu_affinity": cpu_affinity,
                       "pages_start_in": "host memory (pinned staging + async H2D on %d loader threads)" % args.loaders
                                         if (e2e and args.host_input) else "HBM",
                       "blocks_per_page": round(stats["blocks"] / max(stats["pages"], 1), 2) if e2e else None,
                       "lines_per_page": round(stats["lines"] / max(stats["pages"], 1), 2) if e2e else None,
                       "host_cpu_cores_used": round(float(stats.get("cpu_cores", 0.0)), 2),
                       "result_delivery_intervals": delivery,
                       "record_gather_ms_per_step": round(float(stats.get("gather_s", 0.0)) * 1e3 / max(args.steps, 1), 3),
                       "one_device_rehearsal": one_device, "tail_only": bool(args.tail_only and e2e),
                       "parallelism": f"dp{n_gpus} (pages sharded, no data-path collective except the final record "
                                      f"gather; ranks={world}, backend={dist.get_backend() if dist.is_initialized() else 'none'}"
                                      + (", world-size-1 group FORCED (--force-dist): the record gather of the N > 1 step runs on "
                                         "this backend inside every step" if (args.force_dist and world == 1) else "")
                                      + (f", each rank bound to {cpu_affinity['n_cpus']} CPUs of its GPU's NUMA node" if pinned else "")
                                      + (", ALL RANKS ON ONE DEVICE: rehearsal, not a measurement" if one_device else "") + ")"},
            "serial_step": serial,
            "roofline": roof,
            "cpu_baseline": cpu,
            "parity": parity,
            "parity_exact": exact,
            "extra_configs": extra,
            "rocm_baseline": rocm,
        }
        print(json.dumps(out), flush=True)
    pipe.close()                                             # (idempotent)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
