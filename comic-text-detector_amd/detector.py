"""`TextDetector`: mirror of the reference's L4 API (reference inference.py:116-178)
on top of the HIP backend and the native tail.

    det = TextDetector(model_path_or_ckpt, input_size=1024, device='cuda')
    mask, mask_refined, blk_list = det(img_bgr_uint8, refine_mode, keep_undetected_mask)

Same call signature and return triple as the reference; same constructor keywords.  Deviation of the
defaults: `device='cuda'` (there is no CPU path; the reference defaults to 'cpu').  `half=False` is the
reference's default AND its behaviour (its `TextDetBase` always runs fp32, inference.py:129): the
exact-fp32 engine; `half=True` selects the fp16-operand / fp32-accumulate MFMA engine (BASELINE
configs[2]) whose maps differ from fp32 by <1e-3 (see DESIGN.md section 5 for the measured effect on
boxes and masks).

`detect_batch(pages)` is the batched form the reference lacks (it is bs=1 only): one
fused forward and ONE native tail call for the whole batch.  `detect_stream(batches)` pipelines batches:
the tail of batch k runs on worker threads (own HIP stream, interpreter lock released) under the
forward of batch k+1.
"""
from __future__ import annotations

from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, List, Sequence, Tuple, Union

import ctypes as C
import gc
import os
import threading

import numpy as np
import torch

from . import _lib as L
from . import backend as BK
from . import postproc as PP
from .tail import bind_thread, thread_tail
from .textblock import TextBlock
from .textmask import (REFINEMASK_ANNOTATION, REFINEMASK_INPAINT, refine_mask, refine_mask_batch,   # noqa: F401
                       refine_undetected_mask)

__all__ = ["TextDetector", "TextBlock", "REFINEMASK_INPAINT", "REFINEMASK_ANNOTATION", "thread_budget", "serve_tuning"]


def usable_cpus() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cgroup_cpu_quota() -> float:
    """CPUs' worth of time the container's CFS quota grants per scheduling period (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`); 0.0 = no quota.  The affinity mask does not show it: a 1-GPU box of the pool this
    was built on reports 256 usable CPUs under `cpu.max = 1600000 100000` (16 CPUs), and a process group that runs past the
    quota is frozen for the rest of the 100-ms period -- ten batches of this pipeline."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return 0.0 if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return q / p if q > 0 and p > 0 else 0.0
    except (OSError, ValueError):
        return 0.0


def thread_budget(world: int = 1, pinned: bool = False, host_cpus: int = 0) -> dict:
    """Host threads one rank may use: the usable cores divided by the ranks on this host (an 8-rank node runs 8 of these
    processes) -- or, once the rank is bound to its own CPUs (`affinity.apply`, N > 1), simply the CPUs it is bound to --
    and never more than twice the rank's share of the container's CPU quota (`cgroup_cpu_quota`).
    Tail workers x native geometry threads + loaders + the launching thread must fit."""
    avail = usable_cpus()
    share = avail if pinned else avail // max(1, world)
    quota = cgroup_cpu_quota()
    if quota > 0:
        # threads up to TWICE the quota's share: the tail's threads are bursty (6-7 busy cores of 34 threads on the headline), and
        # sizing them on the quota itself -- 3 geometry threads per worker under 16 CPUs instead of 8 -- left the headline where
        # it was (3086-3110 pages/s) but cost the dense pages 9-13 % (2517-2690 against 2904-2988; profiles/r06_cpu_quota.txt)
        share = min(share, 2 * int(quota) // max(1, world))
    per_rank = max(4, share)
    if pinned and host_cpus:
        avail = host_cpus                                     # what the host offered before this rank bound itself
    # 4 workers where a rank has 16 CPUs or more: round 5 re-measured it after the forward got shorter -- on one box 3 = 4 on the
    # headline and +6 % on the canned pages, on another 4 is +2 % on the headline and +6 % on dense pages (three interleaved
    # repetitions each, profiles/r05_e2e_workers_3_vs_4.txt); the dense pages decide
    workers = 4 if per_rank >= 16 else (3 if per_rank >= 8 else 2)
    # geometry threads per worker: a power of two.  A work item is 8 pages (32 / `tail_split` 4) and the native per-page loops hand
    # pages to threads one at a time: 7 threads finish 8 pages in two rounds like 4 do -- which is what a 32-CPU share (one of 8
    # ranks on a 256-CPU host, or twice a 16-CPU quota) got from `(per_rank - 2) // workers` = 7: dense pages 2600 against
    # 2904-2988 with 8 (profiles/r06_cpu_quota.txt).  The launching thread and the loaders are idle most of the time and are
    # not subtracted.
    cap = max(1, min(8, per_rank // workers))
    native = 1 << (cap.bit_length() - 1)
    out = {"usable_cpus": avail, "per_rank": per_rank, "tail_workers": workers, "native_threads_per_worker": native}
    if quota > 0:
        out["cgroup_cpu_quota"] = round(quota, 2)
    return out


_tuned = False


def serve_tuning(refreeze: bool = False) -> bool:
    """What a serving process does ONCE after start-up; `detect_stream` calls it after its pools are warm (opt out with
    `tune=False`), `bench.py` calls it at the same place of its timed region -- so a caller of the API gets the process the
    benchmark measures.  (1) `gc.freeze()`: the interpreter's cyclic collector otherwise re-scans the ~1M long-lived objects
    of torch / numpy whenever the per-page result objects trigger a full collection (~10 ms per batch).  (2) the youngest
    generation's threshold 700 -> 50 000: a batch's results (32 pages x 30 TextBlocks, each a dict, a few lists and numpy
    values) are ~10 k tracked containers, at the default threshold the collector runs a dozen times per batch on the tail
    workers under the interpreter lock; nothing on this path creates reference cycles.  Returns whether it acted
    (`refreeze=True`: again, for a process that built new long-lived objects since)."""
    global _tuned
    if _tuned and not refreeze:
        return False
    gc.collect()
    gc.freeze()
    gc.set_threshold(50000, 20, 20)
    _tuned = True
    return True


Page = Union[np.ndarray, torch.Tensor]


class TextDetector:
    lang_list = ["eng", "ja", "unknown"]                      # inference.py:117
    langcls2idx = {"eng": 0, "ja": 1, "unknown": 2}

    def __init__(self, model_path: Union[str, dict], input_size=1024, device="cuda", half=False,
                 nms_thresh=0.35, conf_thresh=0.4, mask_thresh=0.3, act="leaky", trim_outputs=False,
                 precision: str = None):
        if isinstance(input_size, int):
            input_size = (input_size, input_size)
        self.input_size = input_size
        self.device = device
        self.half = half
        self.conf_thresh = conf_thresh
        self.nms_thresh = nms_thresh
        # `precision` (an addition to the reference's keywords) names the engine directly and overrides `half`:
        #   "fp32"  f32-operand MFMA (an fmaf chain's arithmetic)        = half=False, the default
        #   "fp32s" fp32 tensors, split-operand products on the fp16 MFMA (same results to ~1e-6, several x faster)
        #   "fp16"  fp16 tensors and operands, fp32 accumulate           = half=True
        if precision is None:
            precision = "fp16" if half else "fp32"
        if precision not in ("fp32", "fp32s", "fp16"):
            raise ValueError("precision must be 'fp32', 'fp32s' or 'fp16'")
        self.precision = precision
        self._net_args = dict(model=model_path, device=device, precision=precision, act=act,
                              bitmap_thresh=0.3, outputs="detector" if trim_outputs else "all")
        # trim_outputs=True: the engine computes only what this class consumes (no DB threshold branch, no f32 mask
        # plane; same results, ~1.5 % less GPU time).  Off by default -- the reference's network computes both, and
        # the benchmarked step is the whole network.
        self.net = BK.HipTextDetBackend(**self._net_args)
        self._lanes = [(self.net, None)]                      # (engine, stream) pairs of detect_stream, grown on demand
        self._stage_tl = threading.local()
        self._copy_st, self._copy_lock = None, threading.Lock()
        self._pools = {}                                      # detect_stream's worker / loader pools, kept between calls
        self._warmed = None                                   # the tail pool whose threads already hold their native tails
        self.backend = "hip"
        self.seg_rep = PP.SegRepresenter(thresh=0.3)          # inference.py:139

    # -- preprocessing: `preprocess_img` + `letterbox` (inference.py:72-83, imgproc_utils.py:86-117).
    #    The aspect-keeping bilinear resize and the bottom/right zero padding run on the GPU
    #    (ctd_resize_linear_u8); the /255 and the layout change are fused into the stem kernel.
    #    Channel order: BGR2RGB (:74) followed by [::-1] (:77) = the net consumes BGR planes,
    #    and the per-channel resize commutes with the swaps, so the BGR page is resized as is.
    def _prepare(self, pages: Sequence[Page]):
        Hn, Wn = self.input_size[1], self.input_size[0]
        dev = self.net.device
        gpu, metas = [], []
        for p in pages:
            if isinstance(p, torch.Tensor):
                if p.dtype != torch.uint8 or p.dim() != 3 or p.shape[2] != 3:
                    raise ValueError("pages must be uint8 BGR (H,W,3)")
                src = p.to(dev).contiguous()
            else:
                if p.dtype != np.uint8 or p.ndim != 3 or p.shape[2] != 3:
                    raise ValueError("pages must be uint8 BGR (H,W,3) arrays")
                src = torch.from_numpy(np.ascontiguousarray(p)).to(dev)
            im_h, im_w = src.shape[:2]
            r = min(Hn / im_h, Wn / im_w)
            nw, nh = int(round(im_w * r)), int(round(im_h * r))
            gpu.append(src)
            metas.append((im_h, im_w, int(Wn - nw), int(Hn - nh)))
        if all(m == (Hn, Wn, 0, 0) for m in metas):             # nothing to resize (letterbox: shape == new_unpad)
            x = self._as_batch(gpu)
        else:
            x = torch.stack([g if m == (Hn, Wn, 0, 0) else BK.resize_linear_u8(g, (Hn - m[3], Wn - m[2]), (Hn, Wn))
                             for g, m in zip(gpu, metas)])
        return x, gpu, metas

    @staticmethod
    def _as_batch(pages: Sequence[torch.Tensor]) -> torch.Tensor:
        """Equal-size pages as ONE (B,H,W,3) tensor -- WITHOUT a copy when they already lie back to back in one
        allocation (pages uploaded by `_stage`, slices of a caller's batch tensor): `torch.stack` of 32 1024x1024 pages
        is a 200 MB round trip through HBM per batch (0.12 ms of GPU time, rocprofv3 `CatArrayBatchedCopy`)."""
        p0 = pages[0]
        n = p0.numel()
        try:
            same = all(g.shape == p0.shape and g.dtype == p0.dtype and g.is_contiguous() and
                       g.untyped_storage().data_ptr() == p0.untyped_storage().data_ptr() and
                       g.storage_offset() == p0.storage_offset() + i * n for i, g in enumerate(pages))
        except (RuntimeError, AttributeError):
            same = False
        if same and p0.is_contiguous():
            return torch.as_strided(p0, (len(pages),) + tuple(p0.shape), (n,) + tuple(p0.stride()), p0.storage_offset())
        return torch.stack(list(pages))

    def _forward(self, pages: Sequence[Page], net=None):
        net = net or self.net
        if not all(isinstance(p, torch.Tensor) and p.is_cuda for p in pages):     # host pages: one pinned copy
            pages, ev = self._stage(pages)
            torch.cuda.current_stream(net.device).wait_event(ev)
        x, gpu, metas = self._prepare(pages)
        blks, mask, lines_map = net.forward_u8(x)                           # the seam (inference.py:146)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(net.device))
        return dict(gpu=gpu, metas=metas, blks=blks, mask_u8=net.mask_u8, lines_map=lines_map,
                    bitmap=net.bitmap, ev=ev, keep=(mask, x))

    # -- host pages -> HBM (the reference hands numpy images to `__call__`): one pinned staging buffer and ONE
    #    async copy per batch, on a copy stream, from loader threads -- so that PCIe runs under the forward of
    #    earlier batches instead of in front of every forward (per-page pageable copies: ~6 ms per 32 pages).
    def _stage(self, pages: Sequence[Page]):
        """Runs on a loader thread.  Returns (pages as device tensors, event to wait for) -- device tensors pass through."""
        dev = self.net.device
        if all(isinstance(p, torch.Tensor) and p.is_cuda for p in pages):
            return list(pages), None
        bind_thread(dev)
        tl = self._stage_tl                                   # per loader thread: pinned ring + copy stream
        if not hasattr(tl, "ring"):
            tl.ring, tl.k, tl.stream = [], 0, self._copy_stream()
        arrs = []
        for p in pages:
            a = p.cpu().numpy() if isinstance(p, torch.Tensor) else np.asarray(p)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("pages must be uint8 BGR (H,W,3) arrays")
            arrs.append(a)
        total = sum(a.size for a in arrs)
        if len(tl.ring) < 3:                                  # a loader's batches in flight: being filled, copying, in use
            tl.ring.append(dict(buf=None, ev=None))
        slot = tl.ring[tl.k % len(tl.ring)]
        tl.k += 1
        if slot["ev"] is not None:
            slot["ev"].synchronize()                          # the copy that last read this buffer
        if slot["buf"] is None or slot["buf"].numel() < total:
            slot["buf"] = torch.empty((total,), dtype=torch.uint8).pin_memory()
        # one C call gathers the pages into the pinned buffer: ctypes drops the interpreter lock for it (a numpy
        # slice assignment held it: +5 ms per batch of 32 pages for every other thread of the pipeline; torch's
        # copy_ fights over the intra-op thread pool: 10x worse)
        arrs = [np.ascontiguousarray(a) for a in arrs]
        n = len(arrs)
        srcs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        sizes = (C.c_size_t * n)(*[a.size for a in arrs])
        L.check(L.lib().ctd_host_gather(slot["buf"].data_ptr(), srcs, sizes, n, 4), "ctd_host_gather")
        off, views = 0, []
        for a in arrs:
            views.append((off, a.shape))
            off += a.size
        with torch.cuda.stream(tl.stream):
            d = slot["buf"][:total].to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(tl.stream)
        slot["ev"] = ev
        return [d[o: o + h * w * 3].view(h, w, 3) for o, (h, w, _) in views], ev

    def _copy_stream(self):
        """ONE upload stream per detector, shared by its loader threads (uploads are serial on the link anyway), created on
        first use -- which `warm_tails` arranges to be AFTER the tail workers' streams exist.  Stream count, priority class
        and creation order matter on this stack: streams beyond the runtime's hardware queues of their class share a queue
        with an earlier one (tail._Lease; with the tails at the highest priority, bench `--host-input` as a fresh process
        ran at 2025 pages/s with two loader streams created before the tails' and 2380 this way round; with the tails at
        the default priority it runs at 2500, 2 % under device-resident pages)."""
        with self._copy_lock:
            if self._copy_st is None:
                self._copy_st = torch.cuda.Stream(self.net.device)
            return self._copy_st

    def warm_tails(self, pool: ThreadPoolExecutor, workers: int) -> None:
        """Makes every worker thread of `pool` take (lease or create) its native tail now, before any loader stream exists."""
        import threading as _th
        gate = _th.Barrier(max(1, int(workers)))

        def grab():
            thread_tail(self.net.device)
            try:
                gate.wait(timeout=10.0)                       # keep `workers` distinct threads busy at once
            except _th.BrokenBarrierError:
                pass
        for f in [pool.submit(grab) for _ in range(max(1, int(workers)))]:
            f.result()

    def _lane(self, i: int):
        """Engine + stream number i of `detect_stream`.  Every lane is a whole engine (its own arena) on its own
        HIP stream, so consecutive batches run concurrently: one batch's HBM-bound layers fill the gaps of the
        other's MFMA-bound ones (scripts/gpu_dual.py: 10.86 -> 10.28 ms per 32 pages on the network alone).  End
        to end the tail's kernels already fill those gaps and a second lane LOSES 10 % (bench.py --engines 2), so
        `detect_stream` defaults to one lane; the option is for network-only serving."""
        while len(self._lanes) <= i:
            self._lanes.append((BK.HipTextDetBackend(**self._net_args), None))
        net, st = self._lanes[i]
        if st is None:
            st = torch.cuda.Stream(net.device)
            self._lanes[i] = (net, st)
        return net, st

    def _tail(self, job, refine_mode, keep_undetected_mask, lo=None, hi=None, records=None, lazy=False):
        """The native tail of pages [lo, hi) of a forwarded batch (default: all of it) on the calling thread's `Tail`.
        records=(cap_blk, cap_line): the pages come back as `PageResult`s carrying their multi-GPU gather records."""
        sl = slice(lo, hi)
        return thread_tail(self.net.device).run(job["gpu"][sl], job["metas"][sl], job["blks"][sl], job["mask_u8"][sl],
                                                job["lines_map"][sl], job["bitmap"][sl], self.conf_thresh, self.nms_thresh,
                                                0.6, True, refine_mode, keep_undetected_mask, job["ev"], records=records,
                                                lazy=lazy)

    @staticmethod
    def _split(n: int, parts: int):
        """Page ranges of a batch's tail work items: `parts` near-equal contiguous pieces (pages are independent after
        the forward, reference inference.py:148-178 is per page)."""
        parts = max(1, min(int(parts), n))
        return [(n * k // parts, n * (k + 1) // parts) for k in range(parts)]

    @torch.no_grad()
    def detect_batch(self, pages: Sequence[Page], refine_mode=REFINEMASK_INPAINT,
                     keep_undetected_mask=False) -> List[Tuple[np.ndarray, np.ndarray, List[TextBlock]]]:
        return self._tail(self._forward(pages), refine_mode, keep_undetected_mask)

    @torch.no_grad()
    def detect_stream(self, batches: Iterable[Sequence[Page]], refine_mode=REFINEMASK_INPAINT,
                      keep_undetected_mask=False, workers: int = 0, depth: int = 4, engines: int = 1,
                      loaders: int = 2, tail_split: int = 0, lazy: bool = False, records=None,
                      tune: bool = True) -> Iterator[list]:
        """Yields `detect_batch(batch)` for every batch, in order, with up to `depth` batches in flight:
        the forward of the next batches is launched while `workers` threads run the tails of earlier ones.
        Host (numpy) pages are staged to the GPU by `loaders` threads up to `depth` batches ahead (`_stage`).
        `engines` > 1 alternates the batches over that many engine copies on their own streams (`_lane`).
        `tail_split` cuts every batch's tail into that many page ranges, each a work item of its own for the workers
        (0 = one per worker): lower latency per batch and a shorter drain when the stream ends, for more, smaller
        native calls -- measured +6 % end to end at 32 pages per batch (2311 -> 2456 pages/s, 3 workers).
        `lazy=False` (default): every page's blk_list is the reference's return type, a plain list of `TextBlock`s, built
        on the worker threads as `detect_batch` does.  `lazy=True`: a `textblock.BlockList` instead -- the native records,
        complete on the host, which builds the `TextBlock` objects when first iterated / indexed, on the consumer's thread
        (not a `list`: no `append` / `sort` / `json.dumps`); for consumers that read the columnar records or only a few
        pages' blocks it saves the workers ~0.1 ms of interpreter-lock time per page (+2 % end to end at 30 blocks a page).
        `workers=0` (default): from the host's thread budget (`thread_budget`: 4 where the process has 16 CPUs, else 3 / 2), and
        the native per-page geometry threads of every worker from the same budget unless `tail.set_host_threads` was called.
        `tune=True`: `serve_tuning()` once per process, after the pools are warm.  `records=(cap_blk, cap_line)`: every page
        comes back as a `PageResult` carrying its multi-GPU gather record (`dist.gather_results`).
        This generator IS what `bench.py`'s headline times (its `Pipeline` only feeds it batches and counts the results).
        The pools stay alive between calls (`close()` stops them): each worker thread keeps a native tail object with a
        HIP stream and ~250 MB of device tables at 32 pages per batch."""
        if int(workers) <= 0:
            tb = thread_budget()
            workers = tb["tail_workers"]
            from . import tail as _TL
            if _TL._host_threads is None:
                _TL.set_host_threads(tb["native_threads_per_worker"])
        # The pools live on the detector: their threads own the native `Tail` objects (a HIP stream, ~250 MB of
        # fixed-capacity device tables at 32 pages, pinned buffers) and the pinned staging rings, which a pool per call
        # would create and destroy every time.
        pool = self._pool("tail", workers)
        if self._warmed is not pool:                          # once per pool: its threads then hold their leases
            self.warm_tails(pool, workers)                    # tail streams first, the upload stream after them
            self._warmed = pool
        if tune:
            serve_tuning()
        lpool = self._pool("load", loaders)
        pending = deque()
        engines = max(1, int(engines))
        tail_split = int(tail_split) if int(tail_split) > 0 else max(1, workers)
        main = torch.cuda.current_stream(self.net.device)

        def staged():
            ahead, it = deque(), iter(batches)
            for batch in it:
                ahead.append(lpool.submit(self._stage, batch))
                if len(ahead) > max(1, depth):
                    yield ahead.popleft().result()
            while ahead:
                yield ahead.popleft().result()

        try:
            for i, (batch, ev) in enumerate(staged()):
                if ev is not None:
                    main.wait_event(ev)
                if engines == 1:
                    job = self._forward(batch)
                else:
                    net, st = self._lane(i % engines)
                    st.wait_stream(main)                  # pages the caller produced on its stream
                    with torch.cuda.stream(st):
                        job = self._forward(batch, net)
                pending.append([pool.submit(self._tail, job, refine_mode, keep_undetected_mask, lo, hi, records, lazy)
                                for lo, hi in self._split(len(job["metas"]), tail_split)])
                while len(pending) >= depth:
                    yield [r for f in pending.popleft() for r in f.result()]
            while pending:
                yield [r for f in pending.popleft() for r in f.result()]
        finally:
            for futs in pending:                          # the consumer stopped early: let the queued tails finish
                for f in futs:
                    f.cancel() or f.exception()

    def _pool(self, kind: str, n: int) -> ThreadPoolExecutor:
        n = max(1, int(n))
        cur = self._pools.get(kind)
        if cur is None or cur[1] != n:
            if cur is not None:
                cur[0].shutdown(wait=True)
            cur = (ThreadPoolExecutor(max_workers=n, thread_name_prefix=f"ctd-{kind}"), n)
            self._pools[kind] = cur
        return cur[0]

    def close(self, drain: bool = False) -> None:
        """Stops `detect_stream`'s worker threads.  Their pinned staging rings go with them; their native tails (a HIP
        stream, ~250 MB of device tables at 32 pages per batch, pinned result buffers each) go back to the module's free
        list for the next pipeline of this process to take over (`tail._Lease`: reuse keeps the process's stream count
        inside the hardware queues) -- they are NOT freed unless `drain=True` (`tail.drain_free_tails`: for a process
        that is done detecting on this device)."""
        for ex, _ in self._pools.values():
            ex.shutdown(wait=True)
        self._pools = {}
        self._warmed = None
        if drain:
            from .tail import drain_free_tails
            import gc
            gc.collect()                                      # ended threads' thread-local leases return their tails
            drain_free_tails(self.net.device)

    def __del__(self):
        try:
            for ex, _ in getattr(self, "_pools", {}).values():
                ex.shutdown(wait=False)
        except Exception:
            pass

    def tail_batch(self, pages: Sequence[Page], blks: torch.Tensor, mask_u8: torch.Tensor, prob: torch.Tensor,
                   bitmap: torch.Tensor, refine_mode=REFINEMASK_INPAINT, keep_undetected_mask=False, metas=None,
                   want_extras: bool = False):
        """Everything after the network (inference.py:148-178) for a batch whose network outputs are
        already on the GPU: blks (B,rows,no) f32, mask_u8 (B,H,W) u8, prob = lines_map[:,0] (B,H,W) f32,
        bitmap (B,H,W) u8.  metas[b] = (im_h, im_w, dw, dh) of the letterbox (default: no resize)."""
        dev = self.net.device
        gpu = [p.to(dev).contiguous() if isinstance(p, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(p)).to(dev)
               for p in pages]
        if metas is None:
            metas = [(g.shape[0], g.shape[1], 0, 0) for g in gpu]
        torch.cuda.current_stream(dev).synchronize()
        return thread_tail(dev).run(gpu, metas, blks, mask_u8, prob, bitmap, self.conf_thresh, self.nms_thresh, 0.6, True,
                                    refine_mode, keep_undetected_mask, None, want_extras)

    def __call__(self, img: np.ndarray, refine_mode=REFINEMASK_INPAINT, keep_undetected_mask=False):
        return self.detect_batch([img], refine_mode, keep_undetected_mask)[0]
