"""`Tail`: Python handle of the native detector tail (`ctd_tail_*`, csrc/tail.hip) -- everything
`TextDetector.__call__` does after `self.net(img_in)` (reference inference.py:148-178) for a batch of
pages in ONE native call: NMS, DB boxes, mask crop / resize, `group_output`, `refine_mask`,
`refine_undetected_mask`.  The call releases the interpreter lock, owns its HIP stream and buffers, and
different `Tail` objects may run on different threads (`TextDetector.detect_stream` overlaps the tail of
batch k with the network forward of batch k+1 that way).  This module only marshals arguments and turns
the native records into the reference's Python objects.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from .textblock import BLK_DTYPE, BlockList, TextBlock, blocks_from_batch, blocks_from_records


def _pinned_u8(h: int, w: int) -> np.ndarray:
    return torch.empty((h, w), dtype=torch.uint8, pin_memory=True).numpy()


def _pinned_pages(shapes) -> List[np.ndarray]:
    """One page-locked allocation for a batch of u8 pages, handed out as per-page views."""
    sizes = [h * w for h, w in shapes]
    buf = torch.empty((sum(sizes),), dtype=torch.uint8, pin_memory=True).numpy()
    out, off = [], 0
    for (h, w), n in zip(shapes, sizes):
        out.append(buf[off: off + n].reshape(h, w))
        off += n
    return out


class PageResult(tuple):
    """(mask, mask_refined, blk_list) of one page -- the reference's return triple -- that may also carry `.record`, the
    page's fixed-capacity f64 block record for the multi-GPU gather (dist.pack_results), built natively."""
    record = None


_host_threads = None            # native threads per Tail for its per-page / per-window host loops (None: library default)


def set_host_threads(n: int) -> None:
    """Host threads every `Tail` created from now on may use inside a native call (`ctd_tail_set_threads`).  One process
    per GPU on a shared host: usable cores / ranks / tail workers."""
    global _host_threads
    _host_threads = max(1, int(n))


class Tail:
    def __init__(self, device: torch.device):
        self._lib = L.lib()
        self.device = torch.device(device)
        h = C.c_void_p()
        L.check(self._lib.ctd_tail_create(C.byref(h), self.device.index or 0), "ctd_tail_create")
        self._h = h
        if _host_threads is not None:
            L.check(self._lib.ctd_tail_set_threads(h, _host_threads), "ctd_tail_set_threads")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.ctd_tail_destroy(h)
            except Exception:
                pass
            self._h = None

    # -- helpers --------------------------------------------------------------------------------
    @staticmethod
    def _page_table(pages_gpu: Sequence[Optional[torch.Tensor]], metas):
        n = len(metas)
        tab = (L.CtdTailPage * n)()
        for b, (im_h, im_w, dw, dh) in enumerate(metas):
            pg = pages_gpu[b] if pages_gpu is not None else None
            if pg is not None:
                if not pg.is_cuda or pg.dtype != torch.uint8 or tuple(pg.shape) != (im_h, im_w, 3) or not pg.is_contiguous():
                    raise ValueError("pages must be contiguous uint8 (im_h, im_w, 3) GPU tensors")
                tab[b].img_dev = pg.data_ptr()
            tab[b].im_h, tab[b].im_w, tab[b].dw, tab[b].dh = int(im_h), int(im_w), int(dw), int(dh)
        return tab

    def _blocks(self, b: int, want_extras: bool = False) -> Tuple[List[TextBlock], Optional[dict]]:
        lib = self._lib
        nb, nl, nd, nx, ny = (C.c_int32() for _ in range(5))
        L.check(lib.ctd_tail_page_counts(self._h, b, C.byref(nb), C.byref(nl), C.byref(nd), C.byref(nx), C.byref(ny)),
                "ctd_tail_page_counts")
        recs = (L.CtdBlk * max(nb.value, 1))()
        lines = np.empty((nl.value, 8), np.int32)
        dist = np.empty((nd.value, 3), np.float64)
        if not want_extras:
            L.check(lib.ctd_tail_page_fetch(self._h, b, recs, lines.ctypes.data, dist.ctypes.data, None, None, None, None,
                                            None), "ctd_tail_page_fetch")
            return blocks_from_records(recs, lines, dist, nb.value), None
        boxes = np.empty((nx.value, 4, 2), np.int16)
        scores = np.empty((nx.value,), np.float32)
        yx = np.empty((ny.value, 4), np.int32)
        yc = np.empty((ny.value,), np.int32)
        yf = np.empty((ny.value,), np.float32)
        L.check(lib.ctd_tail_page_fetch(self._h, b, recs, lines.ctypes.data, dist.ctypes.data, boxes.ctypes.data,
                                        scores.ctypes.data, yx.ctypes.data, yc.ctypes.data, yf.ctypes.data),
                "ctd_tail_page_fetch")
        extras = {"db_boxes": boxes, "db_scores": scores, "yolo": (yx, yc, np.round(yf, 3))}
        return blocks_from_records(recs, lines, dist, nb.value), extras

    def _block_lists(self, B: int, lazy: bool = True):
        """The grouped blocks of every page of the last run: two native calls and three arrays for the batch; `lazy`:
        per-page views of them wrapped as `BlockList`s (no per-block Python here), else the reference's lists of
        `TextBlock`s, built for the whole batch in one conversion (`textblock.blocks_from_batch`)."""
        lib = self._lib
        cnt = np.empty((B, 5), np.int32)
        L.check(lib.ctd_tail_batch_counts(self._h, cnt.ctypes.data), "ctd_tail_batch_counts")
        nb, nl, nd = (int(v) for v in cnt[:, :3].sum(0))
        recs = np.empty((max(nb, 1),), BLK_DTYPE)
        lines = np.empty((max(nl, 1), 8), np.int32)
        dist = np.empty((max(nd, 1), 3), np.float64)
        L.check(lib.ctd_tail_batch_fetch(self._h, recs.ctypes.data, lines.ctypes.data, dist.ctypes.data), "ctd_tail_batch_fetch")
        if not lazy:
            return blocks_from_batch(recs, lines, dist, cnt[:, :3])
        ob = np.concatenate(([0], np.cumsum(cnt[:, 0]))).tolist()
        ol = np.concatenate(([0], np.cumsum(cnt[:, 1]))).tolist()
        od = np.concatenate(([0], np.cumsum(cnt[:, 2]))).tolist()
        return [BlockList(recs[ob[b]: ob[b + 1]], lines[ol[b]: ol[b + 1]], dist[od[b]: od[b + 1]]) for b in range(B)]

    # -- the whole tail ---------------------------------------------------------------------------
    def run(self, pages_gpu: Sequence[torch.Tensor], metas, blks: torch.Tensor, mask_u8: torch.Tensor,
            lines_map: torch.Tensor, bitmap: torch.Tensor, conf_thresh=0.4, nms_thresh=0.35, box_thresh=0.6,
            refine: bool = True, refine_mode: int = 0, keep_undetected_mask: bool = False,
            ready_event: Optional[torch.cuda.Event] = None, want_extras: bool = False, records=None,
            lazy: bool = False, pinned: bool = True):
        """metas[b] = (im_h, im_w, dw, dh); blks (B,rows,no) f32, mask_u8 (B,Hn,Wn) u8, lines_map (B,2,Hn,Wn) f32
        or its plane 0 (B,Hn,Wn), bitmap (B,Hn,Wn) u8 -- all on the GPU.  Returns per page
        (mask, mask_refined, blk_list[, extras]) as the reference's `TextDetector.__call__` does.
        records=(cap_blk, cap_line): every page's tuple is a `PageResult` whose `.record` is its gather record.
        lazy: blk_list is a `BlockList` (the native records; `TextBlock` objects are built when first accessed, by whoever
        accesses them) instead of a list built here -- what `detect_stream`'s workers return.
        pinned=False: the result arrays are ordinary (pageable) numpy arrays, as a C caller's malloc'ed buffers would be --
        `ctd_hip.h` only asks for host memory; the native side then downloads with hipMemcpyAsync instead of the copy
        kernel / DMA into page-locked memory (slower; exists so that the contract is tested)."""
        B = len(metas)
        for tns in (blks, mask_u8, lines_map, bitmap):
            if not tns.is_cuda:
                raise L.CtdError("Tail.run: the network outputs must live on the GPU")
        bind_thread(self.device)
        Hn, Wn = mask_u8.shape[-2:]
        # The native tail reads these buffers on ITS stream and waits for `ready_event` only.  A layout / dtype
        # conversion here runs on torch's current stream AFTER that event: order the tail behind it with a fresh event.
        need = not (blks.is_contiguous() and mask_u8.is_contiguous() and bitmap.is_contiguous()) or \
            lines_map.dtype != torch.float32 or lines_map.stride(-1) != 1 or lines_map.stride(-2) != Wn
        if need:
            # The conversion kernels run on THIS thread's current stream, which has not necessarily seen the forward
            # (a `detect_stream` lane, a caller's side stream): make it wait for the producer's event first, then hand
            # the native tail a fresh event recorded behind the conversions.
            cur = torch.cuda.current_stream(self.device)
            if ready_event is not None:
                cur.wait_event(ready_event)
            blks = blks.contiguous()
            mask_u8 = mask_u8.contiguous()
            bitmap = bitmap.contiguous()
            if lines_map.dtype != torch.float32 or lines_map.stride(-1) != 1 or lines_map.stride(-2) != Wn:
                lines_map = lines_map.float().contiguous()
            ready_event = torch.cuda.Event()
            ready_event.record(cur)
        prob_stride = lines_map.stride(0)
        tab = self._page_table(pages_gpu, metas)
        prm = L.CtdTailParams(conf_thresh, nms_thresh, box_thresh, 1000, 1.5, int(bool(refine)), int(refine_mode),
                              int(bool(keep_undetected_mask)), 0)
        # page-locked result arrays (torch's caching host allocator): the tail DMAs straight into them
        if pinned:
            masks = _pinned_pages([(m[0], m[1]) for m in metas])
            refined = _pinned_pages([(m[0], m[1]) for m in metas]) if refine else [None] * B
        else:
            masks = [np.empty((m[0], m[1]), np.uint8) for m in metas]
            refined = [np.empty((m[0], m[1]), np.uint8) for m in metas] if refine else [None] * B
        mptr = (C.c_void_p * B)(*[m.ctypes.data for m in masks])
        rptr = (C.c_void_p * B)(*[r.ctypes.data for r in refined]) if refine else None
        ev = C.c_void_p(ready_event.cuda_event) if ready_event is not None else None
        L.check(self._lib.ctd_tail_run(self._h, B, Hn, Wn, blks.data_ptr(), blks.shape[1], blks.shape[2],
                                       mask_u8.data_ptr(), lines_map.data_ptr(), prob_stride, bitmap.data_ptr(), tab,
                                       C.byref(prm), mptr, rptr, ev), "ctd_tail_run")
        rec = None
        if records is not None:
            cb, cl = int(records[0]), int(records[1])
            rec = np.empty((B, 4 + 12 * cb + 8 * cl), np.float64)
            L.check(self._lib.ctd_tail_pack_records(self._h, cb, cl, rec.ctypes.data), "ctd_tail_pack_records")
        out = []
        lists = self._block_lists(B, lazy) if not want_extras else None
        for b in range(B):
            if want_extras:
                blk_list, extras = self._blocks(b, True)
                r = (masks[b], refined[b], blk_list, extras)
            else:
                r = (masks[b], refined[b], lists[b])
            if rec is not None:
                r = PageResult(r)
                r.record = rec[b]
            out.append(r)
        return out

    def timings(self) -> dict:
        """Host wall clock (ms) of the stages of the last `run`."""
        ms = (C.c_double * 16)()
        L.check(self._lib.ctd_tail_timings(self._h, ms), "ctd_tail_timings")
        keys = ("enqueue1", "wait1", "db_tables+geometry", "yolo+group_output", "refine_wait_hist", "refine_wait_xor",
                "refine_host+enqueue", "undetected", "final_wait+copies", "total", "db_table_wait", "enq1_nms+buffers",
                "enq1_labelling+tables", "enq1_mask_copies", "final_wait_only", "refine_wait_merge")
        return {k: round(v, 3) for k, v in zip(keys, ms)}

    def refine_paths(self) -> dict:
        """Which path the refine windows of the last `run` / `refine` took: `lds` = merged by the window-local kernel (one
        block per window on bit planes in LDS, csrc/kernels_twlds.hip), `canvas` = through the packed canvases (too large for
        the LDS, or re-done after an overflow), `overflow` = run-table overflows of the window-local kernel."""
        c = (C.c_int32 * 3)()
        L.check(self._lib.ctd_tail_refine_paths(self._h, c), "ctd_tail_refine_paths")
        return {"lds": int(c[0]), "canvas": int(c[1]), "overflow": int(c[2])}

    # -- SegDetectorRepresenter alone -----------------------------------------------------------------
    def db_boxes(self, prob: torch.Tensor, bitmap: torch.Tensor, max_candidates=1000, unclip_ratio=1.5):
        B, Hn, Wn = bitmap.shape
        prob = prob.float().contiguous()
        bitmap = bitmap.contiguous()
        torch.cuda.current_stream(self.device).synchronize()
        L.check(self._lib.ctd_tail_db_boxes(self._h, B, Hn, Wn, prob.data_ptr(), prob.stride(0), bitmap.data_ptr(),
                                            int(max_candidates), float(unclip_ratio)), "ctd_tail_db_boxes")
        boxes, scores = [], []
        for b in range(B):
            nx = C.c_int32()
            L.check(self._lib.ctd_tail_page_counts(self._h, b, None, None, None, C.byref(nx), None), "ctd_tail_page_counts")
            bx = np.empty((nx.value, 4, 2), np.int16)
            sc = np.empty((nx.value,), np.float32)
            L.check(self._lib.ctd_tail_page_fetch(self._h, b, None, None, None, bx.ctypes.data, sc.ctypes.data, None, None,
                                                  None), "ctd_tail_page_fetch")
            boxes.append(bx)
            scores.append(sc)
        return boxes, scores

    # -- refine_mask (+ refine_undetected_mask) alone ------------------------------------------------
    def refine(self, pages_gpu: Sequence[torch.Tensor], masks: Sequence[np.ndarray], boxes: Sequence[Sequence],
               refine_mode: int = 0, keep_undetected_mask: bool = False):
        """masks[b]: host uint8 (im_h, im_w); boxes[b]: the blocks' xyxy.  Returns (refined, masks_after)."""
        n = len(masks)
        metas = [(m.shape[0], m.shape[1], 0, 0) for m in masks]
        tab = self._page_table(pages_gpu, metas)
        masks = [np.ascontiguousarray(m, np.uint8) for m in masks]
        xy = np.ascontiguousarray(np.array([list(map(int, bb)) for pb in boxes for bb in pb], np.int32).reshape(-1, 4))
        cnt = np.array([len(pb) for pb in boxes], np.int32)
        refined = [_pinned_u8(*m.shape) for m in masks]
        after = [_pinned_u8(*m.shape) for m in masks] if keep_undetected_mask else None
        inp = (C.c_void_p * n)(*[m.ctypes.data for m in masks])
        rptr = (C.c_void_p * n)(*[r.ctypes.data for r in refined])
        aptr = (C.c_void_p * n)(*[a.ctypes.data for a in after]) if after is not None else None
        torch.cuda.current_stream(self.device).synchronize()        # the pages were uploaded on torch's stream
        L.check(self._lib.ctd_tail_refine(self._h, n, tab, inp, xy.ctypes.data if len(xy) else None, cnt.ctypes.data,
                                          int(refine_mode), int(bool(keep_undetected_mask)), aptr, rptr), "ctd_tail_refine")
        return refined, (after if after is not None else masks)


_tls = threading.local()
_free = {}                      # device index -> native tails whose threads have ended, ready for the next thread
_free_lock = threading.Lock()


class _Lease:
    """A thread's hold on a `Tail`.  When the thread ends (its thread-local state is dropped) the tail goes back to the
    module's free list instead of being destroyed, and the next worker thread takes it over: the process never owns more
    tail STREAMS than it has had concurrent tail threads.  That matters on this stack: a process that has had more compute
    streams of one priority class alive than the runtime has hardware queues for them makes later streams share a queue,
    and it does not recover when the old streams are destroyed -- measured (scripts/gpu_inprocess.py) with the tails'
    streams at the highest priority (rounds 2-3): a pipeline started while an earlier pipeline's three idle tails still
    existed ran at 1970-2050 pages/s instead of 2510, and so did every pipeline after it; with the default priority
    (round 4) the cliff comes at the third kept-alive pool instead of the second.  Pipelines that REUSE their tails run at
    2450-2550 however often they are rebuilt.  (Round 3 had seen this as "a second pipeline in one process is 12-18 %
    slower" and moved bench.py's sub-runs into child processes.)"""

    def __init__(self, tail: Tail):
        self.tail = tail

    def __del__(self):
        try:
            with _free_lock:
                _free.setdefault(self.tail.device.index or 0, []).append(self.tail)
        except Exception:
            pass


def bind_thread(device) -> None:
    """Makes `device` the calling thread's current device.  A new host thread starts on device 0: on a multi-GPU node a
    worker or loader thread of rank r would otherwise create a context on GPU 0 with its first device-less torch call
    (pinned allocations, events) -- 7 foreign contexts on rank 0's GPU in an 8-rank launch."""
    device = torch.device(device)
    if device.index is not None and torch.cuda.current_device() != device.index:
        torch.cuda.set_device(device)


def thread_tail(device) -> Tail:
    """One `Tail` per (host thread, device): its stream and buffers are not shared between live threads; tails of threads
    that have ended are reused (`_Lease`)."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    bind_thread(device)
    pool = getattr(_tls, "pool", None)
    if pool is None:
        pool = _tls.pool = {}
    if device.index not in pool:
        with _free_lock:
            spare = _free.get(device.index)
            t = spare.pop() if spare else None
        if t is None:
            t = Tail(device)
        elif _host_threads is not None:
            L.check(t._lib.ctd_tail_set_threads(t._h, _host_threads), "ctd_tail_set_threads")
        pool[device.index] = _Lease(t)
    return pool[device.index].tail


def release_thread_tail() -> None:
    """Hands the calling thread's tails back to the free list now (they return by themselves when the thread ends): a
    long-lived thread that ran a tail once -- the main thread after a `detect_batch` -- otherwise keeps a stream of its own."""
    pool = getattr(_tls, "pool", None)
    if pool:
        pool.clear()


def drain_free_tails(device=None) -> int:
    """Destroys the native tails waiting in the free list (those of `device`, or all): each holds a HIP stream, ~250 MB of
    device tables at 32 pages per batch and its pinned buffers.  Tails leased to live threads are not touched.  A process
    that is done with detection for good (or about to hand the GPU to something else) calls this after
    `TextDetector.close()`; a process that will build another pipeline should NOT -- reused tails are what keeps the
    stream count of the process inside the hardware queues (`_Lease`).  Returns how many were destroyed."""
    with _free_lock:
        if device is None:
            gone = [t for ts in _free.values() for t in ts]
            _free.clear()
        else:
            idx = torch.device(device).index or 0
            gone = _free.pop(idx, [])
    n = len(gone)
    for t in gone:
        t.__del__()
    return n


def live_tails() -> int:
    """Native tails this process owns right now (in use by a thread or waiting in the free list) -- for tests."""
    import gc
    return sum(1 for o in gc.get_objects() if isinstance(o, Tail))
