"""Multi-GPU: pages are independent units, so the batch is sharded
contiguously across ranks (one process per GPU) and the ONLY collective on the
data path is one all-gather of the fixed-capacity per-page result records
(SURVEY 8(e)).  `backend="nccl"` is RCCL over xGMI on ROCm; the same code runs
on `gloo` for the CPU tests.

The reference has no distributed code at all; this module is new.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

MAX_DET = 300      # reference utils/yolov5_utils.py:125 max_det
MAX_LINES = 1000   # reference utils/db_utils.py:33 max_candidates


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """Joins the process group of the torchrun environment.  A single process has no group and no collective on its data
    path -- unless `force=True`, which gives it a world-size-1 group of `backend` so that the N > 1 code (RCCL communicator
    on this rank's device, the record gather on the communication stream) runs on the one GPU that is there.

    CALL IT AFTER the detector and its pipeline exist (`TextDetector(...)`, the first `detect_stream` / `warm_tails`): RCCL
    creates streams of its own when the communicator starts, and HIP streams beyond the runtime's hardware queues share a
    queue with an earlier one -- with the group initialised FIRST the end-to-end step measured 20 % slower for the whole
    life of the process (2480-2590 against 3134 pages/s, profiles/r06_rccl_init_order.txt), with or without a gather."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # CTD_DIST_BACKEND=gloo: lets a 1-GPU box rehearse the N > 1 code path (all ranks on one device)
            backend = os.environ.get("CTD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split: rank r owns pages [lo, hi).  Remainder pages go to the
    lowest ranks so sizes differ by at most one."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_records(dets: torch.Tensor, counts: torch.Tensor, lines: Optional[torch.Tensor] = None,
                 line_counts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One fixed-size f32 record per page:
       [n_blk, n_line, blocks MAX_DET x 6 (xyxy, conf, cls), lines MAX_LINES x 9 (4 pts + score)]."""
    B = dets.shape[0]
    rec = torch.zeros((B, 2 + MAX_DET * 6 + MAX_LINES * 9), dtype=torch.float32, device=dets.device)
    rec[:, 0] = counts.to(torch.float32)
    rec[:, 2:2 + MAX_DET * 6] = dets.reshape(B, -1)
    if lines is not None:
        rec[:, 1] = line_counts.to(torch.float32)
        rec[:, 2 + MAX_DET * 6:] = lines.reshape(B, -1)
    return rec


def unpack_records(rec: torch.Tensor):
    B = rec.shape[0]
    counts = rec[:, 0].to(torch.int32)
    line_counts = rec[:, 1].to(torch.int32)
    dets = rec[:, 2:2 + MAX_DET * 6].reshape(B, MAX_DET, 6)
    lines = rec[:, 2 + MAX_DET * 6:].reshape(B, MAX_LINES, 9)
    return dets, counts, lines, line_counts


# ---- the grouped result of a page (the reference's final `blk_list`, inference.py:173) -----------------
MAX_BLK = MAX_DET + MAX_LINES      # group_output yields at most one block per yolo box + one per unassigned line
BLK_F = 12                          # xyxy(4), language, vertical, angle, font_size, n_lines, norm, vec(2)


HDR = 4                             # n_blk, n_lines (true counts), cap_blk, cap_line
CAP_BLK, CAP_LINE = 128, 512        # the compact record: 45 KB per page instead of 208 KB at the worst-case capacities


def pack_results(results, device=None, cap_blk: int = MAX_BLK, cap_line: int = MAX_BLK) -> torch.Tensor:
    """`detect_batch` results [(mask, mask_refined, blk_list), ...] -> one fixed-size f64 record per page:
    [n_blk, n_lines, cap_blk, cap_line, blocks cap_blk x 12, lines cap_line x 8 (4 points, in block order)].
    float64 holds every field exactly (coordinates and angles are integers, font sizes / norms are doubles).
    The counts are the TRUE counts: a page with more blocks / lines than the capacities is stored truncated and is
    recognisable as such (`gather_results` then repeats the gather at the worst-case capacities).  Masks stay on
    the rank that produced them."""
    import numpy as np
    from .textblock import LANGCLS2IDX
    width = HDR + cap_blk * BLK_F + cap_line * 8
    # records the native tail already built (`Tail.run(records=...)` -> `PageResult.record`): no Python loop
    pre = [getattr(r, "record", None) for r in results]
    if results and all(p is not None and p.shape == (width,) and p[2] == cap_blk and p[3] == cap_line for p in pre):
        out = torch.from_numpy(np.stack(pre))
        return out.to(device) if device is not None else out
    rec = np.zeros((len(results), width), np.float64)
    rec[:, 2], rec[:, 3] = cap_blk, cap_line
    for p, r in enumerate(results):
        blks = r[2]
        if not blks:
            continue
        head = np.array([[*b.xyxy, LANGCLS2IDX[b.language], bool(b.vertical), b.angle, b.font_size, len(b.lines), b.norm,
                          b.vec[0], b.vec[1]] for b in blks], np.float64)
        lines = [np.asarray(b.lines, np.float64).reshape(-1, 8) for b in blks if len(b.lines)]
        lines = np.concatenate(lines) if lines else np.zeros((0, 8))
        rec[p, 0], rec[p, 1] = len(blks), len(lines)
        head, lines = head[:cap_blk], lines[:cap_line]
        rec[p, HDR: HDR + head.size] = head.ravel()
        lb = HDR + cap_blk * BLK_F
        rec[p, lb: lb + lines.size] = lines.ravel()
    out = torch.from_numpy(rec)
    return out.to(device) if device is not None else out


def unpack_results(rec: torch.Tensor):
    """Inverse of pack_results: per page a list of dicts (xyxy, language, vertical, angle, font_size, norm, vec, lines)."""
    from .textblock import LANG_LIST
    rec = rec.cpu()
    out = []
    for p in range(rec.shape[0]):
        nb, cap_blk, cap_line = int(rec[p, 0]), int(rec[p, 2]), int(rec[p, 3])
        if nb > cap_blk or int(rec[p, 1]) > cap_line:
            raise ValueError("truncated page record: gather at the worst-case capacities (gather_results does)")
        blks, nl = [], 0
        for i in range(nb):
            f = rec[p, HDR + i * BLK_F: HDR + (i + 1) * BLK_F].tolist()
            n = int(f[8])
            lb = HDR + cap_blk * BLK_F + nl * 8
            lines = rec[p, lb: lb + 8 * n].reshape(n, 4, 2).to(torch.int64).tolist()
            nl += n
            blks.append(dict(xyxy=[int(v) for v in f[:4]], language=LANG_LIST[int(f[4])], vertical=bool(f[5]),
                             angle=int(f[6]), font_size=f[7], norm=f[9], vec=[f[10], f[11]], lines=lines))
        out.append(blks)
    return out


class GatherHandle:
    """A record gather in flight (`gather_results_async`).  Nothing here touches the GPU from the host until `result()`:
    the true block / line counts of every page ride back to page-locked host memory behind the collective, and an event
    marks the end -- so a pipelined caller never waits on a stream that is busy with the next forwards."""

    def __init__(self, out, counts_host, event, redo, shape=None):
        self.out, self._counts, self._event, self._redo, self._shape = out, counts_host, event, redo, shape

    def done(self) -> bool:
        return self._event is None or self._event.query()

    def result(self) -> torch.Tensor:
        """The gathered records (n_total, R) in global page order.  Waits for the collective; when some page of some rank
        did not fit the compact capacities (every rank sees that in the same counts) the gather is repeated at the
        worst-case capacities, synchronously -- rare."""
        if self._event is not None:
            self._event.synchronize()
            self._event = None
        if self._redo is not None:
            c = self._counts
            over = c.shape[0] > 0 and bool(((c[:, 0] > CAP_BLK) | (c[:, 1] > CAP_LINE)).any())
            redo, self._redo = self._redo, None
            if over:
                self.out, self._shape = redo(), None
        if self._shape is not None:                          # the ring slot's padded (world * per) rows -> global page order
            n_total, world = self._shape
            self.out, self._shape = _unpad(self.out, n_total, world), None
        return self.out


def _unpad(out: torch.Tensor, n_total: int, world: int) -> torch.Tensor:
    per = -(-n_total // world)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts, 0)


class _Ring:
    """Page-locked staging + device buffers of the record gather, allocated ONCE per (shard, capacities, device) and reused
    round robin: `Tensor.pin_memory()` per step cost 5 ms of host time in the pipelined step (a fresh page-locked
    allocation each time -- the caching allocator's blocks were still held by the uploads in flight) and stalled the
    launching thread for a whole pipeline depth (bench.py --force-dist nccl: 32 ms per step instead of 10)."""
    SLOTS = 12

    def __init__(self, nloc, width, n_total, world, device):
        per = -(-n_total // world)
        self.slots = []
        for _ in range(self.SLOTS):
            self.slots.append(dict(host=torch.zeros((nloc, width), dtype=torch.float64).pin_memory(),
                                   pad=torch.zeros((per, width), dtype=torch.float64, device=device),
                                   out=torch.empty((world * per, width), dtype=torch.float64, device=device),
                                   counts=torch.zeros((world * per, 2), dtype=torch.float64).pin_memory(), ev=None))
            self.slots[-1]["host_np"] = self.slots[-1]["host"].numpy()
        self.k = 0

    def take(self):
        s = self.slots[self.k % self.SLOTS]
        self.k += 1
        if s["ev"] is not None:
            s["ev"].synchronize()                            # twelve gathers ago: long done
        return s


_rings = {}


def gather_results_async(results, n_total: int, rank: int, world: int, device=None, pin: bool = False,
                         force: bool = False) -> GatherHandle:
    """The data path's collective, enqueued and NOT waited for: all-gather of this rank's page records at the COMPACT
    capacities (CAP_BLK blocks, CAP_LINE lines: 4.6x fewer bytes than the worst case, host packing included) on the
    current stream.  Over gloo the records never leave the host (no device work at all)."""
    gloo = dist.is_initialized() and dist.get_backend() == "gloo"
    dev = None if gloo else device
    grouped = dist.is_initialized() and (world > 1 or force)

    def one(cb, cl):                                         # the plain path: host records, or the (rare) worst-case re-gather
        rec = pack_results(results, None, cb, cl)
        if dev is not None:
            rec = (rec.pin_memory() if pin else rec).to(dev, non_blocking=pin)
        return gather_records(rec, n_total, rank, world, force)

    redo = lambda: one(MAX_BLK, MAX_BLK)                    # noqa: E731
    if dev is None or not grouped or not results:
        out = one(CAP_BLK, CAP_LINE)
        return GatherHandle(out, out[:, :2].cpu() if out.is_cuda else out[:, :2], None, redo)
    rec = pack_results(results, None, CAP_BLK, CAP_LINE)     # host, (nloc, width)
    key = (rec.shape[0], rec.shape[1], n_total, world, str(dev))
    ring = _rings.get(key)
    if ring is None:
        ring = _rings[key] = _Ring(rec.shape[0], rec.shape[1], n_total, world, dev)
    s = ring.take()
    # numpy, not `Tensor.copy_`: torch's CPU copy runs on the intra-op thread pool, which the tail workers' native threads
    # keep busy -- 9.7 ms for these 1.4 MB inside the pipelined step (and `pin_memory()` 5 ms), 0.1 ms this way
    import numpy as np
    np.copyto(s["host_np"], rec.numpy())
    s["pad"][: rec.shape[0]].copy_(s["host"], non_blocking=True)
    dist.all_gather_into_tensor(s["out"], s["pad"])
    s["counts"].copy_(s["out"][:, :2], non_blocking=True)    # behind the collective, on the same stream
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(s["out"].device))
    s["ev"] = ev
    return GatherHandle(s["out"], s["counts"], ev, redo, (n_total, world))


def gather_results(results, n_total: int, rank: int, world: int, device=None, pin: bool = False,
                   force: bool = False) -> torch.Tensor:
    """`gather_results_async(...).result()`: every rank ends up with every page's records; all ranks agree without further
    communication on whether some page did not fit the compact record -- only then the gather is repeated at the
    worst-case capacities (MAX_BLK)."""
    return gather_results_async(results, n_total, rank, world, device, pin, force).result()


def gather_records(rec: torch.Tensor, n_total: int, rank: int, world: int, force: bool = False) -> torch.Tensor:
    """All-gather the per-page records; returns (n_total, R) in global page order on
    every rank.  Shards may differ by one page, so each rank pads to the largest shard.
    A single process returns its records as they are; `force=True` sends them through the collective of its world-size-1
    group anyway (`init(force=True)`)."""
    if world == 1 and not (force and dist.is_initialized()):
        return rec
    per = -(-n_total // world)
    home = rec.device
    if dist.get_backend() == "gloo" and rec.is_cuda:        # gloo gathers host tensors (callers with host records never get here)
        rec = rec.cpu()
    pad = torch.zeros((per, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    pad[: rec.shape[0]] = rec
    out = torch.empty((world * per, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, pad)
    out = out.to(home)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r * per: r * per + (hi - lo)])
    return torch.cat(parts, 0)
