"""N>1 path on CPU: world_size-2 gloo run of the page sharding + the one
collective of the data path (all-gather of the fixed-capacity page records)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_page(i):
    g = torch.Generator().manual_seed(1000 + i)
    n = int(torch.randint(0, 301, (1,), generator=g))
    dets = torch.zeros(300, 6)
    dets[:n] = torch.rand(n, 6, generator=g) * 1000
    return dets, n


def _worker(rank, world, port, n_total, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D = pkg().dist
    r, lr, w = D.init("gloo")
    lo, hi = D.shard_range(n_total, r, w)
    dets = torch.stack([_fake_page(i)[0] for i in range(lo, hi)]) if hi > lo else torch.zeros(0, 300, 6)
    counts = torch.tensor([_fake_page(i)[1] for i in range(lo, hi)], dtype=torch.int32)
    rec = D.pack_records(dets, counts)
    allrec = D.gather_records(rec, n_total, r, w)
    d2, c2, _, _ = D.unpack_records(allrec)
    ok = True
    for i in range(n_total):
        ref, n = _fake_page(i)
        ok &= int(c2[i]) == n and torch.equal(d2[i], ref)
    q.put((rank, ok, tuple(allrec.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_gloo_world2_shard_and_gather(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, f"rank {rank}: gathered records differ from the single-process result"
        assert shape[0] == n_total


def test_shard_range_partitions_everything():
    D = pkg().dist
    for n in (0, 1, 5, 32, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- the whole N>1 data path on CPU: shard -> per-page grouping (native host code) -> gather of the block records


def _page_blocks(i, crowded=False):
    """The grouped block list of global page i (native `ctd_group_output`, host only).  crowded: the LAST page gets more
    blocks than the compact record holds (copies of its blocks), which must send every rank to the full-capacity gather."""
    from test_group_native import random_page
    p = pkg()
    blks, lines, im_w, im_h, mask = random_page(500 + i)
    out = p.textblock.group_output(blks, lines, im_w, im_h, mask)
    if crowded and i == crowded - 1 and out:
        out = (out * (p.dist.CAP_BLK // len(out) + 2))[: p.dist.CAP_BLK + 7]
    return out


def _worker_blocks(rank, world, port, n_total, q, crowded=0):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D = pkg().dist
    r, lr, w = D.init("gloo")
    lo, hi = D.shard_range(n_total, r, w)
    results = [(None, None, _page_blocks(i, crowded)) for i in range(lo, hi)]          # this rank's pages
    allrec = D.gather_results(results, n_total, r, w)
    got = D.unpack_results(allrec)
    ok_caps = int(allrec[0, 2]) == (D.MAX_BLK if crowded else D.CAP_BLK)         # compact unless a page did not fit
    ok = len(got) == n_total and ok_caps
    for i in range(n_total):
        ref = _page_blocks(i, crowded)
        ok &= len(got[i]) == len(ref)
        for a, b in zip(got[i], ref):
            ok &= a["xyxy"] == [int(v) for v in b.xyxy] and a["lines"] == [[[int(x) for x in pt] for pt in ln] for ln in b.lines]
            ok &= (a["language"], a["vertical"], a["angle"]) == (b.language, bool(b.vertical), int(b.angle))
            ok &= a["font_size"] == float(b.font_size) and a["norm"] == float(b.norm) and a["vec"] == [float(v) for v in b.vec]
    q.put((rank, bool(ok), tuple(allrec.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total,crowded", [(6, 0), (5, 0), (5, 5)])
def test_gloo_world2_grouped_blocks_gathered_equal_single_process(n_total, crowded):
    """Every rank ends up with the final block list of EVERY page (SURVEY 8(e) record: blocks + their
    lines), identical to what one process computes -- also with an uneven split, and also when one page (on the last
    rank) holds more blocks than the compact record: all ranks then repeat the gather at the worst-case capacities."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_blocks, args=(r, 2, port, n_total, q, crowded)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok, f"rank {rank}: gathered block lists differ from the single-process result"
        assert shape[0] == n_total
