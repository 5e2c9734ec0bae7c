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


# ---- -m gpu: the record gather's inputs built natively, and the N > 1 bench path rehearsed on one device ------------


@pytest.mark.gpu
def test_native_page_records_equal_the_python_packing():
    """`Tail.run(records=...)` -> `ctd_tail_pack_records`: the per-page gather record the native tail builds is the array
    `dist.pack_results` builds from the Python `TextBlock` objects, field for field, at the compact AND at tiny
    (truncating) capacities -- so the N > 1 step needs no Python loop over blocks."""
    import numpy as np
    p = pkg()
    D = p.dist
    ck = p.synth.make_blob_checkpoint(0)
    det = p.detector.TextDetector(ck, input_size=512, device="cuda", precision="fp16")
    pages = [p.synth.text_like_page((512, 512), s, n_blocks=6) for s in (3, 4, 5)]
    for caps in ((D.CAP_BLK, D.CAP_LINE), (4, 6)):
        job = det._forward(pages)
        res = det._tail(job, 0, False, records=caps)
        assert all(r.record is not None and len(r) == 3 for r in res)
        assert sum(len(r[2]) for r in res) > 10                                  # the pages have blocks to pack
        native = np.stack([r.record for r in res])
        plain = [(r[0], r[1], r[2]) for r in res]                                # no .record: the Python loop packs
        ref = D.pack_results(plain, None, *caps).numpy()
        np.testing.assert_array_equal(native, ref)
        assert D.pack_results(res, None, *caps).numpy().tobytes() == ref.tobytes()     # the fast path returns the same
    got = D.unpack_results(D.pack_results(res, None, D.CAP_BLK, D.CAP_LINE))
    assert [len(g) for g in got] == [len(r[2]) for r in res]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_device():
    """`python bench.py --gpus 2` with no torchrun environment re-executes itself under torch.distributed.run; on a box
    with one GPU both ranks share it over gloo (flagged in the line as a rehearsal).  Covers rank start-up, sharding,
    the native page records, the all-gather inside the pipelined step, the barrier / max-over-ranks timing."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if torch.cuda.device_count() >= 2:
        env["CTD_BENCH_ONE_DEVICE"] = "1"
        env["CTD_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--spinup", "2", "--batch", "4", "--size", "512", "--batches", "2", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8
    assert "ranks=2" in out["config"]["parallelism"] and out["config"]["one_device_rehearsal"] is True
    assert out["value"] > 0 and out["config"]["blocks_per_page"] > 0


@pytest.mark.gpu
def test_bench_four_ranks_tail_only_on_one_device():
    """Four ranks (four interpreter pipelines, 4 x tail workers x native threads, four pinned arenas, the record gather
    inside the step) on one device, forward left out (`--tail-only`): the host-side rehearsal of the N > 1 launch."""
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["CTD_BENCH_ONE_DEVICE"] = "1"
    env["CTD_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--tail-only", "--steps", "3", "--warmup",
                        "1", "--spinup", "2", "--batch", "4", "--size", "512", "--batches", "1", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["config"]["global_batch"] == 16 and out["config"]["tail_only"] is True
    assert out["config"]["host_threads"]["per_rank"] * 4 <= max(16, out["config"]["host_threads"]["usable_cpus"])
    aff = out["config"]["cpu_affinity"]                  # every rank bound to its own CPUs of the GPU's NUMA node (affinity.py)
    assert aff["pinned"] is True and aff["n_cpus"] >= 1 and aff["n_cpus"] * 4 <= max(4, out["config"]["host_threads"]["usable_cpus"])
    assert out["value"] > 0 and out["config"]["blocks_per_page"] > 0 and out["config"]["host_cpu_cores_used"] > 0


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two devices in one process")
def test_one_process_drives_detectors_on_two_devices():
    """`TextDetector(device='cuda:1')` next to one on `cuda:0` in ONE process -- the native tails' stage-1 chain
    (mutex, last event, ring of 16 events) is per device, so more than 16 batches on each device, interleaved, must neither
    fail on an event / stream device mismatch nor change a result."""
    import numpy as np
    p = pkg()
    ck = p.synth.make_blob_checkpoint(0, sparse_det=True)
    pages = [p.synth.text_like_page((512, 512), s, n_blocks=6) for s in (3, 4)]
    dets = [p.detector.TextDetector(ck, input_size=512, device=f"cuda:{d}", precision="fp16") for d in (0, 1)]
    ref = None
    for it in range(20):                                                        # the ring wraps after 16
        for det in dets:
            res = det.detect_batch(pages)
            sig = [(np.asarray(r[1]).tobytes(), [tuple(int(v) for v in b.xyxy) for b in r[2]]) for r in res]
            if ref is None:
                ref = sig
            assert sig == ref, f"iteration {it} on {det.device}"
    for det in dets:
        det.close()


def _worker_async(rank, world, port, q):
    """Six pipelined steps of `gather_results_async`, handles resolved at a fixed lag of two steps (bench.Pipeline's rule);
    step 3 carries a page that does not fit the compact record on rank 1 only -- every rank must then issue the re-gather at
    the same place of its sequence of collectives, or the group hangs."""
    import sys
    from collections import deque
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D = pkg().dist
    r, lr, w = D.init("gloo")
    n_total, lag = 4, 2
    lo, hi = D.shard_range(n_total, r, w)
    handles, caps, ok = deque(), [], True
    for step in range(6):
        crowded = n_total if step == 3 else 0                       # the last page (rank 1's) is crowded in step 3
        results = [(None, None, _page_blocks(10 * step + i, crowded and (crowded if i == n_total - 1 else 0))) for i in range(lo, hi)]
        if crowded and r == w - 1:
            blks = results[-1][2]
            results[-1] = (None, None, (blks * (D.CAP_BLK // max(len(blks), 1) + 2))[: D.CAP_BLK + 5])
        handles.append((step, D.gather_results_async(results, n_total, r, w)))
        while len(handles) > lag:
            s, h = handles.popleft()
            out = h.result()
            caps.append((s, int(out[0, 2])))
    while handles:
        s, h = handles.popleft()
        caps.append((s, int(h.result()[0, 2])))
    want = [(s, D.MAX_BLK if s == 3 else D.CAP_BLK) for s in range(6)]
    q.put((rank, caps == want, caps))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_async_gathers_resolved_at_a_fixed_lag():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, caps in res:
        assert ok, f"rank {rank}: capacities per step {caps}"
