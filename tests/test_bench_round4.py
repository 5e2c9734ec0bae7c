"""CPU tests of round 4's bench / workload helpers: the fixture-density checkpoint, the pooled oracle tail of the CPU
baseline (`oracle/tail_pool.py`), the host-thread budget."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from conftest import ROOT, pkg

sys.path.insert(0, ROOT)


def test_fixture_line_density_checkpoint_changes_only_its_two_knobs():
    S = pkg().synth
    a = S.make_blob_checkpoint(0, sparse_det=True)
    b = S.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
    changed = []
    for part in ("text_seg", "text_det"):
        for k in a[part]:
            if not torch.equal(a[part][k], b[part][k]):
                changed.append(f"{part}.{k}")
    wa, wb = a["blk_det"]["weights"], b["blk_det"]["weights"]
    changed += [f"blk_det.{k}" for k in wa if not torch.equal(wa[k], wb[k])]
    assert sorted(changed) == ["blk_det.model.24.m.2.bias", "text_det.binarize.6.bias"], changed
    assert float(b["text_det"]["binarize.6.bias"] - a["text_det"]["binarize.6.bias"]) == pytest.approx(S._FIXTURE_DB_SHIFT, abs=1e-6)
    with pytest.raises(ValueError):
        S.make_blob_checkpoint(0, sparse_det=False, line_density="fixture")
    with pytest.raises(ValueError):
        S.make_blob_checkpoint(0, sparse_det=True, line_density="dense")


def test_pooled_oracle_tail_equals_the_oracle_tail(tmp_path):
    """`python -m oracle.tail_pool` (the CPU baseline's tail leg on a process pool) runs `detector_tail` per page: block and
    line counts equal a direct call's."""
    from oracle import postproc_ref as R
    S = pkg().synth
    size, n = 256, 3
    samples = [S.text_like_outputs(40 + i, size, n_blocks=4) for i in range(n)]
    pages = np.stack([s[0] for s in samples])
    blks = np.concatenate([s[1] for s in samples])
    mask = np.stack([((s[2].astype(np.float32) + 0.5) / 255)[None] for s in samples])
    lines = np.stack([np.stack([s[3], np.zeros_like(s[3])]) for s in samples])
    path = os.path.join(tmp_path, "in.npz")
    np.savez(path, pages=pages, blks=blks, mask=mask, lines=lines)
    r = subprocess.run([sys.executable, "-m", "oracle.tail_pool", path, "2", str(size)], capture_output=True, text=True, cwd=ROOT,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["pages"] == n and out["processes"] == 2 and out["wall_s"] > 0 and len(out["per_page_single_s"]) == n
    for i in range(n):
        _, _, bl = R.detector_tail(pages[i], blks[i: i + 1], mask[i: i + 1], lines[i: i + 1], input_size=(size, size),
                                   refine_mode=0, keep_undetected_mask=False)
        assert out["blocks"][i] == len(bl) and out["lines"][i] == sum(len(b.lines) for b in bl)
    assert sum(out["lines"]) > 0


def test_thread_budget_divides_the_host_between_ranks(monkeypatch):
    import importlib
    import bench
    for cpus, world, want_workers in ((256, 1, 4), (256, 8, 4), (64, 8, 3), (16, 8, 2), (8, 1, 3)):
        DET = importlib.import_module("comic-text-detector_amd.detector")     # the product's budget (detect_stream(workers=0))
        monkeypatch.setattr(DET, "usable_cpus", lambda c=cpus: c)
        monkeypatch.setattr(DET, "cgroup_cpu_quota", lambda: 0.0)
        tb = bench.thread_budget(world)
        assert tb["tail_workers"] == want_workers, (cpus, world, tb)
        if world > 1:                                     # a rank bound to its own CPUs (affinity.py) budgets on those
            pinned = bench.thread_budget(world, True, cpus * world)
            assert pinned["per_rank"] == max(4, cpus) and pinned["usable_cpus"] == cpus * world
        assert tb["tail_workers"] * tb["native_threads_per_worker"] <= max(tb["per_rank"], tb["tail_workers"])
        assert tb["native_threads_per_worker"] in (1, 2, 4, 8)


def test_thread_budget_respects_the_container_cpu_quota(monkeypatch, tmp_path):
    """The affinity mask of a GPU box shows the host's 256 CPUs while its cgroup grants 16 (`cpu.max = 1600000 100000`): the
    budget is capped by the quota's share, per rank."""
    import importlib
    import bench
    DET = importlib.import_module("comic-text-detector_amd.detector")
    monkeypatch.setattr(DET, "usable_cpus", lambda: 256)
    monkeypatch.setattr(DET, "cgroup_cpu_quota", lambda: 16.0)
    tb = bench.thread_budget(1)                           # twice the quota: the tail's threads are bursty (detector.thread_budget)
    assert tb["per_rank"] == 32 and tb["tail_workers"] == 4 and tb["native_threads_per_worker"] == 8 and tb["cgroup_cpu_quota"] == 16.0
    monkeypatch.setattr(DET, "cgroup_cpu_quota", lambda: 4.0)
    tiny = bench.thread_budget(1)
    assert tiny["per_rank"] == 8 and tiny["tail_workers"] == 3 and tiny["native_threads_per_worker"] == 2
    monkeypatch.setattr(DET, "cgroup_cpu_quota", lambda: 64.0)
    tb8 = bench.thread_budget(8)                          # 8 ranks under a 64-CPU quota: 16 each, not 32
    assert tb8["per_rank"] == 16 and tb8["tail_workers"] == 4
    pinned = bench.thread_budget(8, True, 256)            # ... also when every rank is bound to 32 of the host's CPUs
    assert pinned["per_rank"] == 16 and pinned["usable_cpus"] == 256
    monkeypatch.setattr(DET, "cgroup_cpu_quota", lambda: 0.0)
    assert "cgroup_cpu_quota" not in bench.thread_budget(1) and bench.thread_budget(1)["per_rank"] == 256
    # the parser: cgroup v2 `cpu.max`
    real_open = open
    files = {"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}
    import builtins

    def fake_open(path, *a, **k):
        if path in files:
            f = tmp_path / "cpu.max"
            f.write_text(files[path])
            return real_open(f, *a, **k)
        if str(path).startswith("/sys/fs/cgroup/"):
            raise OSError(path)
        return real_open(path, *a, **k)
    monkeypatch.undo()
    monkeypatch.setattr(builtins, "open", fake_open)
    assert DET.cgroup_cpu_quota() == 16.0
    files["/sys/fs/cgroup/cpu.max"] = "max 100000\n"
    assert DET.cgroup_cpu_quota() == 0.0


def test_bench_checkpoint_selection():
    import bench
    p = pkg()
    fx = bench.blob_checkpoint(p, types.SimpleNamespace(dense_blocks=False, line_density="fixture"))
    r3 = bench.blob_checkpoint(p, types.SimpleNamespace(dense_blocks=False, line_density="r3"))
    de = bench.blob_checkpoint(p, types.SimpleNamespace(dense_blocks=True, line_density="fixture"))
    k = "binarize.6.bias"
    assert float(fx["text_det"][k]) != float(r3["text_det"][k]) and float(r3["text_det"][k]) == float(de["text_det"][k])
    wk = "model.24.m.2.bias"
    assert not torch.equal(r3["blk_det"]["weights"][wk], de["blk_det"]["weights"][wk])
