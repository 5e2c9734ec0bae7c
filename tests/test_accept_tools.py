"""CPU: the acceptance tooling itself (oracle/accept.py band_report / explain_geometry -- what the -m gpu band test and
bench.py's `parity.fp16_band` rest on) on constructed cases, and bench.py's host-side helpers."""
import importlib
import os
import sys

import numpy as np

from conftest import ROOT
from oracle import accept


class _Blk:
    def __init__(self, xyxy, lines):
        self.xyxy, self.lines = xyxy, lines


def test_band_report_counts_flips_and_their_distance_to_the_threshold():
    rng = np.random.RandomState(0)
    prob = rng.uniform(0.0, 1.0, (64, 64)).astype(np.float32)
    mask = rng.uniform(0.0, 1.0, (64, 64)).astype(np.float32)
    bitmap = (prob > 0.3).astype(np.uint8)
    mask_u8 = (mask * 255).astype(np.uint8)
    rep = accept.band_report(prob, mask, bitmap, mask_u8, 1e-3)
    assert rep["bitmap_flips"] == 0 and rep["mask127_flips"] == 0
    # flip one pixel close to the threshold and one far from it
    prob[3, 4], prob[10, 10] = 0.3004, 0.9
    bitmap = (prob > 0.3).astype(np.uint8)
    bitmap[3, 4] ^= 1
    bitmap[10, 10] ^= 1
    rep = accept.band_report(prob, mask, bitmap, mask_u8, 1e-3)
    assert rep["bitmap_flips"] == 2 and rep["bitmap_flips_out_of_band"] == 1
    assert abs(rep["bitmap_flips_max_dist_to_thresh"] - 0.6) < 1e-6
    assert rep["_flips"][3, 4] and rep["_flips"][10, 10] and rep["_flips"].sum() == 2
    # the u8 mask threshold: u8 > 127 <=> mask * 255 >= 128
    mask[:] = 0.2
    mask[5, 5] = 128.2 / 255
    m8 = (mask * 255).astype(np.uint8)
    m8[5, 5] = 127                                   # the product truncated to the other side
    rep = accept.band_report(prob, mask, bitmap, m8, 4e-3)
    assert rep["mask127_flips"] == 1 and rep["mask127_flips_out_of_band"] == 0


def test_explain_geometry_attributes_every_difference_to_a_cause():
    flips = np.zeros((100, 100), bool)
    flips[20, 30] = True
    q = lambda x1, y1, x2, y2: [[x1, y1], [x2, y1], [x2, y2], [x1, y2]]          # noqa: E731
    ref = (None, None, [_Blk([10, 10, 50, 40], [q(12, 12, 48, 38)]), _Blk([60, 60, 90, 90], [q(62, 62, 88, 70)]),
                        _Blk([5, 70, 40, 95], [q(6, 72, 38, 80)])])
    # line 1 changed near the flipped pixel, block 2 moved by one pixel (same lines), block 3 identical
    got = (None, None, [_Blk([10, 10, 50, 40], [q(12, 12, 47, 36)]), _Blk([60, 60, 91, 90], [q(62, 62, 88, 70)]),
                        _Blk([5, 70, 40, 95], [q(6, 72, 38, 80)])])
    g = accept.explain_geometry(got, ref, flips)
    assert g["lines_differing"] == 2 and g["lines_by_cause"]["flip"] == 2 and g["lines_unexplained"] == 0
    assert g["blocks_differing"] == 4 and g["blocks_by_cause"]["lines"] == 2 and g["blocks_by_cause"]["near"] == 2
    assert g["blocks_unexplained"] == 0
    # a line far from any flip, without counterpart, is UNEXPLAINED ...
    got2 = (None, None, ref[2] + [_Blk([70, 5, 95, 30], [q(72, 7, 93, 20)])])
    g = accept.explain_geometry(got2, ref, flips)
    assert g["lines_unexplained"] == 1 and g["blocks_unexplained"] == 0          # the block is explained by its line
    # ... unless a detection differs there, a DB score sits on the 0.6 gate there, or the contour cut moved there
    dets = np.array([[70.0, 5, 95, 30, 0.5, 0]])
    assert accept.explain_geometry(got2, ref, flips, dets=dets, ref_dets=np.zeros((0, 6)))["lines_by_cause"]["det"] == 1
    sbb = np.array([q(72, 7, 93, 20)])
    assert accept.explain_geometry(got2, ref, flips, score_band_boxes=sbb)["lines_by_cause"]["score"] == 1
    ours = np.concatenate([np.array([q(72, 7, 93, 20)]), np.zeros((2, 4, 2), int)])
    theirs = np.zeros((3, 4, 2), int)
    assert accept.explain_geometry(got2, ref, flips, candidates=(ours, theirs, 3))["lines_by_cause"]["cut"] == 1
    assert accept.explain_geometry(got2, ref, flips, candidates=(ours, theirs, 1000))["lines_unexplained"] == 1   # lists not full
    # page -> network coordinates
    g = accept.explain_geometry(got, ref, np.roll(np.roll(flips, 20, 0), 30, 1), ratio_xy=(2.0, 2.0))
    assert g["lines_by_cause"]["flip"] == 2


def test_bench_thread_budget_and_torchrun_command(monkeypatch):
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    DET = importlib.import_module("comic-text-detector_amd.detector")       # the budget is the product's (detect_stream(workers=0))
    monkeypatch.setattr(DET, "usable_cpus", lambda: 256)
    monkeypatch.setattr(DET, "cgroup_cpu_quota", lambda: 0.0)
    assert DET.thread_budget() == bench.thread_budget(1)
    assert bench.thread_budget(1) == {"usable_cpus": 256, "per_rank": 256, "tail_workers": 4, "native_threads_per_worker": 8}
    b8 = bench.thread_budget(8)
    assert b8["per_rank"] == 32 and b8["tail_workers"] == 4 and b8["native_threads_per_worker"] == 8      # 8 pages per work item: 8, not 7
    assert bench.thread_budget(16)["tail_workers"] == 4 and bench.thread_budget(20)["tail_workers"] == 3      # 16 / 12 per rank
    monkeypatch.setattr(DET, "usable_cpus", lambda: 16)
    small = bench.thread_budget(8)
    assert small["tail_workers"] == 2 and small["native_threads_per_worker"] == 2                         # per_rank = max(4, 16 // 8)
    # `python bench.py --gpus 2` outside torchrun: N ranks of this script on 127.0.0.1, one-device rehearsal without 2 devices
    calls = {}
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])

    class A:
        gpus = 2
    assert bench.relaunch_under_torchrun(A()) == 0
    cmd = calls["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and os.path.basename(cmd[-5]) == "bench.py"
    assert calls["env"]["CTD_BENCH_ONE_DEVICE"] == "1" and calls["env"]["CTD_DIST_BACKEND"] == "gloo"


def test_env_tuning_is_applied_by_the_host_side(monkeypatch):
    """The library reads no environment; `_lib._apply_env_tuning` turns CTD_TUNING / the older knob names into
    ctd_tuning_set calls (and rejects unknown keys loudly)."""
    import pytest
    L = importlib.import_module("comic-text-detector_amd._lib")
    seen = []

    class Fake:
        @staticmethod
        def ctd_tuning_set(k, v):
            seen.append((k.decode(), v))
            return 0 if k.decode() != "bogus" else -1
    monkeypatch.setenv("CTD_TUNING", "fuse=5, halo_pair=0")
    monkeypatch.setenv("CTD_FUSE", "3")
    monkeypatch.setenv("CTD_NO_REUSE", "yes")
    L._apply_env_tuning(Fake)
    assert ("fuse", 3) in seen and ("no_reuse", 1) in seen and ("fuse", 5) in seen and ("halo_pair", 0) in seen
    assert seen.index(("fuse", 3)) < seen.index(("fuse", 5))           # CTD_TUNING wins over the legacy names
    monkeypatch.setenv("CTD_TUNING", "bogus=1")
    with pytest.raises(L.CtdError):
        L._apply_env_tuning(Fake)
