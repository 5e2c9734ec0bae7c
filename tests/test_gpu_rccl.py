"""-m gpu: RCCL exercised on the one GPU a box has.  The N > 1 step differs from these runs only in the
rank count: `dist.init("nccl")` with `device_id`, the record gather (`dist.gather_results`: compact path AND the overflow
re-gather) on a communication stream from the page-locked record, the barrier and the MAX reduction of the timing."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CTD_DIST_BACKEND", "CTD_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_world_size_one_nccl_group_gathers_the_page_records_like_gloo():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_worker.py")], capture_output=True, text=True,
                       timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["equal"] is True and "nccl" in out and "gloo" in out
    assert out["nccl"]["gathered_shape"] == out["gloo"]["gathered_shape"]
    assert out["nccl"]["overflow_shape"][1] > out["nccl"]["gathered_shape"][1]


def test_bench_one_gpu_forced_onto_rccl():
    """`bench.py --gpus 1 --force-dist nccl`: the pipelined end-to-end step with the record gather on RCCL inside it."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "nccl", "--steps", "3",
                        "--warmup", "1", "--spinup", "2", "--batch", "4", "--size", "512", "--batches", "2", "--no-cpu-baseline",
                        "--no-extras"], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "backend=nccl" in out["config"]["parallelism"]
    assert "detect_stream" in out["config"]["driver"]
    assert out["value"] > 0 and out["config"]["blocks_per_page"] > 0
