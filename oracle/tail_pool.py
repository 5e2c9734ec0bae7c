"""TEST / BASELINE INFRASTRUCTURE (never on the product path): the oracle tail (`postproc_ref.detector_tail`, the
reference's post-processing restated in numpy: inference.py:148-178) for several pages at once on a pool of host
processes -- the CPU baseline's tail leg at the same core budget as its forward leg (bench.py `cpu_baseline`).

Run as a child process so that the pool can be forked from an interpreter that holds neither torch nor a HIP runtime:

    python -m oracle.tail_pool <inputs.npz> <processes> [<input_size>]

inputs.npz: pages (n,H,W,3) u8, blks (n,rows,no) f32, mask (n,1,Hn,Wn) f32, lines (n,2,Hn,Wn) f32.  Prints one JSON line:
wall-clock of the pooled run (pool already started), per-page single-process times, blocks / lines found."""
from __future__ import annotations

import json
import multiprocessing as mp
import sys
import time

import numpy as np

_D = {}


def _one(i: int):
    from oracle import postproc_ref as R
    d, size = _D["d"], _D["size"]
    t0 = time.perf_counter()
    _, _, blk_list = R.detector_tail(d["pages"][i], d["blks"][i: i + 1], d["mask"][i: i + 1], d["lines"][i: i + 1],
                                     input_size=(size, size), refine_mode=0, keep_undetected_mask=False)
    return time.perf_counter() - t0, len(blk_list), sum(len(b.lines) for b in blk_list)


def _noop(_):
    return 0


def main(argv) -> None:
    path, procs = argv[0], max(1, int(argv[1]))
    d = dict(np.load(path))
    _D["d"], _D["size"] = d, int(argv[2]) if len(argv) > 2 else int(d["mask"].shape[-1])
    n = len(d["pages"])
    from oracle import postproc_ref  # noqa: F401   (imported before the fork: the children inherit it)
    with mp.get_context("fork").Pool(min(procs, n)) as pool:
        pool.map(_noop, range(4 * procs))            # every worker is up
        t0 = time.perf_counter()
        res = pool.map(_one, range(n), chunksize=1)
        wall = time.perf_counter() - t0
    print(json.dumps({"pages": n, "processes": min(procs, n), "wall_s": wall, "per_page_single_s": [r[0] for r in res],
                      "blocks": [r[1] for r in res], "lines": [r[2] for r in res]}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
