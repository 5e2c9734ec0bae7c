"""Does a second pipeline in the same process run slower than the first?  (Round 3 measured 12-18 % and moved bench.py's
sub-runs into child processes; VERDICT r3 asked for the cause.)  Runs the end-to-end pipeline of bench.py several times in
ONE process on one detector: fresh worker pool each time (the old pool's threads and their native tails are gone before
the next starts), then with the old pools kept alive, then with a second detector.  Prints pages/s of every run.
usage: python scripts/gpu_inprocess.py [steps]      (GPU_MAX_HW_QUEUES etc. from the environment)"""
import gc
import importlib
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402

pkg = importlib.import_module("comic-text-detector_amd")
D = importlib.import_module("comic-text-detector_amd.dist")
DET = importlib.import_module("comic-text-detector_amd.detector")
TL = importlib.import_module("comic-text-detector_amd.tail")

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
args = types.SimpleNamespace(size=1024, tail_input="forward", dense_blocks=False, line_density="fixture", batches=4)
tb = bench.thread_budget(1)
TL.set_host_threads(tb["native_threads_per_worker"])
ckpt, batches, canned, _ = bench.make_workload(pkg, args, 0, 32, dev)
det = DET.TextDetector(ckpt, input_size=1024, device=dev, precision="fp16")


def one(d, keep=None, spin=40):
    pipe = bench.Pipeline(d, batches, None, dev, 1, 0, 32, D, 3, 4, 3)
    dt = bench.timed(pipe.run, steps, 5, spin, 1, dev, pipe.stats)
    gc.unfreeze()
    if keep is None:
        pipe.close()
    else:
        keep.append(pipe)
    return round(32 * steps / dt, 1)


print("queues:", os.environ.get("GPU_MAX_HW_QUEUES", "(default)"))
print("fresh pool each time, one detector:", [one(det) for _ in range(4)])
kept = []
print("old pools kept alive (idle threads, tails, streams):", [one(det, kept) for _ in range(3)])
for p in kept:
    p.close()
kept.clear()
gc.collect()
print("after closing them:", [one(det) for _ in range(2)])
det2 = DET.TextDetector(ckpt, input_size=1024, device=dev, precision="fp32s")
print("a second detector (fp32s) in the process, its own runs:", [one(det2) for _ in range(2)])
print("first detector again, second one alive:", [one(det) for _ in range(2)])
del det2
gc.collect()
torch.cuda.empty_cache()
print("first detector again, second one deleted:", [one(det) for _ in range(2)])
