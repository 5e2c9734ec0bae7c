#!/usr/bin/env python3
"""What does the N > 1 step's record gather cost on RCCL?  A world-size-1 `nccl` group on this GPU; `dist.gather_results` of 32
pages' native records on a side stream, timed (a) on an idle GPU, (b) while forwards are queued on the main stream, piece by
piece: pack, pin + upload, all_gather, the overflow test's device -> host read."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

pkg = importlib.import_module("comic-text-detector_amd")
D = importlib.import_module("comic-text-detector_amd.dist")


def main():
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29611"))
    backend = sys.argv[1] if len(sys.argv) > 1 else "nccl"
    D.init(backend, force=True)
    dev = torch.device("cuda", 0)
    ck = pkg.synth.make_blob_checkpoint(0, sparse_det=True, line_density="fixture")
    det = pkg.detector.TextDetector(ck, input_size=1024, device=dev, precision="fp16")
    x = torch.from_numpy(np.stack([pkg.synth.text_like_page((1024, 1024), i) for i in range(32)])).to(dev)
    pages = [x[i] for i in range(32)]
    job = det._forward(pages)
    res = det._tail(job, 0, False, records=(D.CAP_BLK, D.CAP_LINE))
    # the pipeline's other streams exist too: four tail workers' (streams beyond the runtime's hardware queues
    # of a class share a queue with an earlier one)
    pool = det._pool("tail", 4)
    det.warm_tails(pool, 4)
    prio = int(os.environ.get("PROBE_COMM_PRIO", "0"))
    comm = torch.cuda.Stream(dev, priority=prio)
    print(f"comm stream priority {prio}, TORCH_NCCL_HIGH_PRIORITY={os.environ.get('TORCH_NCCL_HIGH_PRIORITY')}", flush=True)
    for busy in (False, True):
        for rep in range(3):
            if busy:
                for _ in range(4):
                    det.net.forward_u8(x)                 # ~35 ms of forwards queued on the main stream
            t = [time.perf_counter()]
            with torch.cuda.stream(comm):
                rec = D.pack_results(res, None, D.CAP_BLK, D.CAP_LINE)
                t.append(time.perf_counter())
                recd = rec.pin_memory().to(dev, non_blocking=True)
                t.append(time.perf_counter())
                out = D.gather_records(recd, 32, 0, 1, True)
                t.append(time.perf_counter())
                ev = torch.cuda.Event()
                ev.record(comm)
                ev.synchronize()                           # when is the gather DONE (forwards still queued on the main stream)?
                over = False
                t.append(time.perf_counter())
            torch.cuda.synchronize()
            t.append(time.perf_counter())
            d = [round((b - a) * 1e3, 3) for a, b in zip(t, t[1:])]
            print(f"{backend} busy={busy} rep {rep}: pack {d[0]} ms, pin+upload {d[1]}, all_gather call {d[2]}, until the gather's event fired {d[3]}, "
                  f"final sync {d[4]}; over={over}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
