"""Child process of tests/test_gpu_rccl.py: ONE rank, a world-size-1 process group on RCCL (`backend="nccl"`), the N > 1
step's record gather on a communication stream from the page-locked record -- then the same over gloo, compared.
Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import pkg                                   # noqa: E402
from test_dist import _page_blocks                         # noqa: E402  (native group_output on the host: real block lists)


def gather(D, results, dev, stream):
    """the pipelined step's form: enqueued on the communication stream, resolved later"""
    n = len(results)
    with torch.cuda.stream(stream):
        h = D.gather_results_async(results, n, 0, 1, device=dev, pin=True, force=True)
    with torch.cuda.stream(stream):
        out = h.result()
    stream.synchronize()
    assert h.done()
    return out


def main():
    n_total = 5
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
    D = pkg().dist
    dev = torch.device("cuda", 0)
    rep = {}
    outs = {}
    for backend in ("nccl", "gloo"):
        D.init(backend, force=True)
        assert dist.is_initialized() and dist.get_backend() == backend and dist.get_world_size() == 1
        comm = torch.cuda.Stream(dev)
        for crowded in (0, n_total):
            results = [(None, None, _page_blocks(i, crowded)) for i in range(n_total)]
            out = gather(D, results, dev, comm)
            assert out.is_cuda == (backend == "nccl") and out.shape[0] == n_total      # over gloo the records stay on the host
            cap = D.MAX_BLK if crowded else D.CAP_BLK
            assert int(out[0, 2]) == cap, (backend, crowded, int(out[0, 2]))       # compact unless a page did not fit
            ref = D.pack_results(results, None, cap, D.MAX_BLK if crowded else D.CAP_LINE)
            assert torch.equal(out.cpu(), ref), (backend, crowded)
            got = D.unpack_results(out)
            assert [len(g) for g in got] == [len(r[2]) for r in results]
            outs[(backend, crowded)] = out.cpu()
        # the bench's timing reduction and barrier on this backend
        t = torch.tensor([1.25], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert float(t.item()) == 1.25
        rep[backend] = {"gathered_shape": list(outs[(backend, 0)].shape), "overflow_shape": list(outs[(backend, n_total)].shape)}
        dist.destroy_process_group()
    for crowded in (0, n_total):
        assert torch.equal(outs[("nccl", crowded)], outs[("gloo", crowded)])
    rep["equal"] = True
    print(json.dumps(rep), flush=True)


if __name__ == "__main__":
    main()
