"""The HIP forward against the oracle network over MANY checkpoints, activations, shapes and batch sizes (round 6): for seed after
seed a random-init checkpoint in the reference's format (synth.make_checkpoint(seed, act)), a random input batch of a random
shape (multiples of 64, 128 .. 640), every engine -- maximum absolute difference of the sigmoid maps and the Detect rows against
OracleNet (CPU fp32, bit-exact with the reference's own modules).  NET_SWEEP_N (default 24), NET_SWEEP_SEED."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import pkg          # noqa: E402
from oracle.net_ref import OracleNet      # noqa: E402

p = pkg()
n = int(os.environ.get("NET_SWEEP_N", "24"))
rng = np.random.RandomState(int(os.environ.get("NET_SWEEP_SEED", "1")))
torch.set_num_threads(16)
worst = {"fp32": [0, 0, 0], "fp32s": [0, 0, 0], "fp16": [0, 0, 0]}
fails = 0
for i in range(n):
    seed = int(rng.randint(0, 10000))
    act = ["leaky", "silu", "relu"][i % 3]
    H, W = 64 * int(rng.randint(2, 11)), 64 * int(rng.randint(2, 11))
    B = int(rng.choice([1, 2, 3, 5]))
    ck = p.synth.make_checkpoint(seed, act=act)
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(seed))
    rb, rm, rl = [t.numpy() for t in OracleNet(ck, act=act)(x)]
    line = f"case {i}: ckpt seed {seed} act {act} input {B}x3x{H}x{W}:"
    for prec, (tm, tb) in (("fp32", (2e-5, 1e-4)), ("fp32s", (2e-5, 1e-4)), ("fp16", (2e-2, None))):
        be = p.backend.HipTextDetBackend(ck, device="cuda", precision=prec, act=act)
        blks, mask, lines = be(x.cuda())
        torch.cuda.synchronize()
        dm = float(np.abs(mask.cpu().numpy() - rm).max())
        dl = float(np.abs(lines.cpu().numpy() - rl).max())
        gb = blks.cpu().numpy()
        db = float((np.abs(gb - rb) / (1e-4 * np.abs(rb) + 2e-3)).max()) if tb else float(np.abs(gb - rb).max())
        w = worst[prec]
        w[0], w[1], w[2] = max(w[0], dm), max(w[1], dl), max(w[2], db)
        ok = dm <= tm and dl <= tm and (db <= 1.0 if tb else True)
        fails += not ok
        line += f"  {prec} mask {dm:.2e} lines {dl:.2e} det {db:.2e}{'' if ok else ' FAIL'}"
        be.close() if hasattr(be, "close") else None
        del be
    print(line, flush=True)
print("worst over the sweep:", {k: [float(f"{v:.3g}") for v in w] for k, w in worst.items()},
      "(mask, lines: max abs; det: max of |d| / (1e-4 |ref| + 2e-3) for the fp32-level engines, max abs for fp16)")
print(f"net sweep: {n} cases, {fails} outside the test tolerances")
sys.exit(1 if fails else 0)
