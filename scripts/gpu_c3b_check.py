"""GPU: the bottleneck (+ cv3) kernel of the 64 / 128-channel C3 blocks (csrc/kernels_c3b.hip, fuse bit 8) against the
launches it replaces -- every activation tensor of the network compared bit for bit (engine without arena reuse, so
`read_tensor` sees every intermediate), the first differing tensors named; then per-op times of the chains at the
benchmark shape with the kernel on and off.

    python scripts/gpu_c3b_check.py [check] [time]
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("comic-text-detector_amd")
L = pkg._lib


def tune(key, value):
    L.check(L.lib().ctd_tuning_set(key.encode(), int(value)), "ctd_tuning_set " + key)


def check():
    ck = pkg.synth.make_checkpoint(0)
    tune("no_reuse", 1)
    be = pkg.backend.HipTextDetBackend(ck, device="cuda", precision="fp16")
    tune("no_reuse", 0)
    names = {tid: n for n, tid in be.program.taps.items()}
    writer = {}
    for o in be.program.ops:
        if o["dst"] >= 0:
            writer.setdefault(o["dst"], o["name"])
    bad = 0
    cases = [((2, 256, 256), 1024, 0, 0), ((2, 256, 256), 1, 0, 0), ((1, 320, 448), 1, 0, 0), ((3, 128, 192), 1024, 0, 0)]
    for c64, c128 in ((1, 1), (2, 1)):                   # the other tilings of the kernel
        cases += [((2, 256, 256), 1024, c64, c128), ((1, 320, 448), 1, c64, c128), ((3, 128, 192), 1, c64, c128)]
    for shape, halo_min, c64, c128 in cases:
        g = torch.Generator().manual_seed(7)
        x = torch.rand((shape[0], 3, shape[1], shape[2]), generator=g).cuda()
        tune("c3b_min_patches", 1)
        tune("halo_min_patches", halo_min)
        tune("c3b_cfg64", c64)
        tune("c3b_cfg128", c128)
        tune("halo3_min_blocks", 1)
        res = {}
        for fuse in (7, 63):
            tune("fuse", fuse)
            outs = [t.clone() for t in be(x)] + [be.mask_u8.clone(), be.bitmap.clone()]
            torch.cuda.synchronize()
            tens = {}
            for tid in range(len(be.program.tensors)):
                try:
                    tens[tid] = be.read_tensor(tid)
                except Exception:
                    pass
            res[fuse] = (outs, tens)
        tune("fuse", 63)
        tune("c3b_min_patches", 1024)
        tune("halo_min_patches", 1024)
        tune("halo3_min_blocks", 1024)
        tune("c3b_cfg64", 0)
        tune("c3b_cfg128", 1)
        nd = 0
        for tid in sorted(res[7][1]):
            a, b = res[7][1][tid], res[63][1][tid]
            if not np.array_equal(a, b, equal_nan=True):
                d = np.abs(a.astype(np.float64) - b.astype(np.float64))
                w = writer.get(tid, "?")
                # tensors the fused program never writes (t of a fused bottleneck, y1 overwritten in place) differ by design
                if os.environ.get("C3B_VERBOSE"):
                    print(f"  shape {shape} halo_min {halo_min}: tensor {tid} ({names.get(tid, '')}, written by {w}) differs: "
                          f"{int((d > 0).sum())} of {d.size} values, max |d| {np.nanmax(d):.4g}")
                nd += 1
        same = all(torch.equal(u, v) for u, v in zip(res[7][0], res[63][0]))
        print(f"shape {shape} halo_min_patches {halo_min} cfg64 {c64} cfg128 {c128}: network outputs identical: {same}; "
              f"{nd} intermediate tensors differ (by design: y1 / t of fused bottlenecks)")
        bad += 0 if same else 1
    print("C3B CHECK", "PASS" if bad == 0 else "FAIL")
    return bad


def time_chains():
    ck = pkg.synth.make_checkpoint(0)
    be = pkg.backend.HipTextDetBackend(ck, device="cuda", precision="fp16")
    x = torch.randint(0, 256, (32, 1024, 1024, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(1)).cuda()
    variants = [("unfused", 7, 0, 0), ("c3b 0/0", 15, 0, 0), ("c3b 0/1", 15, 0, 1), ("+post1x1", 31, 0, 1), ("+segtaps", 63, 0, 1)]
    rows = {}
    for name, fuse, c64, c128 in variants:
        tune("fuse", fuse)
        tune("c3b_cfg64", c64)
        tune("c3b_cfg128", c128)
        for _ in range(3):
            be.forward_u8(x)
        torch.cuda.synchronize()
        acc = None
        for _ in range(5):
            p = be.profile(x)
            acc = p["ms"] if acc is None else acc + p["ms"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            be.forward_u8(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(40):
            be.forward_u8(x)
        e1.record()
        torch.cuda.synchronize()
        rows[name] = (acc / 5, p["names"], e0.elapsed_time(e1) / 40)
    tune("fuse", 63)
    tune("c3b_cfg64", 0)
    tune("c3b_cfg128", 1)
    names = rows["unfused"][1]
    print(f"{'op (ms per 32 pages)':44s} " + " ".join(f"{n:>9s}" for n, *_ in variants))
    tot = np.zeros(len(variants))
    chain = np.zeros(len(variants))
    for i, n in enumerate(names):
        v = np.array([rows[k][0][i] for k, *_ in variants])
        tot += v
        if ".m." in n or n.endswith("cv3.conv") or n.endswith(".cv3"):
            chain += v
        if ((".m." in n and "cv1" in n) or n.endswith("conv.1") or "upconv5.conv.0.cv1+cv2" in n or n == "db.conv.0" or n == "seg.upconv6") and v.max() > 0.05:
            print(f"{n:44s} " + " ".join(f"{q:9.4f}" for q in v))
    print(f"{'bottleneck + cv3 ops':44s} " + " ".join(f"{q:9.4f}" for q in chain))
    print(f"{'all ops (sum of per-op events)':44s} " + " ".join(f"{q:9.4f}" for q in tot))
    print(f"{'forward, 40 back to back':44s} " + " ".join(f"{rows[k][2]:9.4f}" for k, *_ in variants))


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what:
        time_chains()
    sys.exit(1 if rc else 0)
