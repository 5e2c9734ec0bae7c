"""Kernel timeline of a bench run (rocprofv3 --kernel-trace csv, slimmed by scripts/gpu_r03_t.sh): per step, how long the
network's kernels take next to the tail's, where the gaps are, and one steady step kernel by kernel.
usage: python scripts/timeline_summary.py gpurun_out/r03t  [step index]"""
import collections
import csv
import gzip
import sys

FWD = ("conv_", "stem_", "c3_fused", "seg_final", "db_up", "sppf", "avgpool", "detect_decode", "export", "input_", "maxpool")


def load(f):
    rows = list(csv.DictReader(gzip.open(f, "rt")))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    return rows


def union(rs):
    if not rs:
        return 0
    iv = sorted((r["s"], r["e"]) for r in rs)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + ce - cs


def short(n):
    return n.replace("void ", "").replace("_ZN12_GLOBAL__N_1", "")[:34]


def main(root, step=None):
    per_kernel = {}
    for mode in ("e2e", "net"):
        rows = load(f"{root}/{mode}/{mode}_kernel_trace_slim.csv.gz")
        st = [r["s"] for r in rows if "stem_conv2" in r["Kernel_Name"] or "stem_split" in r["Kernel_Name"]]
        iv = [(st[i + 1] - st[i]) / 1e6 for i in range(len(st) - 1)]
        steady = sorted(iv)[len(iv) // 2]
        good = [i for i, x in enumerate(iv) if x < 1.15 * steady]           # steps without a trace / warm-up hiccup
        print(f"{mode}: {len(st)} steps, intervals ms {[round(x, 2) for x in iv]}, median {steady:.2f}")
        isf = lambda r: any(k in r["Kernel_Name"] for k in FWD)            # noqa: E731
        fsum = tsum = fun = 0.0
        d = collections.defaultdict(float)
        for i in good:
            win = [r for r in rows if st[i] <= r["s"] < st[i + 1]]
            fw = [r for r in win if isf(r)]
            fsum += sum(r["e"] - r["s"] for r in fw)
            tsum += sum(r["e"] - r["s"] for r in win if not isf(r))
            fun += union(fw)
            for r in fw:
                d[short(r["Kernel_Name"])] += r["e"] - r["s"]
        n = len(good)
        print(f"  over {n} steady steps: network kernels {fsum / n / 1e6:.3f} ms per step (union {fun / n / 1e6:.3f}), "
              f"other kernels {tsum / n / 1e6:.3f} ms per step")
        per_kernel[mode] = {k: v / n / 1e6 for k, v in d.items()}
        if mode == "e2e":
            i = good[len(good) // 2] if step is None else step
            a, b = st[i], st[i + 1]
            fq = next(r["Queue_Id"] for r in rows if r["s"] == a)
            print(f"  step {i} ({(b - a) / 1e6:.2f} ms): network kernels > 200 us and other kernels > 60 us")
            for r in rows:
                if r["e"] > a and r["s"] < b:
                    dur = (r["e"] - r["s"]) / 1e3
                    if (r["Queue_Id"] == fq and dur > 200) or (r["Queue_Id"] != fq and dur > 60):
                        print(f"   {(r['s'] - a) / 1e3:9.1f} us +{dur:8.1f} us  q{r['Queue_Id']} {short(r['Kernel_Name'])}")
    print("network kernels, ms per step: end to end vs network only")
    for k in sorted(per_kernel["e2e"], key=lambda k: -per_kernel["e2e"][k])[:12]:
        print(f"   {k:<36s} {per_kernel['e2e'][k]:.3f}  {per_kernel['net'].get(k, 0):.3f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
