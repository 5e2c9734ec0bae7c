"""Kernel timeline of bench runs (rocprofv3 --kernel-trace CSVs, raw or gzip'ed): per file, how long the network's
kernels take per step next to the tail's, which tail kernels carry the time, where the gaps are, and one steady step kernel
by kernel; with two or more files the network kernels' per-step times side by side (e.g. end to end vs network only).
usage: python scripts/timeline_summary.py <trace.csv[.gz]> [<trace2.csv> ...] [--step N] [--slim out.csv.gz]"""
import collections
import csv
import gzip
import sys

FWD = ("conv_", "stem_", "c3_fused", "c3b_", "seg_final", "db_up", "sppf", "avgpool", "detect_decode", "export", "input_", "maxpool")


def load(f):
    op = gzip.open if f.endswith(".gz") else open
    rows = list(csv.DictReader(op(f, "rt")))
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
    return n.replace("void ", "").replace("_ZN12_GLOBAL__N_1", "").replace("(anonymous namespace)::", "")[:40]


def isf(r):
    return any(k in r["Kernel_Name"] for k in FWD)


def one(f, step, listing):
    rows = load(f)
    st = [r["s"] for r in rows if "stem_conv2" in r["Kernel_Name"] or "stem_split" in r["Kernel_Name"] or "stem_mfma" in r["Kernel_Name"]]
    if len(st) < 3:
        print(f"{f}: fewer than 3 steps in the trace")
        return {}
    iv = [(st[i + 1] - st[i]) / 1e6 for i in range(len(st) - 1)]
    steady = sorted(iv)[len(iv) // 2]
    good = [i for i, x in enumerate(iv) if x < 1.15 * steady]           # steps without a trace / warm-up hiccup
    print(f"{f}: {len(st)} steps, intervals ms {[round(x, 2) for x in iv]}, median {steady:.2f}")
    fsum = tsum = fun = tun = 0.0
    d, dt, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
    for i in good:
        win = [r for r in rows if st[i] <= r["s"] < st[i + 1]]
        fw = [r for r in win if isf(r)]
        tl = [r for r in win if not isf(r)]
        fsum += sum(r["e"] - r["s"] for r in fw)
        tsum += sum(r["e"] - r["s"] for r in tl)
        fun += union(fw)
        tun += union(tl)
        for r in fw:
            d[short(r["Kernel_Name"])] += r["e"] - r["s"]
        for r in tl:
            dt[short(r["Kernel_Name"])] += r["e"] - r["s"]
            cnt[short(r["Kernel_Name"])] += 1
    n = len(good)
    print(f"  over {n} steady steps: network kernels {fsum / n / 1e6:.3f} ms per step (union {fun / n / 1e6:.3f}), "
          f"other kernels {tsum / n / 1e6:.3f} ms per step (union {tun / n / 1e6:.3f})")
    if dt:
        print("  other kernels, per step: ms (launches)")
        for k in sorted(dt, key=lambda k: -dt[k])[:18]:
            print(f"   {k:<42s} {dt[k] / n / 1e6:.3f} ({cnt[k] / n:.1f})")
    if listing:
        i = good[len(good) // 2] if step is None else step
        a, b = st[i], st[i + 1]
        fq = next(r["Queue_Id"] for r in rows if r["s"] == a)
        print(f"  step {i} ({(b - a) / 1e6:.2f} ms): network kernels > 200 us and other kernels > 60 us")
        for r in rows:
            if r["e"] > a and r["s"] < b:
                dur = (r["e"] - r["s"]) / 1e3
                if (r["Queue_Id"] == fq and dur > 200) or (r["Queue_Id"] != fq and dur > 60):
                    print(f"   {(r['s'] - a) / 1e3:9.1f} us +{dur:8.1f} us  q{r['Queue_Id']} {short(r['Kernel_Name'])}")
    return {k: v / n / 1e6 for k, v in d.items()}


def main(argv):
    step, files, slim = None, [], None
    it = iter(argv)
    for a in it:
        if a == "--step":
            step = int(next(it))
        elif a == "--slim":
            slim = next(it)
        else:
            files.append(a)
    if slim:                                                   # keep name / start / end / queue only
        rows = load(files[0])
        with gzip.open(slim, "wt") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"])
            for r in rows:
                w.writerow([r["Kernel_Name"][:60], r["s"], r["e"], r["Queue_Id"]])
    per = [one(f, step, listing=(k == 0)) for k, f in enumerate(files)]
    if len(per) > 1 and per[0]:
        print("network kernels, ms per step: " + " | ".join(files))
        for k in sorted(per[0], key=lambda k: -per[0][k])[:14]:
            print(f"   {k:<42s} " + "  ".join(f"{p.get(k, 0):.3f}" for p in per))


if __name__ == "__main__":
    main(sys.argv[1:])
