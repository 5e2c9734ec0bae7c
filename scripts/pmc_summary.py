#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs under a directory into a
per-kernel table (sum and per-dispatch mean of every counter)."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")[:100]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
lines = []
for k in agg:
    lines.append(k)
    for c, v in sorted(agg[k].items()):
        lines.append("    %-34s total %.5g  per-dispatch %.5g  (n=%d)" % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
