"""Summary of a rocprofv3 --memory-copy-trace CSV: per direction the number of copies, bytes, busy time and rate.
usage: python scripts/memcpy_summary.py <memory_copy_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
if not rows:
    print("no copies")
    sys.exit(0)
keys = rows[0].keys()
print("columns:", list(keys))
d = collections.defaultdict(lambda: [0, 0, 0, []])
for r in rows:
    k = r.get("Direction") or r.get("Kind") or "?"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    b = int(r.get("Size") or r.get("Bytes") or 0) if (r.get("Size") or r.get("Bytes")) else 0
    d[k][0] += 1
    d[k][1] += b
    d[k][2] += e - s
    d[k][3].append((b, e - s))
t0, t1 = min(int(r["Start_Timestamp"]) for r in rows), max(int(r["End_Timestamp"]) for r in rows)
print(f"trace span {(t1 - t0) / 1e6:.1f} ms")
for k, (n, b, t, lst) in d.items():
    big = [(bb, tt) for bb, tt in lst if bb >= (1 << 20)]
    rate = sum(bb for bb, _ in big) / max(1, sum(tt for _, tt in big))
    print(f"{k}: {n} copies, {b / 1e6:.1f} MB, busy {t / 1e6:.2f} ms; copies >= 1 MB: {len(big)}, {rate:.1f} GB/s while active")
