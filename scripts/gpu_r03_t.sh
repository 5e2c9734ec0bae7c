#!/bin/bash
# round 3, GPU call T: kernel TIMELINE of the end-to-end bench (rocprofv3 --kernel-trace, csv): where do the 3.5 ms between the
# network-only step and the end-to-end step go -- overlap, gaps, or stretched kernels?
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03t
mkdir -p $O
cd /tmp
for mode in e2e net; do
  extra=""; [ $mode = net ] && extra="--mode net"
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -o $mode -- python $ROOT/bench.py $extra --steps 12 --warmup 4 --spinup 0 --no-cpu-baseline --no-extras > $O/$mode.log 2>&1
  echo "$mode rc=$?"; tail -c 300 $O/$mode.log | head -c 300; echo
done
find $O -name "*kernel_trace.csv" | while read f; do python3 - "$f" <<'PY'
import sys, csv, gzip
f = sys.argv[1]
rows = list(csv.DictReader(open(f)))
keep = ("Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id")
out = f.replace("kernel_trace.csv", "kernel_trace_slim.csv.gz")
with gzip.open(out, "wt") as g:
    w = csv.writer(g)
    cols = [c for c in keep if c in rows[0]]
    w.writerow(cols)
    for r in rows:
        n = r["Kernel_Name"]
        n = n.replace("(anonymous namespace)::", "")[:60]
        w.writerow([n if c == "Kernel_Name" else r[c] for c in cols])
print(out, len(rows))
PY
rm -f "$f"; done
ls -la $O/*/* | head
