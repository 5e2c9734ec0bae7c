#!/bin/bash
# round 3, GPU call J: PMC (MFMA busy, effective clock) of the split kernel on the three big ConvT shapes at B=32; the default
# bench line refreshed (sub-runs with proper spin-up, the canned-inputs sub-run)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03j
mkdir -p $O
export ST_SPLIT=1 ST_BATCH=32 PMC_ONLY_SQ1=1
CASES="16 17 18 10" bash scripts/gpu_pmc.sh > $O/pmc_split.txt 2>&1
for c in 16 17 18 10; do cp gpurun_out/pmc/case$c/summary.txt $O/pmc_split_case$c.txt 2>/dev/null; find gpurun_out/pmc/case$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pmc_split_case${c}_kernel_stats.csv 2>/dev/null; done
# kernel durations of the same dispatches (kernel trace of the sq1 pass)
python3 - <<'PY'
import csv, glob, collections
for c in (16, 17, 18, 10):
    d = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc/case{c}/sq1/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k, v in d.items():
        print(f"case {c} {k}: n={len(v)} mean {sum(v)/len(v):.4f} ms")
PY
rm -rf gpurun_out/pmc
unset ST_SPLIT ST_BATCH PMC_ONLY_SQ1
( time timeout 900 python bench.py --steps 20 --warmup 5 --dump-ops $O/per_op_fp16.tsv ) > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 200 $O/bench_n1.json; tail -4 $O/bench_n1.err
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], d['parity_exact']['value'], {k:v.get('value') for k,v in d['extra_configs'].items()})"
