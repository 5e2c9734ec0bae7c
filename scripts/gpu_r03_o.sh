#!/bin/bash
# round 3, GPU call O: the tail alone on the real chain's inputs under rocprofv3 (kernel times without the forward next to them)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o tail -- python $GRAFT_REPO_ROOT/scripts/gpu_tail_prof_real.py > $GRAFT_REPO_ROOT/$O/tail_alone.json 2> $GRAFT_REPO_ROOT/$O/tail_alone.err
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_trace.csv" -delete
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_tail_alone.csv
rm -rf $O/prof
tail -1 $O/tail_alone.json
python3 - <<'PY'
import csv,re
rows=list(csv.DictReader(open("gpurun_out/r03o/rocprofv3_kernel_stats_tail_alone.csv")))
nb=13.0   # 3 warm-up + 10 timed batches
g={}
for r in rows:
    n=r["Name"]
    if any(k in n for k in ("conv_","stem_","c3_fused","seg_final","db_up","sppf","avgpool","detect_decode")): continue
    m=re.search(r"([A-Za-z0-9_]+)(<[^>]*>)?\(", n.replace("(anonymous namespace)::",""))
    k=m.group(1) if m else n[:30]
    e=g.setdefault(k,[0,0]); e[0]+=float(r["TotalDurationNs"])/1e6; e[1]+=float(r["Calls"])
print("tail kernel time per batch (alone): %.3f ms"%(sum(v[0] for v in g.values())/nb))
for k,(t,c) in sorted(g.items(), key=lambda x:-x[1][0])[:24]: print("  %-30s %.3f ms/batch  %5.1f calls  avg %.1f us"%(k[:30],t/nb,c/nb,t/c*1e3))
PY
