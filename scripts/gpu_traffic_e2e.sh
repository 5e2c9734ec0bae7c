#!/bin/bash
# HBM bytes per kernel family of the END-TO-END step (FETCH_SIZE / WRITE_SIZE in their own --pmc passes, --kernel-trace only,
# as scripts/gpu_traffic.sh does for the network alone): how much memory traffic the tail's kernels add next to the
# forward's.  Counter passes serialise the kernels, so these are each kernel's own bytes, not a timing.
#   usage: bash scripts/gpu_traffic_e2e.sh <outname>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-traffic_e2e}
mkdir -p "$O"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/$C" -o $C -- \
     python "$ROOT/bench.py" --steps 6 --warmup 2 --spinup 4 --no-cpu-baseline --no-extras > "$O/$C.log" 2>&1
  echo "$C rc=$?"
done
python3 - "$O" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
def family(k):
    if any(s in k for s in ("conv_", "c3b_", "c3_fused", "stem_", "seg_final", "db_up", "sppf", "avgpool", "detect_decode")): return "network"
    if "tw_" in k: return "tail: refine windows (tw_*)"
    if "ccl2_" in k: return "tail: DB labelling (ccl2_*)"
    if "ccl_" in k: return "tail: refine labelling (ccl_*)"
    if "dbc_" in k: return "tail: DB tables (dbc_*)"
    if "nms" in k: return "tail: nms"
    if "copyBuffer" in k or "multi_copy" in k or "fillBuffer" in k or "copy" in k: return "tail: copies / fills"
    return "other"
kb = collections.defaultdict(lambda: collections.defaultdict(float)); per_k = collections.defaultdict(lambda: collections.defaultdict(float))
fwd = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"]
            kb[family(k)][c] += float(r["Counter_Value"])
            per_k[k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:56]][c] += float(r["Counter_Value"])
            if "stem_conv2" in k: fwd[c] += 1
n = max(fwd["FETCH_SIZE"], 1)
# MI355X_MICROARCH.md (HBM): counters in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so the
# corrected figure doubles it -- an upper bound for the tail's narrow (1-4 B per lane) reads, which the uncorrected one bounds from below
res = {"forwards_in_run": n, "unit": "GB per step (32 pages)", "families": {}}
lines = [f"{'family':34s} {'read GB (x2)':>13s} {'read GB (raw)':>14s} {'write GB':>9s}"]
for fam, v in sorted(kb.items(), key=lambda kv: -(2 * kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    rd, wr = v["FETCH_SIZE"] * 1024 / n / 1e9, v["WRITE_SIZE"] * 1024 / n / 1e9
    res["families"][fam] = {"read_gb_corrected": round(2 * rd, 3), "read_gb_uncorrected": round(rd, 3), "write_gb": round(wr, 3)}
    lines.append(f"{fam:34s} {2 * rd:13.2f} {rd:14.2f} {wr:9.2f}")
lines.append("")
lines.append(f"{'kernel (tail, by bytes)':58s} {'read GB (raw)':>14s} {'write GB':>9s}")
for k, v in sorted(per_k.items(), key=lambda kv: -(kv[1]['FETCH_SIZE'] + kv[1]['WRITE_SIZE'])):
    if family(k) == "network": continue
    lines.append(f"{k:58s} {v['FETCH_SIZE'] * 1024 / n / 1e9:14.3f} {v['WRITE_SIZE'] * 1024 / n / 1e9:9.3f}")
json.dump(res, open(out + "/traffic_e2e.json", "w"), indent=1)
open(out + "/summary.txt", "w").write("\n".join(lines[:60]) + "\n")
print("\n".join(lines[:48]))
PY
find "$O" -name "*_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*agent_info.csv" -delete
