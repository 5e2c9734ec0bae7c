#!/bin/bash
# Wave-slot occupancy per kernel of the END-TO-END step: SQ_WAVE_CYCLES (quad-cycles a wave is resident, summed over waves),
# SQ_BUSY_CYCLES, SQ_WAVES per dispatch, summed per kernel name over a short e2e bench run -- which of the tail's kernels hold
# the chip's wave slots.   usage: bash scripts/gpu_pmc_e2e.sh <outname>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-pmc_e2e}
mkdir -p "$O"
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/sq" -o sq -- python "$ROOT/bench.py" --steps 6 --warmup 2 --spinup 4 --no-cpu-baseline --no-extras > "$O/sq.log" 2>&1; echo "rc=$?"
python3 - "$O" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
tot = sum(v["SQ_WAVE_CYCLES"] for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])
lines = [f"{'kernel':62s} {'launches':>8s} {'wave-cycles %':>13s} {'Mquad-cyc':>10s} {'waves/launch':>12s} {'parked %':>9s}"]
for k, v in rows[:45]:
    lines.append(f"{k:62s} {cnt[k]:8d} {100 * v['SQ_WAVE_CYCLES'] / tot:13.2f} {v['SQ_WAVE_CYCLES'] / 1e6:10.1f} {v['SQ_WAVES'] / max(cnt[k], 1):12.0f} {100 * v['SQ_WAIT_ANY'] / max(v['SQ_WAVE_CYCLES'], 1):9.1f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf "$O/sq"
