#!/bin/bash
# Board power / sclk (rocm-smi, ~3 Hz) under the fp32s forward and under the split haloed-patch kernel on one ConvT shape:
# is the exact engine at the power cap?  Output: gpurun_out/power_fp16/summary.json
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/power_fp16
mkdir -p $O
cd $ROOT
sample() { while [ ! -e $1.stop ]; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" >> $1; echo "---" >> $1; done; }
run() { tag=$1; shift; rm -f $O/$tag.txt $O/$tag.txt.stop; sample $O/$tag.txt & SP=$!; "$@" > $O/$tag.out 2>&1; touch $O/$tag.txt.stop; wait $SP; rm -f $O/$tag.txt.stop; echo "$tag: $(grep -c -- '---' $O/$tag.txt) samples"; }
run fp16_net timeout 200 python bench.py --mode net --steps 900 --warmup 3 --spinup 0 --no-cpu-baseline --no-extras
run fp16_e2e timeout 200 python bench.py --steps 700 --warmup 3 --spinup 0 --no-cpu-baseline --no-extras
python3 - <<'PY'
import glob, json, os, re
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "power_fp16")
res = {}
for f in sorted(glob.glob(O + "/*.txt")):
    tag = os.path.basename(f)[:-4]
    pw, sclk = [], []
    for ln in open(f):
        m = re.search(r"Power.*?:\s*([0-9.]+)", ln)
        if m: pw.append(float(m.group(1)))
        m = re.search(r"sclk.*?\((\d+)Mhz\)", ln)
        if m: sclk.append(int(m.group(1)))
    if pw:
        k = max(1, len(pw) // 3)
        hot = sorted(pw[k:])[len(pw[k:]) // 2:]                 # upper half of the steady samples: the loaded phase
        res[tag] = {"samples": len(pw), "power_W_median_loaded": round(sorted(hot)[len(hot) // 2], 1), "power_W_max": max(pw),
                    "sclk_MHz_mean_steady": round(sum(sclk[k:]) / max(len(sclk[k:]), 1)) if sclk else None,
                    "sclk_MHz_min": min(sclk) if sclk else None}
    out = os.path.join(O, tag + ".out")
    if os.path.isfile(out):
        m = re.search(r'"value": ([0-9.]+).*?"ms_per_step": ([0-9.]+)', open(out).read())
        if m: res.setdefault(tag, {})["bench"] = {"pages_s": float(m.group(1)), "ms_per_step": float(m.group(2))}
json.dump(res, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(res))
PY
