#!/bin/bash
# Board power / clocks while the hot kernels run (is the chip at its power cap?).  rocm-smi is sampled in the
# background (~3 Hz) during (a) the network-only bench, fused and unfused program, (b) one ConvTranspose layer of the
# selftest in a long loop on random and on all-zero operands.  Output: gpurun_out/power/*.txt + summary.json
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/power
mkdir -p $O
cd $ROOT
rocm-smi --showpower --showclocks --showtemp --showmaxpower --showperflevel > $O/idle.txt 2>&1
sample() {  # $1 = output file; runs until the file $1.stop exists
  while [ ! -e $1.stop ]; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (edge|junction|memory)" >> $1
    echo "---" >> $1
  done
}
run() {  # $1 = tag, rest = command
  tag=$1; shift
  rm -f $O/$tag.txt $O/$tag.txt.stop
  sample $O/$tag.txt &
  SP=$!
  "$@" > $O/$tag.out 2>&1
  touch $O/$tag.txt.stop
  wait $SP
  rm -f $O/$tag.txt.stop
  echo "$tag: $(grep -c -- '---' $O/$tag.txt) samples"
}
run net_fused   env CTD_FUSE=7 timeout 300 python bench.py --mode net --steps 1200 --warmup 3 --no-cpu-baseline
run net_unfused env CTD_FUSE=0 timeout 300 python bench.py --mode net --steps 1200 --warmup 3 --no-cpu-baseline
run e2e_fused   timeout 300 python bench.py --steps 600 --warmup 3 --no-cpu-baseline
run convt_random env ST_NO_C3=1 ST_CASES=16 ST_VAR=0 ST_ITERS=12000 timeout 120 ./comic-text-detector_amd/ctd_selftest 32
run convt_zero   env ST_NO_C3=1 ST_ZERO=1 ST_CASES=16 ST_VAR=0 ST_ITERS=12000 timeout 120 ./comic-text-detector_amd/ctd_selftest 32
run conv1x1_random env ST_NO_C3=1 ST_CASES=3 ST_VAR=0 ST_ITERS=40000 timeout 120 ./comic-text-detector_amd/ctd_selftest 32
python3 - <<'PY'
import glob, json, os, re
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "power")
res = {}
for f in sorted(glob.glob(O + "/*.txt")):
    tag = os.path.basename(f)[:-4]
    if tag == "idle":
        continue
    pw, sclk = [], []
    for ln in open(f):
        m = re.search(r"Power.*?:\s*([0-9.]+)", ln)
        if m: pw.append(float(m.group(1)))
        m = re.search(r"sclk.*?\((\d+)Mhz\)", ln)
        if m: sclk.append(int(m.group(1)))
    if pw:
        k = max(1, len(pw) // 4)           # steady state: drop the first quarter (ramp-up / process start)
        res[tag] = {"samples": len(pw), "power_W_mean_steady": round(sum(pw[k:]) / len(pw[k:]), 1), "power_W_max": max(pw),
                    "sclk_MHz_mean_steady": round(sum(sclk[k:]) / max(len(sclk[k:]), 1)) if sclk else None,
                    "sclk_MHz_min": min(sclk) if sclk else None, "sclk_MHz_max": max(sclk) if sclk else None}
    out = os.path.join(O, tag + ".out")
    if os.path.isfile(out):
        txt = open(out).read()
        m = re.search(r'"value": ([0-9.]+).*?"ms_per_step": ([0-9.]+)', txt)
        if m: res.setdefault(tag, {})["bench"] = {"pages_s": float(m.group(1)), "ms_per_step": float(m.group(2))}
        m = re.search(r"default: ok ([0-9.]+) ms ([0-9.]+) TF", txt)
        if m: res.setdefault(tag, {})["selftest"] = {"ms": float(m.group(1)), "TFLOPs": float(m.group(2))}
json.dump(res, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
head -40 $O/idle.txt
