#!/bin/bash
# round 3, GPU call H: band dump (after the stem change) + tail-gate A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03h
mkdir -p $O
timeout 300 python scripts/gpu_band_dump.py > $O/band_dump.txt 2>&1
tail -3 $O/band_dump.txt | cut -c1-200
run() { name=$1; shift; timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); print("$name", d["value"], d["ms_per_step"])
except Exception as e: print("$name FAILED", e, open("$O/$name.err").read()[-600:])
PY
}
run gate0_a --tail-gate 0
run gate1_a --tail-gate 1
run gate0_b --tail-gate 0
run gate1_b --tail-gate 1
run gate1_fp32s --tail-gate 1 --precision fp32s
run gate0_fp32s --tail-gate 0 --precision fp32s
run gate1_split1 --tail-gate 1 --tail-split 1
