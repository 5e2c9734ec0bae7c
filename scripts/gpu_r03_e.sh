#!/bin/bash
# round 3, GPU call E: pipeline knobs A/B on the real-chain workload (tail stream priority, work items, depth)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras $EXTRA > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); print("$name", d["value"], d["ms_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
EXTRA=""
run prio0_a CTD_TUNING=tail_priority=0
run prio1 CTD_TUNING=tail_priority=1
run prio2 CTD_TUNING=tail_priority=2
run prio0_b CTD_TUNING=tail_priority=0
EXTRA="--tail-split 1" run split1 X=1
EXTRA="--tail-split 6" run split6 X=1
EXTRA="--workers 4 --tail-split 4" run w4 X=1
EXTRA="--workers 2 --tail-split 2" run w2 X=1
EXTRA="--depth 6" run depth6 X=1
EXTRA="--depth 3" run depth3 X=1
EXTRA="--mode net" run net X=1
EXTRA="--precision fp32s" run fp32s X=1
EXTRA="--precision fp32s --mode net" run fp32s_net X=1
