#!/bin/bash
# round 3, GPU call N: HIP hardware queues.  ROCm maps a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; the
# pipeline has 1 forward stream + 3 tail streams (+ 2 loader streams with --host-input, + 1 per extra worker).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03n
mkdir -p $O
run() { name=$1; shift; envs=""; while [[ "$1" == *=* ]]; do envs="$envs $1"; shift; done; env $envs timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); print("$name", d["value"], d["ms_per_step"])
except Exception as e: print("$name FAILED", e, open("$O/$name.err").read()[-400:])
PY
}
run q4_w3 X=1
run q8_w3 GPU_MAX_HW_QUEUES=8
run q4_w4 X=1 --workers 4 --tail-split 4
run q8_w4 GPU_MAX_HW_QUEUES=8 --workers 4 --tail-split 4
run q8_w6 GPU_MAX_HW_QUEUES=8 --workers 6 --tail-split 6
run q16_w6 GPU_MAX_HW_QUEUES=16 --workers 6 --tail-split 6
run q4_host X=1 --host-input
run q8_host GPU_MAX_HW_QUEUES=8 --host-input
run q8_w3_b GPU_MAX_HW_QUEUES=8
run q2_w3 GPU_MAX_HW_QUEUES=2
