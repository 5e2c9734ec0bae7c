#!/bin/bash
# round 3, GPU call K: tail kernels confined to a CU subset (hipExtStreamCreateWithCUMask) next to the forward
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k
mkdir -p $O
run() { name=$1; shift; env "$@" timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); print("$name", d["value"], d["ms_per_step"])
except Exception as e: print("$name FAILED", e, open("$O/$name.err").read()[-500:])
PY
}
run base_a X=1
run fwdstream BENCH_FWD_STREAM=1
run cus32 BENCH_FWD_STREAM=1 CTD_TUNING=tail_cus=32
run cus64 BENCH_FWD_STREAM=1 CTD_TUNING=tail_cus=64
run cus128 BENCH_FWD_STREAM=1 CTD_TUNING=tail_cus=128
run cus16 BENCH_FWD_STREAM=1 CTD_TUNING=tail_cus=16
run base_b X=1
# are the in-process sub-runs slow because of the CPU legs before them?
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --rocm-timeout 0 > $O/bench_nocpu_extras.json 2> $O/bench_nocpu_extras.err
python -c "
import json; d=json.load(open('$O/bench_nocpu_extras.json')); print('no CPU legs:', d['value'], d['parity_exact']['value'], {k:v.get('value') for k,v in d['extra_configs'].items()})"
