#!/bin/bash
# round 3, GPU call Y: grid cap of the tail's big-grid kernels (they run next to the network): parity tests + A/B of the cap
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03y
mkdir -p $O

for w in 768 1024 1536 2048 768 1024 1536 2048; do
CTD_TUNING=tail_max_blocks=$w timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_cap${w}.json 2> $O/bench_cap${w}.err
python -c "import json;d=json.load(open('$O/bench_cap${w}.json'));print('fp16 e2e cap=$w',d['value'],d['ms_per_step'])"
done
