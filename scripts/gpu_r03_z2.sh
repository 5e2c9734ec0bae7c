#!/bin/bash
# round 3, GPU call Z2: stage 1 of concurrent tail work items chained on the GPU (tail_chain) vs side by side
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z2
mkdir -p $O
for w in 0 1 2 0 1 2; do
CTD_TUNING=tail_chain=$w timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_chain${w}.json 2> $O/bench_chain${w}.err
python -c "import json;d=json.load(open('$O/bench_chain${w}.json'));print('fp16 e2e tail_chain=$w',d['value'],d['ms_per_step'])"
done
