#!/bin/bash
# round 3, GPU call X: threshold of the copy-engine rule for the tail's host copies (both directions)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03x
mkdir -p $O
for w in 262144 16384 1 262144 16384 1; do
CTD_TUNING=tail_dma_min=$w timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_dma${w}.json 2> $O/bench_dma${w}.err
python -c "import json;d=json.load(open('$O/bench_dma${w}.json'));print('fp16 e2e dma_min=$w',d['value'],d['ms_per_step'])"
done
