#!/bin/bash
# round 3, GPU call W: big device -> host copies of the tail through the copy engines instead of a storing kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03w
mkdir -p $O
( cd comic-text-detector_amd && ST_CORUN=1 timeout 300 ./ctd_selftest 32 ) > $O/corun.txt 2>&1
grep -E "^\[corun\]" $O/corun.txt | grep "prio 1" | cut -c1-200
for rep in 1 2; do
for w in 1000000000000 262144; do
CTD_TUNING=tail_dma_min=$w timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_dma${w}_$rep.json 2> $O/bench_dma${w}_$rep.err
python -c "import json;d=json.load(open('$O/bench_dma${w}_$rep.json'));print('fp16 e2e dma_min=$w',d['value'],d['ms_per_step'])"
done
done
timeout 300 python bench.py --precision fp32s --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras > $O/bench_fp32s.json 2> $O/bench_fp32s.err
python -c "import json;d=json.load(open('$O/bench_fp32s.json'));print('fp32s e2e',d['value'],d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_post.py -m gpu -q 2>&1 | tail -2
