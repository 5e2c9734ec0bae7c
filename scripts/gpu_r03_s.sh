#!/bin/bash
# round 3, GPU call S: the fp32s first layer straight from the page (selftest vs INPUT + generic kernel; engine A/B; parity tests)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03s
mkdir -p $O
( cd comic-text-detector_amd && ST_SPLIT=1 ST_CASES=1 timeout 300 ./ctd_selftest 32 ) > $O/split_selftest_b32.txt 2>&1
grep -E "^\[stem-split\]|selftest" $O/split_selftest_b32.txt | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_accept.py tests/test_gpu_edge.py -m gpu -q -k "fp32s or split" > $O/pytest_fp32s.txt 2>&1
tail -4 $O/pytest_fp32s.txt | cut -c1-300
for w in 0 1; do
CTD_TUNING=split_stem=$w timeout 300 python bench.py --precision fp32s --mode net --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras --dump-ops $O/per_op_fp32s_stem$w.tsv > $O/bench_fp32s_net_stem$w.json 2> $O/bench_fp32s_net_stem$w.err
python -c "import json;d=json.load(open('$O/bench_fp32s_net_stem$w.json'));print('stem=$w net',d['value'],d['ms_per_step'],d['roofline']['net_ms_per_step'])"
done
timeout 300 python bench.py --precision fp32s --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras > $O/bench_fp32s_e2e.json 2> $O/bench_fp32s_e2e.err
python -c "import json;d=json.load(open('$O/bench_fp32s_e2e.json'));print('fp32s e2e',d['value'],d['ms_per_step'])"
