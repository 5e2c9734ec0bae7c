#!/bin/bash
# round 3, GPU call P: tw_accept_count with per-block LDS counting: parity tests, tail-alone profile, end-to-end bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03p
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_accept.py tests/test_gpu_post.py -m gpu -q > $O/pytest_e2e.txt 2>&1
tail -3 $O/pytest_e2e.txt | cut -c1-200
bash scripts/gpu_r03_o.sh 2>&1 | tail -12
for i in 1 2; do
timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_$i.json 2> $O/bench_$i.err
python -c "import json;d=json.load(open('$O/bench_$i.json'));print('e2e',d['value'],d['ms_per_step'])"
done
