#!/bin/bash
# round 3, GPU call B: eps-band / records / 2-rank tests, the new default bench line, the MIOpen baseline with a long limit
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_accept.py tests/test_dist.py -m gpu -q -s -k "band or native_page or two_ranks" > $O/pytest_band_dist.txt 2>&1
tail -25 $O/pytest_band_dist.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 --rocm-timeout 90 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 --tail-input canned --no-extras --no-cpu-baseline > $O/bench_canned.json 2> $O/bench_canned.err
tail -c 300 $O/bench_canned.json
timeout 700 python bench.py --mode rocm-baseline --rocm-budget 500 > $O/rocm_baseline.txt 2> $O/rocm_baseline.err
tail -3 $O/rocm_baseline.txt; tail -3 $O/rocm_baseline.err
