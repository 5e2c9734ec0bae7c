#!/bin/bash
# round 3, GPU call G: what the driver runs at round end (GPU suite, smoke, default bench line)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -3 $O/smoke.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 --dump-ops $O/per_op_fp16.tsv ) > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 300 $O/bench_n1.json; tail -4 $O/bench_n1.err
