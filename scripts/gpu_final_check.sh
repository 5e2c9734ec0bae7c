#!/bin/bash
# Last look at HEAD: the staging / stream tests, the bench line (no CPU baseline: it alone takes 3 minutes) + per-op
# table, the network-only and the host-input lines.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/final
mkdir -p $O
cd $ROOT
timeout 120 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -2
timeout 120 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dump-ops $O/bench_per_op.tsv > $O/bench_n1_nocpu.json 2> $O/err.txt; echo rc=$?; cut -c1-200 $O/bench_n1_nocpu.json; tail -3 $O/err.txt
timeout 60 python bench.py --mode net --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_net.json 2>/dev/null; cut -c1-160 $O/bench_net.json
timeout 60 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --host-input > $O/bench_host.json 2>/dev/null; cut -c1-160 $O/bench_host.json
