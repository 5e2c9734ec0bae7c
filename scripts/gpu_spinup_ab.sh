#!/bin/bash
# bench.py with the driver's arguments, cold (--spinup 0) and after the spin-up steps, on one box.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/spinup
mkdir -p $O
cd $ROOT
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline > $O/cold_$i.json 2>/dev/null; cut -c1-200 $O/cold_$i.json
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/spun_$i.json 2>/dev/null; cut -c1-200 $O/spun_$i.json
done
timeout 300 python bench.py --mode net --gpus 1 --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline > $O/net_cold.json 2>/dev/null; cut -c1-200 $O/net_cold.json
timeout 300 python bench.py --mode net --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/net_spun.json 2>/dev/null; cut -c1-200 $O/net_spun.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-ops $O/bench_per_op.tsv > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-200 $O/bench_n1.json
