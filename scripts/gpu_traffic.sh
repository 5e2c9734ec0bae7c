#!/bin/bash
# HBM traffic of the dominant kernel family from PMC counters (FETCH_SIZE / WRITE_SIZE in
# their own --pmc passes, --kernel-trace only), on the same workload as bench.py.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PREC=${1:-fp16}
OUT=$ROOT/gpurun_out/traffic_$PREC
mkdir -p $OUT
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o $C -- \
     python $ROOT/bench.py --precision $PREC --mode net --steps 1 --warmup 1 --spinup 0 --no-cpu-baseline --no-extras > $OUT/$C.log 2>&1
  echo "$C rc=$?"
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
python3 $ROOT/scripts/traffic_summary.py $OUT $PREC
