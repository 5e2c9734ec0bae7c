#!/bin/bash
# rocprofv3 kernel stats of HEAD (network-only and end-to-end bench, --spinup 0 so the files hold the warm-up + timed steps only)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/prof
mkdir -p $O
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/net -o net -- python $ROOT/bench.py --mode net --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline > $O/net.log 2>&1; echo rc=$?
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e -o e2e -- python $ROOT/bench.py --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline > $O/e2e.log 2>&1; echo rc=$?
rm -f $O/*/*kernel_trace.csv $O/*/*/*kernel_trace.csv
ls $O/net $O/e2e
