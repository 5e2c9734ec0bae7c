#!/bin/bash
# round 3, after tail_chain: the driver's bench line again + rocprofv3 kernel stats of the end-to-end run (HEAD)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT || exit 1
O=$ROOT/gpurun_out/r03final2
mkdir -p $O
( time timeout 1200 python bench.py --steps 20 --warmup 5 --dump-ops $O/per_op_fp16.tsv ) > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print('default', d['value'], d['ms_per_step'], 'exact', d['parity_exact'].get('value'), {k:(v.get('value'), v.get('error')) for k,v in d['extra_configs'].items()})" | cut -c1-600
tail -3 $O/bench_n1.err | cut -c1-200
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python $ROOT/bench.py --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extras > $O/prof_e2e.log 2>&1; echo e2e rc=$?
find $O -name "*kernel_trace.csv" -delete
