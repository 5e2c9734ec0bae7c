#!/bin/bash
# round 3, GPU call M: the default bench line with child-process sub-runs
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03m
mkdir -p $O
( time timeout 1200 python bench.py --steps 20 --warmup 5 --dump-ops $O/per_op_fp16.tsv ) > $O/bench_n1.json 2> $O/bench_n1.err
tail -4 $O/bench_n1.err
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(d['value'], d['ms_per_step'], 'exact', d['parity_exact'], {k:(v.get('value'), v.get('error')) for k,v in d['extra_configs'].items()}, d['rocm_baseline'].get('note'))" | cut -c1-1500
