#!/bin/bash
# round 3, GPU call U: the forward's kernels at raised wave priority (s_setprio 3) next to the tail: A/B on one box, both engines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03u
mkdir -p $O
for rep in 1 2; do
for w in 0 1; do
CTD_TUNING=fwd_prio=$w timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_prio${w}_$rep.json 2> $O/bench_prio${w}_$rep.err
python -c "import json;d=json.load(open('$O/bench_prio${w}_$rep.json'));print('fp16 e2e prio=$w',d['value'],d['ms_per_step'])"
done
done
for w in 0 1; do
CTD_TUNING=fwd_prio=$w timeout 200 python bench.py --mode net --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_net_prio$w.json 2> $O/bench_net_prio$w.err
python -c "import json;d=json.load(open('$O/bench_net_prio$w.json'));print('fp16 net prio=$w',d['value'],d['ms_per_step'])"
CTD_TUNING=fwd_prio=$w timeout 300 python bench.py --precision fp32s --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras > $O/bench_fp32s_prio$w.json 2> $O/bench_fp32s_prio$w.err
python -c "import json;d=json.load(open('$O/bench_fp32s_prio$w.json'));print('fp32s e2e prio=$w',d['value'],d['ms_per_step'])"
done
