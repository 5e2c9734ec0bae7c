#!/bin/bash
# round 3, GPU call Z4: pages arriving as host arrays (--host-input: loader threads + copy streams join the four streams of the
# pipeline) with the default 4 hardware queues vs GPU_MAX_HW_QUEUES=8
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z4
mkdir -p $O
for q in 4 8 4 8; do
GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --host-input --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_host_q$q.json 2> $O/bench_host_q$q.err
python -c "import json;d=json.load(open('$O/bench_host_q$q.json'));print('host-input queues=$q',d['value'],d['ms_per_step'])"
done
