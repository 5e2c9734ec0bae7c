#!/bin/bash
# A/B of pipeline / tail knobs on the headline workload inside ONE box (boxes differ by a few %): each line = pages/s, ms per step.
# usage (GPU box): bash scripts/experiments/e2e_knobs.sh            (edit the list below)
run() { echo "$1 $2: $(CTD_TUNING=$1 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
run tail_chain=1
run tail_chain=1 "--tail-split 1"
run tail_chain=1 "--tail-split 2"
run tail_chain=1 "--tail-split 6"
run tail_chain=1 "--tail-split 1 --depth 6"
run halo3=0
run tail_fused_rounds=0
run tail_chain=1
