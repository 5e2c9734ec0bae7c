#!/bin/bash
# Round 5: 3 or 4 tail workers, now that the forward is 0.8 ms shorter (4 was +4 % on dense pages in round 4) -- headline,
# dense and canned pages, interleaved twice for noise.  Each line = pages/s, ms per step, CPU cores used.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$*: $(python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
for rep in 1 2; do
  for w in 3 4; do
    run --workers $w
    run --workers $w --dense-blocks
    run --workers $w --tail-input canned
    run --workers $w --host-input
  done
done
run --workers 3 --tail-split 4
run --workers 3 --tail-split 2
run --workers 2
