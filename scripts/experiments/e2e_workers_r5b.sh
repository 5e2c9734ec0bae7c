#!/bin/bash
# Round 5, second look: 3 vs 4 tail workers on the dense-block and host-input pages, interleaved three times (the first A/B had
# them equal within a noisy +-6 %; the driver-style sub-runs of three boxes read lower with 3).
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$*: $(python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
for rep in 1 2 3; do
  for w in 3 4; do
    run --workers $w --dense-blocks
    run --workers $w --host-input
    run --workers $w
  done
done
