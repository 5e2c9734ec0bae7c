#!/bin/bash
# Round 5 (the forward is 0.8 ms shorter than when the pipeline's knobs were last swept): workers / depth / work items per
# batch on the headline pages and the dense ones, one box, 40 timed steps each.  Each line = pages/s, ms per step, cores.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$*: $(python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
run
run --workers 3
run --workers 5
run --workers 6
run --depth 3
run --depth 6
run --workers 4 --tail-split 2
run --workers 4 --tail-split 8
run --lazy-blocks
run --dense-blocks
run --dense-blocks --workers 6
run
