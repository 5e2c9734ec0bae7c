#!/bin/bash
# Tail worker threads 4 / 5 / 6 on the headline pages and on the dense-block pages, one box, 40 timed steps each.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$*: $(python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
for w in 4 5 6; do run --workers $w; done
for w in 4 5 6; do run --dense-blocks --workers $w; done
run --workers 6 --tail-split 3
run --dense-blocks --workers 6 --depth 5
