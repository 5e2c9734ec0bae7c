#!/bin/bash
# round 3, GPU call Z5: pipeline knobs again after the copy-engine / chain changes (one box)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z5
mkdir -p $O
run() { timeout 200 python bench.py --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras "$@" > $O/b.json 2> $O/b.err; python -c "import json,sys;d=json.load(open('$O/b.json'));print(' '.join(sys.argv[1:]),d['value'],d['ms_per_step'])" "$@"; }
run --tail-split 3
run --tail-split 2
run --tail-split 4
run --tail-split 6
run --workers 2 --tail-split 2
run --depth 6
run --tail-split 3
