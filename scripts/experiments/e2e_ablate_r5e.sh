#!/bin/bash
# Round 5: which part of the tail costs the forward how much -- the end-to-end step with the whole tail, without the refine
# stage ("tail_ablate" = 1: NMS + DB stage + grouping only; a measurement knob, the results are incomplete), and the
# network + NMS alone.  One box, 60 timed steps each, interleaved.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$LABEL $*: $(python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
LABEL="whole tail" run
LABEL="no refine stage" CTD_TUNING=tail_ablate=1 run
LABEL="network + NMS" run --mode net
LABEL="whole tail" run
LABEL="no refine stage" CTD_TUNING=tail_ablate=1 run
LABEL="whole tail, dense pages" run --dense-blocks
LABEL="no refine stage, dense pages" CTD_TUNING=tail_ablate=1 run --dense-blocks
