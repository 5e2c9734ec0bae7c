#!/bin/bash
# Round 5 experiment: partition the chip between the network and the tail instead of letting their kernels displace each other
# (DESIGN 4.9): tails' streams on T CUs (the same share of every XCD), the forwards' stream on the other 256 - T.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$*: $(timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>&1 | python -c "
import sys,json
ls=[l for l in sys.stdin.read().splitlines() if l.startswith('{')]
d=json.loads(ls[-1]) if ls else None
print((d['value'], d['ms_per_step']) if d else 'FAILED')")"; }
run
run --cu-split 16
run --cu-split 32
run --cu-split 48
run --cu-split 64
run --cu-split 32 --workers 4
run --cu-split 32 --dense-blocks
run --dense-blocks
run
