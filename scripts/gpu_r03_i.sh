#!/bin/bash
# round 3, GPU call I: the GPU suite + smoke after the attribution gained the score-gate / contour-cut causes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
tail -2 $O/smoke.txt | cut -c1-200
timeout 600 python bench.py --steps 10 --warmup 3 --spinup 30 --no-extras > $O/bench_parity.json 2> $O/bench_parity.err
python -c "
import json; d=json.load(open('$O/bench_parity.json')); print(d['value'], d['parity']['fp16_band']['holds'], d['parity']['engines']['fp16']['band'])" | cut -c1-900
