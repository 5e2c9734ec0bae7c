#!/bin/bash
# e2e bench: a batch's tail as 1 / 2 / 4 work items (page ranges), driver-length runs and long runs, one box.
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/split
mkdir -p $O
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "detect_stream" 2>&1 | tail -3
for S in 1 2 4; do
  for W in 3 4; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --tail-split $S --workers $W > $O/s${S}_w${W}.json 2>/dev/null
    echo "split $S workers $W steps 20: $(python3 -c "import json;d=json.load(open('$O/s${S}_w${W}.json'));print(d['value'], d['ms_per_step'])")"
  done
done
for S in 1 4; do
  timeout 300 python bench.py --gpus 1 --steps 300 --warmup 5 --no-cpu-baseline --tail-split $S --workers 4 > $O/long_s${S}.json 2>/dev/null
  echo "split $S workers 4 steps 300: $(python3 -c "import json;d=json.load(open('$O/long_s${S}.json'));print(d['value'], d['ms_per_step'])")"
done
