#!/bin/bash
# e2e bench: a batch's tail as 1 / 2 / 3 / 4 work items (page ranges) on 3 workers, driver-length runs (x3) and long
# runs, one box.  (4 workers: 1611-1985 pages/s against 2324-2464 with 3 -- measured, cause not traced.)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/split
mkdir -p $O
cd $ROOT
for R in 1 2 3; do
  for S in 1 2 3 4; do
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --tail-split $S > $O/s${S}_r${R}.json 2>/dev/null
    echo "split $S run $R steps 20: $(python3 -c "import json;d=json.load(open('$O/s${S}_r${R}.json'));print(d['value'], d['ms_per_step'])")"
  done
done
for S in 1 2 4; do
  timeout 300 python bench.py --gpus 1 --steps 300 --warmup 5 --no-cpu-baseline --tail-split $S > $O/long_s${S}.json 2>/dev/null
  echo "split $S steps 300: $(python3 -c "import json;d=json.load(open('$O/long_s${S}.json'));print(d['value'], d['ms_per_step'])")"
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --tail-split 2 --depth 5 > $O/s2_d5.json 2>/dev/null
echo "split 2 depth 5: $(python3 -c "import json;d=json.load(open('$O/s2_d5.json'));print(d['value'], d['ms_per_step'])")"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --tail-split 2 --workers 2 > $O/s2_w2.json 2>/dev/null
echo "split 2 workers 2: $(python3 -c "import json;d=json.load(open('$O/s2_w2.json'));print(d['value'], d['ms_per_step'])")"
