#!/bin/bash
# Round 6: 4 / 5 / 6 tail workers on the headline and the dense pages after the merge stage moved into LDS (three interleaved repetitions)
export TMPDIR=/tmp
O=${1:-gpurun_out/workers_r6}; mkdir -p $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-80} --warmup 5 $2 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:16s} {d['value']:8.1f} pages/s {d['ms_per_step']:7.3f} ms | cores {d['config']['host_cpu_cores_used']} | {sys.argv[3]}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2 3; do
for w in 4 5 6; do
run head_w${w}_$rep "--workers $w"
run dense_w${w}_$rep "--dense-blocks --workers $w"
done
done
