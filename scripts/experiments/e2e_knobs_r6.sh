#!/bin/bash
# Round 6: the pipeline's knobs re-measured after the merge stage moved into LDS (workers / depth / work items / the stage-1
# chain), one box, two interleaved repetitions.  Usage: gpurun -- 'bash scripts/experiments/e2e_knobs_r6.sh <out>'
export TMPDIR=/tmp
O=${1:-gpurun_out/e2e_knobs_r6}; mkdir -p "$O"
run() {  # name tuning args
  CTD_TUNING="$2" timeout 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-60} --warmup 5 $3 > "$O/$1.json" 2> "$O/$1.err"
  python - "$O/$1.json" "$1" "$2 $3" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:22s} {d['value']:8.1f} pages/s {d['ms_per_step']:7.3f} ms | cores {d['config']['host_cpu_cores_used']} | {sys.argv[3]}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
run base_$rep "" ""
run w3_$rep "" "--workers 3"
run w5_$rep "" "--workers 5"
run w6_s6_$rep "" "--workers 6 --tail-split 6"
run d3_$rep "" "--depth 3"
run d6_$rep "" "--depth 6"
run s8_$rep "" "--tail-split 8"
run s2_$rep "" "--tail-split 2"
run chain0_$rep "tail_chain=0" ""
run chain2_$rep "tail_chain=2" ""
run prio0_$rep "fwd_prio=0" ""
done
run dense_base "" "--dense-blocks"
run dense_w5 "" "--dense-blocks --workers 5"
run dense_s8 "" "--dense-blocks --tail-split 8"
run dense_chain0 "tail_chain=0" "--dense-blocks"
