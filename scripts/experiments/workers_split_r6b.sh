#!/bin/bash
# Round 6, last session: workers x work items per batch again, after the tail's per-item latency went down (grouping stage in
# parallel) and after tail_delay.sh showed the step following the item latency.  One box, interleaved, 60 steps.
export TMPDIR=/tmp
O=${1:-gpurun_out/ws6b}; mkdir -p "$O"; shift
run() {
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-60} --warmup 5 --workers $2 --tail-split $3 --depth $4 "${@:5}" > "$O/$1.json" 2> "$O/$1.err"
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:16s} {d['value']:8.1f} pages/s {d['ms_per_step']:7.3f} ms | net ms {(d.get('roofline') or {}).get('net_ms_per_step')} | host cores {d['config'].get('host_cpu_cores_used')} | deliveries {d['config'].get('result_delivery_intervals')}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
run w4s4d4_$rep 4 4 4 "$@"
run w5s5d4_$rep 5 5 4 "$@"
run w6s6d4_$rep 6 6 4 "$@"
run w8s8d4_$rep 8 8 4 "$@"
run w4s8d4_$rep 4 8 4 "$@"
run w6s4d4_$rep 6 4 4 "$@"
run w6s6d6_$rep 6 6 6 "$@"
run w3s3d4_$rep 3 3 4 "$@"
done
