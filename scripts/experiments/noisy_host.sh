#!/bin/bash
# The driver's line next to a busy host: L spinning shell loops (the box's 256 logical CPUs are shared with other tenants on the
# driver's runs; one such run read 2888 pages/s with the serial tail at 13.1 ms instead of 9.5) for depth / workers / tail-split
# settings, each twice, interleaved.  Usage: bash scripts/experiments/noisy_host.sh <out-name> <L>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/${1:-noisy}; L=${2:-192}
mkdir -p "$O"
PIDS=""
for i in $(seq 1 "$L"); do ( while :; do :; done ) & PIDS="$PIDS $!"; done
trap 'kill $PIDS 2>/dev/null' EXIT
sleep 1
one() {  # name, args
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras $2 > "$O/$1.json" 2> "$O/$1.err"
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c = d["config"]
    print("%-14s value %7.1f ms/step %6.3f deliveries %s host cores %s serial tail %s" % (sys.argv[2], d["value"], d["ms_per_step"], c.get("result_delivery_intervals"),
          c.get("host_cpu_cores_used"), (d.get("serial_step") or {}).get("tail_ms")))
except Exception as e:
    print(sys.argv[2], "no JSON line:", e)
PY
}
for rep in 1 2; do
  one "d4w4_$rep" "--depth 4"
  one "d6w4_$rep" "--depth 6"
  one "d8w4_$rep" "--depth 8"
  one "d6w6_$rep" "--depth 6 --workers 6"
  one "d8w8_$rep" "--depth 8 --workers 8"
done
kill $PIDS 2>/dev/null
echo "--- quiet host again"
one "quiet_d4" "--depth 4"
one "quiet_d6" "--depth 6"
one "quiet_d8" "--depth 8"
