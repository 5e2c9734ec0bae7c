#!/bin/bash
# The driver's line (`bench.py --steps 20 --warmup 5`, here without the baselines / sub-runs) N times in fresh processes on ONE box:
# value, ms per step and the intervals between result deliveries inside the timed region (config.result_delivery_intervals).
# Usage: bash scripts/experiments/headline_jitter.sh <out-name> [N] [extra bench.py args]
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/${1:-jitter}; N=${2:-8}; shift; shift
mkdir -p "$O"
thr() { awk '/nr_throttled/{n=$2} /throttled_usec/{u=$2} END{print n, u}' /sys/fs/cgroup/cpu.stat 2>/dev/null; }
for i in $(seq 1 "$N"); do
  T0=$(thr)
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > "$O/h$i.json" 2> "$O/h$i.err"
  python - "$O/h$i.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); c = d["config"]
    print("value", d["value"], "ms/step", d["ms_per_step"], "deliveries", c.get("result_delivery_intervals"), "host cores", c.get("host_cpu_cores_used"),
          "serial tail", (d.get("serial_step") or {}).get("tail_ms"))
except Exception as e:
    print("no JSON line:", e)
PY
  T1=$(thr); echo "   cgroup throttled periods / usec before: $T0  after: $T1 (whole process, start-up included)"
done
