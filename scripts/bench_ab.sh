#!/bin/bash
# A/B of an environment knob on the bench workload: bash scripts/bench_ab.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 250 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null > /tmp/ab.json
  python - "$VAR=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
r = d["roofline"]
print(sys.argv[1], "pages/s", d["value"], "ms/step", d["ms_per_step"], "net", r["net_ms_per_step"], "family", r["family_ms_per_step"])
PY
done
