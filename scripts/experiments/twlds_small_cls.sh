#!/bin/bash
# Round 6, last session: can the window-local merge kernel's SMALL windows sit next to two 72-74-KB conv blocks on a CU?  (12-16 KB of
# LDS are free there; a launch class's blocks all take its largest window's LDS.)  Launch classes at 10 / 12 / 16 KB, with smaller
# run tables, on the end-to-end step; one box, interleaved.  Usage: gpurun -- 'bash scripts/experiments/twlds_small_cls.sh <out>'
export TMPDIR=/tmp
O=${1:-gpurun_out/twlds_small}; mkdir -p "$O"
run() {  # name tuning args
  CTD_TUNING="$2" timeout 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-60} --warmup 5 $3 > "$O/$1.json" 2> "$O/$1.err"
  python - "$O/$1.json" "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    s = d["serial_step"]["tail_stages_ms"]
    print(f"{sys.argv[2]:22s} {d['value']:8.1f} pages/s {d['ms_per_step']:7.3f} ms | net ms {(d.get('roofline') or {}).get('net_ms_per_step')} | merge wait {s.get('refine_wait_merge')} | {sys.argv[3]}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2 3; do
run base_$rep ""
run c12_40_$rep "tail_lds_cls0=12288,tail_lds_cls1=40960"
run c16_40_$rep "tail_lds_cls0=16384,tail_lds_cls1=40960"
run c12_40_r15_$rep "tail_lds_cls0=12288,tail_lds_cls1=40960,tail_lds_runs_x10=15"
run c10_24_r10_$rep "tail_lds_cls0=10240,tail_lds_cls1=24576,tail_lds_runs_x10=10"
done
