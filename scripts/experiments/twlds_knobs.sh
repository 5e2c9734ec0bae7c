#!/bin/bash
# Round 6: knobs of the window-local merge kernel (kernels_twlds.hip) on the end-to-end step, one box, interleaved:
# threads per window, run-table size (LDS footprint), launch classes.  Usage: gpurun -- 'bash scripts/experiments/twlds_knobs.sh <out>'
export TMPDIR=/tmp
O=${1:-gpurun_out/twlds_knobs}; mkdir -p "$O"
run() {  # name tuning args
  CTD_TUNING="$2" timeout 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-60} --warmup 5 $3 > "$O/$1.json" 2> "$O/$1.err"
  python - "$O/$1.json" "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    s = d["serial_step"]["tail_stages_ms"]
    print(f"{sys.argv[2]:28s} {d['value']:8.1f} pages/s {d['ms_per_step']:7.3f} ms | serial tail {d['serial_step']['tail_ms']:6.2f} merge wait {s.get('refine_wait_merge')} | {sys.argv[3]}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
run base_$rep ""
run t512_$rep "tail_lds_threads=512"
run t1024_$rep "tail_lds_threads=1024"
run r15_$rep "tail_lds_runs_x10=15"
run r15_t512_$rep "tail_lds_runs_x10=15,tail_lds_threads=512"
run onecls_$rep "tail_lds_cls0=153600,tail_lds_cls1=153600"
run cls24_48_$rep "tail_lds_cls0=24576,tail_lds_cls1=49152"
run canvas_$rep "tail_lds=0"
done
run dense_base "" "--dense-blocks"
run dense_t512 "tail_lds_threads=512" "--dense-blocks"
run dense_t1024 "tail_lds_threads=1024" "--dense-blocks"
run dense_r15_t512 "tail_lds_runs_x10=15,tail_lds_threads=512" "--dense-blocks"
run dense_canvas "tail_lds=0" "--dense-blocks"
