#!/bin/bash
# One parameterised GPU-box session (replaces round 3's 38 single-use gpu_r03_*.sh; those are in the history up to
# commit 8daff86).  Run through gpurun from the repo root:
#
#   gpurun --timeout 900 -- 'bash scripts/gpu.sh <tag> <step> [<step> ...]'
#
# Outputs go to gpurun_out/<tag>/ (scratch; copy what is worth keeping into profiles/).  Steps, run in the order given:
#   suite[:<pytest -k expression>]      pytest -m gpu (the whole GPU suite, or a selection)
#   smoke                               __graft_entry__.smoke()
#   bench[:<name>[:<bench.py args>]]    the driver's line (--steps 20 --warmup 5) + per-op table; args are appended
#   run:<name>:<bench.py args>          a sub-run line: bench.py --no-cpu-baseline --no-extras <args>; prints value / ms
#   st:<name>:<batch>:<ENV=V,ENV=V>     ctd_selftest <batch> with the environment given (ST_CASES=..,ST_SPLIT=1,CTD_TUNING=a=1;b=2)
#   prof:<name>:<bench.py args>         rocprofv3 --kernel-trace --stats of a sub-run (kernel stats CSV kept, trace dropped)
#   trace:<name>:<bench.py args>        the same, keeping the kernel trace slimmed by scripts/timeline_summary.py
#   pmc:<cases>[:<batch>[:sq1]]         scripts/gpu_pmc.sh on selftest cases (own passes, --kernel-trace only)
#   traffic:<precision>                 scripts/gpu_traffic.sh (FETCH_SIZE / WRITE_SIZE passes of the network)
#   py:<name>:<script and args>         python <script and args>
#   sh:<name>:<command>                 bash -c <command>
# A `,` inside an ENV list separates variables; use `;` inside a CTD_TUNING value (turned into `,`).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
TAG=${1:?tag}; shift
O=$ROOT/gpurun_out/$TAG
mkdir -p "$O"
show() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    c = d.get("config", {})
    print("  value", d.get("value"), d.get("unit"), "| ms/step", d.get("ms_per_step"), "| blocks/lines per page", c.get("blocks_per_page"), c.get("lines_per_page"),
          "| net ms", (d.get("roofline") or {}).get("net_ms_per_step"), "| backbone frac", ((d.get("roofline") or {}).get("backbone") or {}).get("hbm_frac"),
          "| deliveries", c.get("result_delivery_intervals"))
    s = d.get("serial_step")
    if s: print("  serial: forward", s["forward_ms"], "tail", s["tail_ms"], s.get("tail_stages_ms"))
    for k in ("parity_exact",):
        if d.get(k): print(" ", k, d[k].get("value"), d[k].get("ms_per_step"))
    for k, v in (d.get("extra_configs") or {}).items(): print("  extra", k, v.get("value"), v.get("ms_per_step"), v.get("error"))
    if d.get("cpu_baseline"): print("  cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("sample", "")[:200])
except Exception as e:
    print("  (no JSON line:", e, ")")
PY
}
for STEP in "$@"; do
  KIND=${STEP%%:*}; REST=${STEP#*:}; [ "$REST" = "$STEP" ] && REST=""
  NAME=${REST%%:*}; ARGS=${REST#*:}; [ "$ARGS" = "$REST" ] && ARGS=""
  echo "== $STEP"
  case $KIND in
    suite) if [ -n "$REST" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$REST" > "$O/pytest_gpu.txt" 2>&1; else timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.txt" 2>&1; fi
           tail -4 "$O/pytest_gpu.txt" | cut -c1-300 ;;
    smoke) timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1; tail -3 "$O/smoke.txt" | cut -c1-300 ;;
    bench) N=${NAME:-bench_n1}; ( time timeout 1500 python bench.py --steps 20 --warmup 5 --dump-ops "$O/$N.per_op.tsv" $ARGS ) > "$O/$N.json" 2> "$O/$N.err"
           show "$O/$N.json"; tail -4 "$O/$N.err" | cut -c1-200 ;;
    run)   timeout 600 python bench.py --no-cpu-baseline --no-extras $ARGS > "$O/$NAME.json" 2> "$O/$NAME.err"; show "$O/$NAME.json"; tail -2 "$O/$NAME.err" | cut -c1-200 ;;
    st)    B=${ARGS%%:*}; ENVS=${ARGS#*:}; [ "$ENVS" = "$ARGS" ] && ENVS=""
           ( cd comic-text-detector_amd && env $(echo "$ENVS" | tr ',' ' ' | tr ';' ',') timeout 600 ./ctd_selftest "$B" ) > "$O/$NAME.txt" 2>&1; tail -${ST_TAIL:-25} "$O/$NAME.txt" | cut -c1-260 ;;
    prof|trace)
           MC=""; [ "$KIND" = trace ] && MC="--memory-copy-trace"
           ( cd /tmp && timeout 300 rocprofv3 --kernel-trace $MC --stats --output-format csv -d "$O/prof_$NAME" -o "$NAME" -- python "$ROOT/bench.py" --no-cpu-baseline --no-extras --spinup 0 $ARGS > "$O/prof_$NAME.log" 2>&1; echo "  rc=$?" )
           if [ "$KIND" = trace ]; then for f in $(find "$O/prof_$NAME" -name "*kernel_trace.csv"); do python scripts/timeline_summary.py "$f" --slim "$O/trace_$NAME.csv.gz" > "$O/timeline_$NAME.txt" 2>&1; head -60 "$O/timeline_$NAME.txt" | cut -c1-200; done; fi
           for f in $(find "$O/prof_$NAME" -name "*memory_copy_trace.csv"); do gzip -c "$f" > "$O/memcpy_$NAME.csv.gz"; python scripts/memcpy_summary.py "$f" | head -20; done
           find "$O/prof_$NAME" -name "*_trace.csv" -delete
           for f in $(find "$O/prof_$NAME" -name "*kernel_stats.csv"); do cp "$f" "$O/kernel_stats_$NAME.csv"; head -14 "$f" | cut -c1-200; done ;;
    pmc)   C=$NAME; B=${ARGS%%:*}; M=${ARGS#*:}; [ "$M" = sq1 ] && export PMC_ONLY_SQ1=1
           CASES="$(echo $C | tr ',' ' ')" ST_BATCH=${B:-32} bash scripts/gpu_pmc.sh 2>&1 | tail -60 | cut -c1-220; cp -r gpurun_out/pmc "$O/" 2>/dev/null ;;
    traffic) bash scripts/gpu_traffic.sh $REST 2>&1 | tail -4 | cut -c1-300; cp -r gpurun_out/traffic* "$O/" 2>/dev/null ;;
    py)    timeout 900 python $ARGS > "$O/$NAME.txt" 2>&1; tail -${PY_TAIL:-30} "$O/$NAME.txt" | cut -c1-300 ;;
    sh)    timeout 900 bash -c "$ARGS" > "$O/$NAME.txt" 2>&1; tail -30 "$O/$NAME.txt" | cut -c1-300 ;;
    *) echo "unknown step $STEP" ;;
  esac
done
ls "$O" | head -60
