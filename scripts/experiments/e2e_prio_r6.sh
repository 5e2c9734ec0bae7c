export TMPDIR=/tmp
O=gpurun_out/c12; mkdir -p $O
run() { CTD_TUNING="$2" timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 $3 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json "$1" "$2 $3" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(f"{sys.argv[2]:22s} {d['value']:8.1f} pages/s {d['ms_per_step']:7.3f} ms | {sys.argv[3]}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
run base_$rep "" ""
run fwdhigh_$rep "" "--fwd-stream high"
run taillow_$rep "tail_priority=2" ""
run fwdhigh_taillow_$rep "tail_priority=2" "--fwd-stream high"
done
