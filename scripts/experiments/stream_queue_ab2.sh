run() { echo "$1 | $2: $(CTD_TUNING=$1 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
run tail_priority=1 "--workers 3"
run tail_priority=1 "--workers 4"
run tail_priority=1 "--workers 5"
run tail_priority=1 "--workers 4 --tail-split 4 --depth 5"
run tail_priority=1 "--workers 3 --dense-blocks"
run tail_priority=1 "--workers 4 --dense-blocks"
run tail_priority=1 "--workers 5 --dense-blocks"
run tail_priority=1 "--workers 4 --host-input"
run tail_priority=1 "--workers 3 --host-input"
run tail_priority=2 "--workers 4"
CTD_TUNING=tail_priority=1 python scripts/gpu_inprocess.py 30 2>&1 | grep -v amdgpu | head -4
