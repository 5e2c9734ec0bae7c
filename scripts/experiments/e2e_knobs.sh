run() { echo "$1: $(CTD_TUNING=$1 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; }
run tail_chain=1
run tail_max_blocks=512
run tail_max_blocks=256
run tail_priority=2
run tail_priority=1
run tail_chain=0
run tail_chain=2
run tail_chain=1 "--workers 2"
run tail_chain=1 "--depth 6"
run halo2=0
run tail_chain=1
