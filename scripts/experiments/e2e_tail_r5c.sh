#!/bin/bash
# Round 5, after the tail kernels' load batching: the block cap of the tail's big-grid kernels, the worker count and a
# variant library (TWB_U = 8: eight groups per thread in flight in the per-window kernels) on the headline pages and the
# dense ones, one box, 40 timed steps each.  Each line = pages/s, ms per step, cores.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$CTD_TUNING $*: $(python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
run
CTD_TUNING=tail_max_blocks=512 run
CTD_TUNING=tail_max_blocks=768 run
CTD_TUNING=tail_max_blocks=2048 run
run --workers 3
run --dense-blocks
CTD_TUNING=tail_max_blocks=512 run --dense-blocks
CTD_TUNING=tail_max_blocks=2048 run --dense-blocks
if [ -f comic-text-detector_amd/variants/libctd_hip_u8.so ]; then
  cp comic-text-detector_amd/libctd_hip.so /tmp/libctd_hip.keep && cp comic-text-detector_amd/variants/libctd_hip_u8.so comic-text-detector_amd/libctd_hip.so
  echo "-- TWB_U = 8"
  run
  run --dense-blocks
  cp /tmp/libctd_hip.keep comic-text-detector_amd/libctd_hip.so
fi
run
