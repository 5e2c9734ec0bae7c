#!/bin/bash
# round 3, GPU call A: the split-operand kernel (selftest), the fp32s engine's parity tests, a first forward time
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
( cd comic-text-detector_amd && ST_SPLIT=1 timeout 300 ./ctd_selftest 8 ) > $O/split_selftest_b8.txt 2>&1
tail -5 $O/split_selftest_b8.txt
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_accept.py -m gpu -x -q -s > $O/pytest_net_accept.txt 2>&1
tail -15 $O/pytest_net_accept.txt
for prec in fp32s fp32; do
  timeout 300 python bench.py --precision $prec --mode net --batch 32 --steps 10 --warmup 3 --spinup 10 --no-cpu-baseline \
      --dump-ops $O/per_op_${prec}_b32.tsv > $O/bench_net_${prec}_b32.json 2> $O/bench_net_${prec}_b32.err
  tail -c 600 $O/bench_net_${prec}_b32.json
done
timeout 300 python bench.py --precision fp32s --mode e2e --batch 32 --steps 10 --warmup 3 --spinup 20 --no-cpu-baseline > $O/bench_e2e_fp32s_b32.json 2> $O/bench_e2e_fp32s_b32.err
tail -c 400 $O/bench_e2e_fp32s_b32.json
