#!/bin/bash
# One GPU-box session: kernel self-test, GPU parity tests, smoke, bench, rocprof.
# Everything is wrapped in `timeout` so a hung kernel cannot eat the box budget.
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== selftest"; timeout 300 ./comic-text-detector_amd/ctd_selftest ${SELFTEST_B:-8} > gpurun_out/selftest.log 2>&1; echo "rc=$?"; tail -30 gpurun_out/selftest.log
if [ -z "$SKIP_PYTEST" ]; then
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -rA --tb=short ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; echo "rc=$?"; grep -E "PASSED|FAILED|ERROR|passed|failed|error|fp16 vs" gpurun_out/pytest_gpu.log | tail -60
fi
if [ -z "$SKIP_SMOKE" ]; then
echo "== smoke"; timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/smoke.log
fi
if [ -z "$SKIP_BENCH" ]; then
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 --dump-ops gpurun_out/ops.tsv ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
sort -t$'\t' -k3 -g -r gpurun_out/ops.tsv | head -25
fi
if [ -n "$DO_ROCPROF" ]; then
echo "== rocprofv3"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rc=$?"; cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/rocprof.log; find gpurun_out/prof -name "*stats*" | head
fi
