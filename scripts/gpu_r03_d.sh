#!/bin/bash
# round 3, GPU call D: whole GPU suite (tail with one-launch copies, fp32 SPPF, stem split, band test), split selftest with
# the 256-pixel variant, default bench + rocprofv3 kernel stats (csv) of the end-to-end and fp32s runs
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03d
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -12 $O/pytest_gpu.txt | cut -c1-300
( cd comic-text-detector_amd && ST_SPLIT=1 ST_CASES=0,1,8,12,13,18 timeout 300 ./ctd_selftest 8 ) > $O/split_selftest_bm256.txt 2>&1
grep -E "N64|3x3s2 32->64|convT4 128|db tail|32->32|PASSED|FAILED" $O/split_selftest_bm256.txt | cut -c1-900
( time timeout 900 python bench.py --steps 20 --warmup 5 --rocm-timeout 60 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json; tail -4 $O/bench_default.err
timeout 300 python bench.py --precision fp32s --steps 16 --warmup 4 --spinup 20 --no-cpu-baseline --no-extras --dump-ops $O/per_op_fp32s.tsv > $O/bench_fp32s.json 2> $O/bench_fp32s.err
tail -c 300 $O/bench_fp32s.json
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_e2e -o e2e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_e2e.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fp32s -o fp32s -- python $GRAFT_REPO_ROOT/bench.py --precision fp32s --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_fp32s.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_e2e $O/prof_fp32s -name "*kernel_trace.csv" -delete
find $O/prof_e2e -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_e2e.csv
find $O/prof_fp32s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_fp32s_e2e.csv
rm -rf $O/prof_e2e $O/prof_fp32s
head -40 $O/rocprofv3_kernel_stats_e2e.csv | cut -c1-170
