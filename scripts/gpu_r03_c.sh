#!/bin/bash
# round 3, GPU call C: the tail with one-launch copies / fills, fp32 SPPF fusion, the stem on the split kernel (selftest + the
# whole GPU suite), the band dump, the default bench line on the reference-density workload + rocprof of it
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c
mkdir -p $O
( cd comic-text-detector_amd && ST_SPLIT=1 ST_CASES=99 timeout 300 ./ctd_selftest 8 ) > $O/split_selftest_stem.txt 2>&1
tail -4 $O/split_selftest_stem.txt | cut -c1-400
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
timeout 300 python scripts/gpu_band_dump.py > $O/band_dump.txt 2>&1
tail -3 $O/band_dump.txt | cut -c1-300
( time timeout 900 python bench.py --steps 20 --warmup 5 --rocm-timeout 60 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json; tail -4 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_e2e -o e2e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --spinup 20 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_prof.json 2> $GRAFT_REPO_ROOT/$O/bench_prof.err
cd $GRAFT_REPO_ROOT
find $O/prof_e2e -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_e2e.csv
rm -rf $O/prof_e2e
head -30 $O/rocprofv3_kernel_stats_e2e.csv | cut -c1-160
