#!/bin/bash
# round 3, GPU call F: the measurement set the docs cite (GPU suite, default bench line, rocprofv3 kernel stats of the end-to-end /
# network-only / fp32s runs, PMC traffic of both engines' conv families, the split kernel's selftest, stem A/B)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -4 $O/pytest_gpu.txt
( cd comic-text-detector_amd && ST_SPLIT=1 timeout 400 ./ctd_selftest 8 ) > $O/split_selftest_b8.txt 2>&1
tail -2 $O/split_selftest_b8.txt | cut -c1-200
( cd comic-text-detector_amd && ST_ONLY_C3=1 timeout 400 ./ctd_selftest 32 ) > $O/selftest_fused_b32.txt 2>&1
grep -E "stem2|selftest" $O/selftest_fused_b32.txt | cut -c1-260
( time timeout 900 python bench.py --steps 20 --warmup 5 --rocm-timeout 70 --dump-ops $O/per_op_fp16.tsv ) > $O/bench_n1.json 2> $O/bench_n1.err
tail -c 300 $O/bench_n1.json; tail -4 $O/bench_n1.err
timeout 300 python bench.py --precision fp32s --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras --dump-ops $O/per_op_fp32s.tsv > $O/bench_fp32s_e2e.json 2> $O/bench_fp32s_e2e.err
timeout 300 python bench.py --mode net --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_net.json 2> $O/bench_net.err
timeout 300 python bench.py --host-input --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_host_input.json 2> $O/bench_host_input.err
timeout 300 python bench.py --keep-undetected --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_keep_undetected.json 2> $O/bench_keep_undetected.err
timeout 300 python bench.py --mode mixed --steps 3 --warmup 1 > $O/bench_mixed.json 2> $O/bench_mixed.err
for f in bench_fp32s_e2e bench_net bench_host_input bench_keep_undetected bench_mixed; do python -c "import json;d=json.load(open('$O/$f.json'));print('$f',d['value'],d['ms_per_step'])"; done
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_e2e -o e2e -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_e2e.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_net -o net -- python $GRAFT_REPO_ROOT/bench.py --mode net --steps 20 --warmup 5 --spinup 0 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_net.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fp32s -o fp32s -- python $GRAFT_REPO_ROOT/bench.py --precision fp32s --mode net --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/prof_fp32s.log 2>&1
cd $GRAFT_REPO_ROOT
for n in e2e net fp32s; do
  find $O/prof_$n -name "*kernel_trace.csv" -delete
  find $O/prof_$n -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/rocprofv3_kernel_stats_$n.csv
  rm -rf $O/prof_$n
done
bash scripts/gpu_traffic.sh fp16 > $O/traffic_fp16.txt 2>&1; cp gpurun_out/traffic_fp16/traffic.json $O/traffic_pmc_fp16.json
bash scripts/gpu_traffic.sh fp32s > $O/traffic_fp32s.txt 2>&1; cp gpurun_out/traffic_fp32s/traffic.json $O/traffic_pmc_fp32s.json
rm -rf gpurun_out/traffic_fp16 gpurun_out/traffic_fp32s
cat $O/traffic_pmc_fp16.json $O/traffic_pmc_fp32s.json
