#!/bin/bash
# round 3, final measurements at HEAD: GPU suite, smoke, the default bench line (driver arguments), network-only and fp32s
# lines with per-op tables, rocprofv3 kernel stats of the three, selftests (split kernels, co-run), PMC traffic of the fp32s forward
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT || exit 1
O=$ROOT/gpurun_out/r03final
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt | cut -c1-300
( time timeout 1200 python bench.py --steps 20 --warmup 5 --dump-ops $O/per_op_fp16.tsv ) > $O/bench_n1.json 2> $O/bench_n1.err
python -c "
import json; d=json.load(open('$O/bench_n1.json')); print('default', d['value'], d['ms_per_step'], 'exact', d['parity_exact'], {k:(v.get('value'), v.get('error')) for k,v in d['extra_configs'].items()})" | cut -c1-800
tail -3 $O/bench_n1.err | cut -c1-200
timeout 200 python bench.py --mode net --steps 40 --warmup 5 --spinup 60 --no-cpu-baseline --no-extras > $O/bench_net.json 2> $O/bench_net.err
timeout 300 python bench.py --precision fp32s --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras > $O/bench_fp32s_e2e.json 2> $O/bench_fp32s_e2e.err
timeout 300 python bench.py --precision fp32s --mode net --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras --dump-ops $O/per_op_fp32s.tsv > $O/bench_fp32s_net.json 2> $O/bench_fp32s_net.err
for f in bench_net bench_fp32s_e2e bench_fp32s_net; do python -c "import json;d=json.load(open('$O/$f.json'));print('$f',d['value'],d['ms_per_step'])"; done
( cd comic-text-detector_amd && ST_SPLIT=1 timeout 500 ./ctd_selftest 32 ) > $O/split_selftest_b32.txt 2>&1; tail -1 $O/split_selftest_b32.txt
( cd comic-text-detector_amd && ST_CORUN=1 timeout 300 ./ctd_selftest 32 ) > $O/corun_selftest.txt 2>&1; tail -1 $O/corun_selftest.txt
cd /tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python $ROOT/bench.py --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extras > $O/prof_e2e.log 2>&1; echo e2e rc=$?
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_net -o net -- python $ROOT/bench.py --mode net --steps 10 --warmup 2 --spinup 0 --no-cpu-baseline --no-extras > $O/prof_net.log 2>&1; echo net rc=$?
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32s -o fp32s -- python $ROOT/bench.py --precision fp32s --steps 6 --warmup 2 --spinup 0 --no-cpu-baseline --no-extras > $O/prof_fp32s.log 2>&1; echo fp32s rc=$?
find $O -name "*kernel_trace.csv" -delete
cd $ROOT
bash scripts/gpu_traffic.sh fp32s 2>&1 | tail -3 | cut -c1-300
cp -r gpurun_out/traffic_fp32s $O/ 2>/dev/null
ls $O | head -40
