#!/bin/bash
# Round-2 measurement session: every number quoted in DESIGN.md / profiles/ comes from this script.
# Outputs under gpurun_out/r02/ (copied into profiles/ by hand after review).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r02
mkdir -p $O
cd $ROOT
echo "== bench e2e (headline)";  timeout 600 python bench.py --steps 20 --warmup 3 --dump-ops $O/bench_per_op.tsv > $O/bench_n1.json 2> $O/bench_n1.err; echo rc=$?; cut -c1-200 $O/bench_n1.json
echo "== bench net";             timeout 300 python bench.py --mode net --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_net.json 2>/dev/null; cut -c1-160 $O/bench_net.json
echo "== bench fp32 bs8 net";    timeout 300 python bench.py --precision fp32 --batch 8 --mode net --steps 10 --warmup 2 --no-cpu-baseline --dump-ops $O/bench_fp32_per_op.tsv > $O/bench_fp32_bs8_net.json 2>/dev/null; cut -c1-160 $O/bench_fp32_bs8_net.json
echo "== bench fp32 bs8 e2e";    timeout 300 python bench.py --precision fp32 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fp32_bs8_e2e.json 2>/dev/null; cut -c1-160 $O/bench_fp32_bs8_e2e.json
echo "== bench mixed";           timeout 300 python bench.py --mode mixed --steps 5 --warmup 2 > $O/bench_mixed.json 2>/dev/null; cut -c1-160 $O/bench_mixed.json
echo "== bench e2e keep_undetected"; timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --keep-undetected > $O/bench_e2e_keep.json 2>/dev/null; cut -c1-160 $O/bench_e2e_keep.json
echo "== acceptance";            timeout 600 python -m pytest tests/test_gpu_accept.py -m gpu -q -s 2>&1 | grep -E "acceptance|passed|failed" > $O/acceptance.txt; tail -1 $O/acceptance.txt
echo "== extra";                 timeout 300 python scripts/gpu_extra.py > $O/supplementary.json 2>/dev/null; cat $O/supplementary.json
echo "== tail alone";            GC_FREEZE=1 TAIL_ITERS=10 python scripts/gpu_tail_prof.py 2>/dev/null | tail -1 > $O/tail_alone.json; cat $O/tail_alone.json
cd /tmp
echo "== rocprofv3 e2e";  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_e2e.log 2>&1; echo rc=$?
echo "== rocprofv3 net";  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_net -o net -- python $ROOT/bench.py --mode net --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_net.log 2>&1; echo rc=$?
echo "== rocprofv3 tail"; GC_FREEZE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tail -o tail -- python $ROOT/scripts/gpu_tail_prof.py > $O/rocprof_tail.log 2>&1; echo rc=$?
echo "== rocprofv3 fp32"; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -o fp32 -- python $ROOT/bench.py --precision fp32 --batch 8 --mode net --steps 5 --warmup 1 --no-cpu-baseline > $O/rocprof_fp32.log 2>&1; echo rc=$?
rm -f $O/prof_*/*kernel_trace.csv   # traces are large; the stats are what is kept
ls $O
