#!/bin/bash
# Round-2, session 2: the multi-layer kernels (stem + layer 1, 32-channel C3 block, SPPF pools).
# One box visit, most important evidence first; every step has its own timeout.  Outputs: gpurun_out/r03/.
#   1 selftest of the fused kernels against the launches they replace (bit identity + timing)
#   2 the GPU bit-identity test of the whole network with / without them
#   -> if either fails the rest of the session runs (and reports) the layer-per-launch program: CTD_FUSE=0
#   3 the full GPU test suite   4 the headline bench (+ per-op table)   5 the same with CTD_FUSE=0 (A/B, same box)
#   6 rocprofv3 kernel stats of the bench   7 HBM traffic (PMC)   8 net-mode bench + stats   9 extras
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r03
mkdir -p $O
cd $ROOT
T0=$(date +%s)
el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
FUSE_OK=1
el "== 1 selftest (fused kernels, B=8)"
ST_C3_DBG=1 ST_ONLY_C3=1 timeout 300 ./comic-text-detector_amd/ctd_selftest 8 > $O/selftest_fused_b8.txt 2>&1; echo rc=$?
cat $O/selftest_fused_b8.txt | cut -c1-260
grep -q "selftest: PASSED" $O/selftest_fused_b8.txt || FUSE_OK=0
el "== 2 pytest fused bit-identity"
timeout 900 python -m pytest tests/test_gpu_edge.py -m gpu -q -x -k "fused" > $O/pytest_fused.txt 2>&1 || FUSE_OK=0
tail -15 $O/pytest_fused.txt | cut -c1-300
echo "FUSE_OK=$FUSE_OK" | tee $O/fuse_ok.txt
if [ $FUSE_OK = 0 ]; then export CTD_FUSE=0; fi
el "== 3 full GPU suite"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo rc=$?; tail -5 $O/pytest_gpu.txt | cut -c1-300
el "== 4 bench e2e (headline)"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --dump-ops $O/bench_per_op.tsv > $O/bench_n1.json 2> $O/bench_n1.err; echo rc=$?; cut -c1-220 $O/bench_n1.json
if [ $FUSE_OK = 1 ]; then
  el "== 5 bench e2e, CTD_FUSE=0 (A/B on the same box)"
  CTD_FUSE=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dump-ops $O/bench_per_op_fuse0.tsv > $O/bench_n1_fuse0.json 2>/dev/null; cut -c1-220 $O/bench_n1_fuse0.json
fi
cd /tmp
el "== 6 rocprofv3 e2e"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_e2e.log 2>&1; echo rc=$?
rm -f $O/prof_*/*kernel_trace.csv $O/prof_*/*/*kernel_trace.csv
el "== 7 HBM traffic (PMC, separate passes)"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/traffic/$C -o $C -- \
     python $ROOT/bench.py --mode net --steps 1 --warmup 1 --spinup 0 --no-cpu-baseline > $O/traffic_$C.log 2>&1
  echo "$C rc=$?"
done
python3 $ROOT/scripts/traffic_summary.py $O/traffic > $O/traffic_summary.txt 2>&1; tail -3 $O/traffic_summary.txt
rm -f $O/traffic/*/*kernel_trace.csv $O/traffic/*/*/*kernel_trace.csv
cd $ROOT
el "== 8 bench net + rocprofv3 net"
timeout 300 python bench.py --mode net --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_net.json 2>/dev/null; cut -c1-160 $O/bench_net.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_net -o net -- python $ROOT/bench.py --mode net --steps 10 --warmup 2 --no-cpu-baseline > $O/rocprof_net.log 2>&1; echo rc=$?
rm -f $O/prof_*/*kernel_trace.csv $O/prof_*/*/*kernel_trace.csv
cd $ROOT
el "== 9 extras: selftest B=32 timing, smoke, mixed, fp32"
ST_ONLY_C3=1 timeout 200 ./comic-text-detector_amd/ctd_selftest 32 > $O/selftest_fused_b32.txt 2>&1; grep -E "^\[c3\]|^\[stem2\]|^\[sppf\]|selftest:" $O/selftest_fused_b32.txt | cut -c1-260
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo smoke rc=$?; tail -3 $O/smoke.txt | cut -c1-200
timeout 300 python bench.py --mode mixed --steps 5 --warmup 2 > $O/bench_mixed.json 2>/dev/null; cut -c1-160 $O/bench_mixed.json
timeout 300 python bench.py --precision fp32 --batch 8 --mode net --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fp32_bs8_net.json 2>/dev/null; cut -c1-160 $O/bench_fp32_bs8_net.json
timeout 300 python bench.py --precision fp32 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fp32_bs8_e2e.json 2>/dev/null; cut -c1-160 $O/bench_fp32_bs8_e2e.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --keep-undetected > $O/bench_e2e_keep.json 2>/dev/null; cut -c1-160 $O/bench_e2e_keep.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-input > $O/bench_e2e_host.json 2>/dev/null; cut -c1-160 $O/bench_e2e_host.json
timeout 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > $O/bench_e2e_long.json 2>/dev/null; cut -c1-160 $O/bench_e2e_long.json
el "done"; ls $O
