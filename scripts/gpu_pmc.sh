#!/bin/bash
# PMC counter passes on the standalone selftest (no torch -> fast).  Counters are
# collected in their own passes with --kernel-trace only (never with sys/hip traces).
# usage: CASES="16 0 11" bash scripts/gpu_pmc.sh      (selftest case indices, one dir each)
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUTBASE=$ROOT/gpurun_out/pmc
mkdir -p $OUTBASE
cd /tmp
rocprofv3 -L > $OUTBASE/counters_list.txt 2>&1
export ST_VAR=${ST_VAR:-0}
CASES=${CASES:-"16 0 11"}

pass() {  # name, counters...
  name=$1; shift
  have=""
  for c in "$@"; do
    if grep -qw "$c" $OUTBASE/counters_list.txt; then have="$have $c"; else echo "[pmc] counter $c not available"; fi
  done
  [ -z "$have" ] && return
  echo "== case $ST_CASES pass $name:$have"
  timeout 300 rocprofv3 --pmc $have --kernel-trace --output-format csv -d $OUT/$name -o $name -- \
      $ROOT/comic-text-detector_amd/ctd_selftest ${ST_BATCH:-8} > $OUT/$name.log 2>&1
  echo "rc=$?"
}

for CASE in $CASES; do
  export ST_CASES=$CASE
  export OUT=$OUTBASE/case$CASE
  mkdir -p $OUT
  pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
  if [ -n "$PMC_ONLY_SQ1" ]; then python3 $ROOT/scripts/pmc_summary.py $OUT | head -40; continue; fi
  pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
  pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
  pass ta1 TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
  python3 $ROOT/scripts/pmc_summary.py $OUT | head -80
done
