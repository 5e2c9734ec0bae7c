#!/bin/bash
# Instruction-mix PMC pass on selftest cases (dynamic instruction counts per wave).
# usage: CASES="0 9" B=32 ST_VAR=1 bash scripts/gpu_pmc_inst.sh
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUTBASE=$ROOT/gpurun_out/pmc_inst
mkdir -p $OUTBASE
cd /tmp
export ST_VAR=${ST_VAR:-1}
for CASE in ${CASES:-"0"}; do
  export ST_CASES=$CASE
  OUT=$OUTBASE/case$CASE
  mkdir -p $OUT
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES \
      --kernel-trace --output-format csv -d $OUT/i1 -o i1 -- $ROOT/comic-text-detector_amd/ctd_selftest ${B:-32} > $OUT/i1.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $OUT/i2 -o i2 -- $ROOT/comic-text-detector_amd/ctd_selftest ${B:-32} > $OUT/i2.log 2>&1
  python3 $ROOT/scripts/pmc_summary.py $OUT | grep -A16 "conv_igemm"
done
