#!/bin/bash
# Two SQ counter passes (MFMA busy / waits; LDS conflicts / VMEM) over a few selftest cases in ONE process per pass, raw CSVs
# deleted after the per-kernel summary (a full gpu_pmc.sh session writes > 64 MB).  Counters in their own rocprofv3 runs with
# --kernel-trace only.   usage: CASES=16,17,18 [ST_SPLIT=1] [BATCH=32] bash scripts/gpu_pmc_mini.sh <outname>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-pmc_mini}
mkdir -p "$O"
export ST_CASES=${CASES:-16,17,18} ST_VAR=0 ST_NO_C3=1
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$O/$n" -o "$n" -- "$ROOT/comic-text-detector_amd/ctd_selftest" ${BATCH:-32} > "$O/$n.log" 2>&1; echo "$n rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 "$ROOT/scripts/pmc_summary.py" "$O" > "$O/summary_all.txt" 2>&1
grep -A22 "conv_halo3_kernel\|conv_halo2_kernel\|conv_split_halo_kernel\|conv_halo_kernel<" "$O/summary_all.txt" > "$O/summary.txt"
grep "\[case\]" "$O/sq1.log" | cut -c1-200 >> "$O/summary.txt"
rm -rf "$O/sq1" "$O/sq2" "$O/grbm" "$O/summary_all.txt"
cat "$O/summary.txt" | cut -c1-160 | head -150
