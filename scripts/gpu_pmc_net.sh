#!/bin/bash
# SQ counter passes (MFMA busy / waits; LDS conflicts / VMEM; clock) over the NETWORK's kernels at the benchmark shape: three
# rocprofv3 --pmc runs of `bench.py --mode net --steps 2` (counters in their own runs, --kernel-trace only), per-kernel
# summary of the kernels matching a pattern, raw CSVs deleted.
#   usage: [CTD_TUNING=key=value,...] bash scripts/gpu_pmc_net.sh <outname> '<grep -E pattern of kernel names>'
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-pmc_net}
PAT=${2:-c3b_kernel}
mkdir -p "$O"
cd /tmp
run() { n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$O/$n" -o "$n" -- python "$ROOT/bench.py" --mode net --steps 2 --warmup 1 --spinup 0 --no-cpu-baseline --no-extras > "$O/$n.log" 2>&1; echo "$n rc=$?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM
run sq3 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 "$ROOT/scripts/pmc_summary.py" "$O" > "$O/summary_all.txt" 2>&1
grep -E -A34 "$PAT" "$O/summary_all.txt" > "$O/summary.txt"
rm -rf "$O/sq1" "$O/sq2" "$O/sq3" "$O/grbm" "$O/summary_all.txt"
cut -c1-170 "$O/summary.txt" | head -${PMC_HEAD:-160}
