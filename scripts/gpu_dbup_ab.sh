#!/bin/bash
# The head tails on the MFMA (db_up_mfma_kernel, seg_final_mfma_kernel): selftest against the VALU kernels, the GPU
# suite with them, bench A/B on one box (both on / both off, twice).
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/dbup
mkdir -p $O
cd $ROOT
ST_ONLY_C3=1 timeout 120 ./comic-text-detector_amd/ctd_selftest 32 > $O/selftest.txt 2>&1; grep -E "dbup|segfinal|selftest:" $O/selftest.txt | cut -c1-250
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt | cut -c1-200
for M in 1 0 1 0; do
  CTD_DBUP_MFMA=$M CTD_SEGFINAL_MFMA=$M timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dump-ops $O/ops_m$M.tsv > $O/bench_m$M.json 2>/dev/null
  echo "head tails on the MFMA=$M: $(python3 -c "import json;d=json.load(open('$O/bench_m$M.json'));print(d['value'], d['ms_per_step'], d['roofline']['net_ms_per_step'])") db.up $(grep -P '^db.up\t' $O/ops_m$M.tsv | cut -f3) seg.upconv6 $(grep -P '^seg.upconv6\t' $O/ops_m$M.tsv | cut -f3)"
done
