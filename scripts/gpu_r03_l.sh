#!/bin/bash
# round 3, GPU call L: split kernel with all fragment reads of a K step issued before its MFMAs (selftest B=32 + engine forward)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03l
mkdir -p $O
( cd comic-text-detector_amd && ST_SPLIT=1 ST_CASES=3,10,11,16,17,18 timeout 400 ./ctd_selftest 32 ) > $O/split_selftest_b32.txt 2>&1
grep -E "^\[split\]|selftest" $O/split_selftest_b32.txt | sed 's/err vs f64 rms [0-9.e+-]* max [0-9.e+-]*,//g; s/max|d| vs f32-MFMA [0-9.e+-]* ([0-9]* > 2e-5),//g' | cut -c1-330
for w in 1 0; do
CTD_TUNING=split_wdma=$w timeout 300 python bench.py --precision fp32s --mode net --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras --dump-ops $O/per_op_fp32s_wdma$w.tsv > $O/bench_fp32s_net_wdma$w.json 2> $O/bench_fp32s_net_wdma$w.err
python -c "import json;d=json.load(open('$O/bench_fp32s_net_wdma$w.json'));print('wdma=$w net',d['value'],d['ms_per_step'],d['roofline']['net_ms_per_step'])"
done
