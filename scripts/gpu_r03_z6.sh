#!/bin/bash
# round 3, GPU call Z6: 64-channel 3x3 layers of the fp32s engine on 16x8 haloed patches (two blocks per CU): selftest, parity tests, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z6
mkdir -p $O
( cd comic-text-detector_amd && ST_SPLIT=1 ST_CASES=9,10,14,18 timeout 300 ./ctd_selftest 32 ) > $O/split_selftest.txt 2>&1
grep -E "^\[split\]|selftest" $O/split_selftest.txt | sed 's/err vs f64 rms [0-9.e+-]* max [0-9.e+-]*,//g; s/max|d| vs f32-MFMA [0-9.e+-]* ([0-9]* > 2e-5),//g; s/f32-MFMA:[^|]*|//; s/split(reg):[^|]*|//; s/split(dma,bm256):[^|]*|//; s/split(dma):[^|]*|//' | cut -c1-330
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_accept.py tests/test_gpu_edge.py tests/test_gpu_selftest.py -m gpu -q -k "fp32s or split" 2>&1 | tail -2
for w in 0 1 0 1; do
CTD_TUNING=split_halo_small=$w timeout 300 python bench.py --precision fp32s --mode net --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras --dump-ops $O/per_op_small$w.tsv > $O/bench_small$w.json 2> $O/bench_small$w.err
python -c "import json;d=json.load(open('$O/bench_small$w.json'));print('fp32s net split_halo_small=$w',d['value'],d['ms_per_step'])"
done
