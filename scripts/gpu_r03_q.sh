#!/bin/bash
# round 3, GPU call Q: split-plane activations in the fp32s engine (selftest: layouts against the f32-MFMA kernel and float64;
# engine: parity tests, forward with split_planes 0 / 1, per-op table)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03q
mkdir -p $O
( cd comic-text-detector_amd && ST_SPLIT=1 timeout 500 ./ctd_selftest 32 ) > $O/split_selftest_b32.txt 2>&1
grep -E "^\[split\]|selftest" $O/split_selftest_b32.txt | sed 's/err vs f64 rms [0-9.e+-]* max [0-9.e+-]*,//g; s/max|d| vs f32-MFMA [0-9.e+-]* ([0-9]* > 2e-5),//g; s/f32-MFMA:[^|]*|//; s/split(reg):[^|]*|//; s/split(dma,bm256):[^|]*|//; s/split(dma):[^|]*|//' | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_accept.py tests/test_gpu_edge.py -m gpu -q -k "fp32s or split" > $O/pytest_fp32s.txt 2>&1
tail -4 $O/pytest_fp32s.txt | cut -c1-300
for w in 0 1; do
CTD_TUNING=split_halo=$w timeout 300 python bench.py --precision fp32s --mode net --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras --dump-ops $O/per_op_fp32s_halo$w.tsv > $O/bench_fp32s_net_halo$w.json 2> $O/bench_fp32s_net_halo$w.err
python -c "import json;d=json.load(open('$O/bench_fp32s_net_halo$w.json'));print('halo=$w net',d['value'],d['ms_per_step'],d['roofline']['net_ms_per_step'])"
done
timeout 300 python bench.py --precision fp32s --steps 20 --warmup 5 --spinup 30 --no-cpu-baseline --no-extras > $O/bench_fp32s_e2e.json 2> $O/bench_fp32s_e2e.err
python -c "import json;d=json.load(open('$O/bench_fp32s_e2e.json'));print('fp32s e2e',d['value'],d['ms_per_step'])"
