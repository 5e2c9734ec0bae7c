#!/bin/bash
# round 3, GPU call V: which kind of co-running kernel stretches the stem + layer-1 kernel (0.42 ms alone, 2.1 ms next to the tail)?
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03v
mkdir -p $O
( cd comic-text-detector_amd && ST_CORUN=1 timeout 300 ./ctd_selftest 32 ) > $O/corun.txt 2>&1
grep -E "^\[corun\]|^\[stem2\]|selftest" $O/corun.txt | cut -c1-200
