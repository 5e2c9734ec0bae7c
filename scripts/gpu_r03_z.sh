#!/bin/bash
# round 3, GPU call Z: the stem + layer-1 kernel next to the REAL dual labelling (one stream / three streams) under three grid caps
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03z
mkdir -p $O
for cap in 100000 256; do
( cd comic-text-detector_amd && GPU_MAX_HW_QUEUES=8 CTD_TUNING=tail_max_blocks=$cap ST_CORUN=1 timeout 300 ./ctd_selftest 32 ) > $O/corun_cap$cap.txt 2>&1
grep -E 'real kernels' $O/corun_cap$cap.txt | grep 'prio 1' | sed "s/^/cap $cap: /" | cut -c1-200
done
