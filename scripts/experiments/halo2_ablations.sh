#!/bin/bash
# Timing-only ablations of the big-tile ConvT kernel (kernels_halo2.hip; selftest build): ABL bits 1 no stagger, 2 no DMA in
# the loop, 4 no MFMAs, 8 no fragment reads, 16 no barriers.  usage (GPU box): bash scripts/experiments/halo2_ablations.sh [list]
cd comic-text-detector_amd
for A in ${1:-0 1 2 4 8 6 10 12 14 30}; do
  echo "ABL=$A: $(ST_H2_ABL=$A ST_CASES=16,18 ST_NO_C3=1 ST_VAR=0 ./ctd_selftest 32 2>&1 | grep '\[case\]' | sed 's/B=32 out [0-9x]*//' | cut -c1-110 | tr '\n' ' ')"
done
echo "noprio: $(ST_H2_NOPRIO=1 ST_CASES=16,18 ST_NO_C3=1 ST_VAR=0 ./ctd_selftest 32 2>&1 | grep '\[case\]' | sed 's/B=32 out [0-9x]*//' | cut -c1-110 | tr '\n' ' ')"
