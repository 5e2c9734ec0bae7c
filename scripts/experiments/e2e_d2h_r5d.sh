#!/bin/bash
# Round 5: what do the page-size device -> host copies (mask + refined mask, 64 MB per 32 pages) cost the end-to-end step, and
# does the runtime's choice of copy path matter?  The e2e trace shows them as `__amd_rocclr_copyBuffer` KERNELS (~155 us per
# 8 MB), i.e. shader copies that store to host memory next to the forward.  One box, 60 timed steps each.
cd "${GRAFT_REPO_ROOT:-$(pwd)}" || exit 1
run() { echo "$LABEL $*: $(python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['host_cpu_cores_used'])")"; }
LABEL="default" run
LABEL="no page downloads (tail_skip_page_download=1, measurement only)" CTD_TUNING=tail_skip_page_download=1 run
LABEL="every device -> host copy through the copy kernel (tail_dma_min=2^40)" CTD_TUNING=tail_dma_min=1099511627776 run
LABEL="HSA_ENABLE_SDMA=0 (everything by shader copies)" HSA_ENABLE_SDMA=0 run
LABEL="default" run
LABEL="no page downloads" CTD_TUNING=tail_skip_page_download=1 run
