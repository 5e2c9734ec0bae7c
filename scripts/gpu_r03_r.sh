#!/bin/bash
# round 3, GPU call R: full GPU suite + smoke at HEAD (split-plane / halo fp32s engine, tail count kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03r
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -3 $O/smoke.txt | cut -c1-300
