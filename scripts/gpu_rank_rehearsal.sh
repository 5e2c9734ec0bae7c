#!/bin/bash
# 8-rank HOST rehearsal on a 1-GPU box: `bench.py --gpus N --tail-only` with every rank on
# device 0 over gloo -- N interpreter pipelines, N x tail workers x geometry threads, N pinned arenas and the record gather
# inside the step, but no forward: what do the ranks' host sides cost each other?  Then the full step (forward + tail) at
# N = 1, 2 for reference (the GPU is shared, so those are NOT scaling numbers).
# usage: gpurun --timeout 900 -- 'bash scripts/gpu_rank_rehearsal.sh'  ->  gpurun_out/ranks/summary.txt
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT" || exit 1
O=$ROOT/gpurun_out/ranks
mkdir -p "$O"
echo "host: $(nproc) logical CPUs, $(python -c 'import os;print(len(os.sched_getaffinity(0)))') usable" | tee "$O/summary.txt"
for N in 1 2 4 8; do
  timeout 600 python bench.py --gpus $N --tail-only --steps ${STEPS:-30} --warmup 5 --spinup 20 --no-cpu-baseline --no-extras > "$O/tail_only_n$N.json" 2> "$O/tail_only_n$N.err"
  python - "$O/tail_only_n$N.json" $N <<'PY' | tee -a "$O/summary.txt"
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    hb = d["config"]["host_threads"]
    print(f"tail-only N={sys.argv[2]}: {d['value']:.0f} pages/s aggregate, {d['ms_per_step']:.2f} ms per step of {d['config']['global_batch']} pages; per rank: "
          f"{hb['tail_workers']} workers x {hb['native_threads_per_worker']} native threads (budget {hb['per_rank']} of {hb['usable_cpus']} CPUs), "
          f"rank 0 used {d['config'].get('host_cpu_cores_used')} cores of CPU time")
except Exception as e:
    print("N=%s failed: %r" % (sys.argv[2], e))
PY
done
for N in 1 2; do
  timeout 600 python bench.py --gpus $N --steps 20 --warmup 5 --spinup 20 --no-cpu-baseline --no-extras > "$O/e2e_n$N.json" 2> "$O/e2e_n$N.err"
  python -c "
import json,sys
d=json.loads([l for l in open('$O/e2e_n$N.json') if l.startswith('{')][-1]); print('full step N=$N (one device shared):', d['value'], 'pages/s', d['ms_per_step'], 'ms; rank 0 used', d['config'].get('host_cpu_cores_used'), 'cores of CPU time')" | tee -a "$O/summary.txt"
done
