#!/bin/bash
# cycle-shape / over-correction sweep of the multigrid preconditioner: tools/mg_sweep.sh <out file>
out=${1:-gpurun_out/mg_sweep.txt}
for cfg in "1024 spheres" "768 blobs" "512 bernoulli"; do
  set -- $cfg
  for m in 0 14 15 30 31 62; do
    for sc in 1.0 1.25 1.5; do
      OI_MG_WMASK=$m OI_MG_SCALE=$sc timeout -k 10 120 python tools/mg_bench.py $1 1e-6 $2 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['size'], d['structure'], 'wmask', d['wmask'], 'scale', d['scale'], 'iters', d['iterations'], 'sec', d['seconds'], 'ms/it', d['ms_per_iteration'], 'vcycle', d['kernels_ms'].get('mg_vcycle'), flush=True)
    except Exception: print(l.strip())
" | tee -a $out
    done
  done
done
