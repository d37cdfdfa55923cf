#!/bin/bash
# Per-rank work of the N-GPU strong-scaling runs rehearsed on ONE GPU: 1024 x 1024 x nz slabs, once as a plain
# single-domain launch and once through the split-launch + RCCL path (one-rank communicator).
# usage: tools/slab_rehearsal.sh > gpurun_out/slab_rehearsal.txt
cd "$(dirname "$0")/.."
F="--steps 300 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants"
for nz in 128 256 512; do
  python bench.py --nz $nz $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nz', $nz, 'plain', round(d['value']), 'Mvox-it/s', round(d['ms_per_step'],3), 'ms', d['kernels_ms_per_launch'])"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --nz $nz $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nz', $nz, 'rccl ', round(d['value']), 'Mvox-it/s', round(d['ms_per_step'],3), 'ms', d['kernels_ms_per_launch'])"
done
