cd "$(dirname "$0")/.."
F="--steps 300 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants"
for nz in 128 256; do
  for zc in 0 16 24 32; do
    OI_ZCHUNK=$zc python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist --nz $nz $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nz', $nz, 'rccl path, OI_ZCHUNK', $zc, round(d['value']), 'Mvox-it/s', round(d['ms_per_step'],3), 'ms', d['kernels_ms_per_launch'], 'kept', round(d['vector_placement']['ms_kept'],3))"
  done
done
