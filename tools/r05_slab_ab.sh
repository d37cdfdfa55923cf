# split-launch + one-rank RCCL path of a 1024 x 1024 x 128 slab: how the exchange stream is created
F="--steps 300 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants --nz 128"
for rep in 1 2 3; do
for mode in greatest flags least; do
  export OI_COMM_PRIO=$mode
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('exchange stream: $mode', round(d['value']), 'Mvox-it/s', round(d['ms_per_step'],3), 'ms', d['kernels_ms_per_launch'], d['vector_placement']['ms_kept'])"
done
done
