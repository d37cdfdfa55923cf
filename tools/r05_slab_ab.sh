# split-launch + one-rank RCCL path of a 1024 x 1024 x 128 slab: which of the two launches runs on the solver's stream
F="--steps 300 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants --nz 128"
for rep in 1 2 3; do
for mode in new old; do
  export OI_SPLIT_ORDER=$mode
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('interior launch on the solver stream: $mode', round(d['value']), 'Mvox-it/s', round(d['ms_per_step'],3), 'ms', d['kernels_ms_per_launch'], d['vector_placement']['ms_kept'])"
done
done
timeout -k 10 600 python -m pytest tests/test_gpu_slabs.py tests/test_bench_contract.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
