# the same 128-plane slab through the split-launch + one-rank RCCL path: this round's tree against round 4's (r04tree/, a git worktree of effe3f1)
F="--steps 300 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants --nz 128"
for rep in 1 2 3; do
for tree in . r04tree; do
  (cd $tree && python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force-dist $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree', '$tree', round(d['value']), 'Mvox-it/s', round(d['ms_per_step'],3), 'ms', d['kernels_ms_per_launch'], d['vector_placement']['ms_kept'])")
done
done
