# multigrid-PCG (dense and compacted vectors), two-kernel and capacity-mode iterations: 16- vs 8-voxel lines
L16=$PWD/openimpala_amd/libopenimpala_amd.so
L8=$PWD/openimpala_amd/libopenimpala_amd_lv8.so
F="--steps 100 --warmup 10 --no-solve --no-cpu-baseline --no-apply-bench --no-variants"
for lib in $L16 $L8 $L16 $L8; do
  echo "== $(basename $lib)"
  OI_LIB_PATH=$lib python tools/mg_bench.py 1024 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mg dense  ', d['iterations'], d['seconds'], d['device_GB'], {k: d['kernels_ms'][k] for k in ('fused_a','update_r','mg_vcycle')})"
  OI_LIB_PATH=$lib OI_LAYOUT=5 python tools/mg_bench.py 1024 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mg compact', d['iterations'], d['seconds'], d['device_GB'], {k: d['kernels_ms'][k] for k in ('fused_a','update_r','mg_vcycle')})"
  for k in fused lean; do
    OI_LIB_PATH=$lib python bench.py --kernel $k $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k', round(d['value']), round(d['ms_per_step'],3))"
  done
done
