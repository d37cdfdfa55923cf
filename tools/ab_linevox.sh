# single-kernel CG iteration with compacted lines of 16, 8 and 4 voxels (libraries built with -DOI_LINE_VOX=...)
F="--steps 200 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants"
for rep in 1 2; do
for lib in libopenimpala_amd.so libopenimpala_amd_lv8.so libopenimpala_amd_lv4.so; do
  OI_LIB_PATH=$PWD/openimpala_amd/$lib python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spheres', '$lib', round(d['value']), round(d['ms_per_step'],3), d['device_bytes']/1e9)"
  OI_LIB_PATH=$PWD/openimpala_amd/$lib python bench.py --structure blobs $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blobs  ', '$lib', round(d['value']), round(d['ms_per_step'],3), d['device_bytes']/1e9)"
done; done
