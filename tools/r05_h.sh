set -x
mkdir -p gpurun_out/r05
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py -m gpu -x -q -p no:cacheprovider -k "initial_residual or slabs_match" > gpurun_out/r05/pytest_h.txt 2>&1
echo "rc=$?"; tail -3 gpurun_out/r05/pytest_h.txt
A="--no-cg-solve --no-cpu-baseline --no-variants --no-apply-bench --steps 10 --warmup 3"
for i in 1 2 3; do
python bench.py $A > gpurun_out/r05/bench_h$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r05/bench_h$i.json').read().strip().splitlines()[-1])
m=d['time_to_1e-6_multigrid_pcg']
print('run $i', d['value'], m['seconds'], m['launches'], m['iterations'], m['kernels_ms_per_launch'])
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05_h -- python3 $GRAFT_REPO_ROOT/bench.py $A > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
grep "k_residual0\|k_init_field\|fillBuffer" gpurun_out/prof_r05_h/*/*kernel_stats.csv | cut -c1-200
