set -x
mkdir -p gpurun_out/r05
tag=${1:-d}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py tests/test_xcheck_gpu.py tests/test_app.py -m gpu -x -q -p no:cacheprovider -k "initial_residual or slabs or xcheck or cross_check or chi_plotfiles or rev_study or known_answers or value_caches" > gpurun_out/r05/pytest_$tag.txt 2>&1
echo "rc=$?" >> gpurun_out/r05/pytest_$tag.txt
tail -5 gpurun_out/r05/pytest_$tag.txt
for pad in 0 1 0 1; do
OI_SEG_PAD=$pad timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-variants --no-cg-solve > gpurun_out/r05/bench_${tag}_pad$pad.json 2> gpurun_out/r05/bench_$tag.err
echo "bench pad $pad rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/r05/bench_${tag}_pad$pad.json').read().strip().splitlines()[-1])
m=d['time_to_1e-6_multigrid_pcg']
print('pad $pad', d['value'], d['roofline']['ms_per_launch'], d['device_bytes'], m['seconds'], m['kernels_ms_per_launch'])
PY
done
OI_SEG_PAD=1 timeout -k 10 600 bash tools/rocprof_rdreq.sh r05rd_${tag}_pad1 > gpurun_out/r05/rocprof_rdreq_${tag}_pad1.txt 2>&1
echo "rdreq rc=$?"
head -12 gpurun_out/r05/rocprof_rdreq_${tag}_pad1.txt
