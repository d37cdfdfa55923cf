set -x
mkdir -p gpurun_out/r05
tag=${1:-c}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py tests/test_xcheck_gpu.py tests/test_app.py -m gpu -x -q -p no:cacheprovider -k "initial_residual or slabs or xcheck or cross_check or chi_plotfiles or rev_study or known_answers or value_caches" > gpurun_out/r05/pytest_$tag.txt 2>&1
echo "rc=$?" >> gpurun_out/r05/pytest_$tag.txt
tail -5 gpurun_out/r05/pytest_$tag.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_$tag.json 2> gpurun_out/r05/bench_$tag.err
echo "bench rc=$?"
timeout -k 10 600 bash tools/rocprof_rdreq.sh r05rd_$tag > gpurun_out/r05/rocprof_rdreq_$tag.txt 2>&1
echo "rdreq rc=$?"
tail -30 gpurun_out/r05/rocprof_rdreq_$tag.txt
