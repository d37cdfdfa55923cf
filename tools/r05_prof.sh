# xcheck tests, the bench line, then the two rocprof recipes (single-kernel CG + apply; multigrid path); everything under gpurun_out/r05
set -x
mkdir -p gpurun_out/r05
tag=${1:-a}
timeout -k 10 900 python -m pytest tests/test_xcheck_gpu.py tests/test_bench_contract.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/pytest_x_$tag.txt 2>&1
echo "rc=$?" >> gpurun_out/r05/pytest_x_$tag.txt
tail -5 gpurun_out/r05/pytest_x_$tag.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_$tag.json 2> gpurun_out/r05/bench_$tag.err
echo "bench rc=$?"
OI_PMC_KEY=1024:1 timeout -k 10 900 bash tools/rocprof_mg.sh r05mg_$tag > gpurun_out/r05/rocprof_mg_$tag.txt 2>&1
echo "rocprof_mg rc=$?"
tail -40 gpurun_out/r05/rocprof_mg_$tag.txt
