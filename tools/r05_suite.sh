# full GPU suite once, then the driver's bench line; everything lands under gpurun_out/r05
set -x
mkdir -p gpurun_out/r05
tag=${1:-a}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/pytest_gpu_$tag.txt 2>&1
rc=$?
echo "rc=$rc" >> gpurun_out/r05/pytest_gpu_$tag.txt
tail -15 gpurun_out/r05/pytest_gpu_$tag.txt
[ $rc -eq 0 ] && timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_$tag.json 2> gpurun_out/r05/bench_$tag.err
echo "bench rc=$?"
