# the placement tests, then N consecutive bench processes (one line each: was a second set of allocations searched -- tried 96 -- and what was kept)
set -x
mkdir -p gpurun_out/r05
N=${1:-20}
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -p no:cacheprovider -k "placement or reproducible" > gpurun_out/r05/pytest_place.txt 2>&1
echo "rc=$?" >> gpurun_out/r05/pytest_place.txt
tail -5 gpurun_out/r05/pytest_place.txt
timeout -k 10 900 bash tools/place_runs.sh $N gpurun_out/r05/place_runs_retry.txt --no-apply-bench
cat gpurun_out/r05/place_runs_retry.txt
