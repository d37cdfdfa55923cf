set -x
mkdir -p gpurun_out/r05
timeout -k 10 1100 bash tools/place_runs.sh ${1:-80} gpurun_out/r05/place_runs_retry_${2:-b}.txt --no-apply-bench
