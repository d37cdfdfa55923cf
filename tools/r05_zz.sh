set -x
mkdir -p gpurun_out/r05
timeout -k 10 900 python -m pytest tests/test_zz_robustness_gpu.py -x -q -p no:cacheprovider > gpurun_out/r05/zz1.txt 2>&1
echo "rc=$?" >> gpurun_out/r05/zz1.txt
tail -30 gpurun_out/r05/zz1.txt
