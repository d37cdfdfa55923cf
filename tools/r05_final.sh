# the round's record: full GPU suite once, the driver's bench line, the two rocprof recipes (trace + FETCH/WRITE passes), merged PMC summary
set -x
mkdir -p gpurun_out/r05
tag=${1:-final}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r05/pytest_gpu_$tag.txt 2>&1
rc=$?
echo "rc=$rc" >> gpurun_out/r05/pytest_gpu_$tag.txt
tail -6 gpurun_out/r05/pytest_gpu_$tag.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_$tag.json 2> gpurun_out/r05/bench_$tag.err
echo "bench rc=$?"
OI_PMC_KEY=1024:1 timeout -k 10 900 bash tools/rocprof.sh r05_$tag --steps 20 --warmup 5 > gpurun_out/r05/rocprof_$tag.txt 2>&1
echo "rocprof rc=$?"
OI_PMC_KEY=1024:1 timeout -k 10 900 bash tools/rocprof_mg.sh r05mg_$tag > gpurun_out/r05/rocprof_mg_$tag.txt 2>&1
echo "rocprof_mg rc=$?"
python3 - <<PY
import json
a=json.load(open('gpurun_out/prof_r05_$tag/pmc_summary.json'))
b=json.load(open('gpurun_out/prof_r05mg_$tag/pmc_summary.json'))
m=dict(b); m.update({k:v for k,v in a.items() if not k.startswith('mg_')})
json.dump(m, open('gpurun_out/r05/pmc_summary_$tag.json','w'), indent=1)
print(sorted(m))
PY
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-solve --no-variants --no-cpu-baseline > gpurun_out/r05/bench_${tag}_short.json 2>/dev/null
