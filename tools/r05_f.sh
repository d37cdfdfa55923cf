# BASELINE config 5's per-GPU share (4096 x 4096 x 512, porosity 0.2 blobs): the bench line, SQ counters, request-size counters, writes
set -x
mkdir -p gpurun_out/r05
C5="--size 4096 --nz 512 --structure blobs"
timeout -k 10 500 python bench.py --gpus 1 --steps 10 --warmup 3 $C5 --no-solve --no-cpu-baseline --no-variants > gpurun_out/r05/bench_c5_share.json 2> gpurun_out/r05/bench_c5_share.err
echo "bench c5 rc=$?"
timeout -k 10 500 python bench.py --gpus 1 --steps 10 --warmup 3 $C5 --precond jacobi --no-solve --no-cpu-baseline --no-variants --no-apply-bench > gpurun_out/r05/bench_c5_share_jacobi.json 2>> gpurun_out/r05/bench_c5_share.err
echo "bench c5 jacobi rc=$?"
timeout -k 10 900 bash tools/sq_counters.sh r05_c5 $C5 --steps 8 --warmup 2 > gpurun_out/r05/sq_c5_share.txt 2>&1
echo "sq rc=$?"
tail -40 gpurun_out/r05/sq_c5_share.txt
timeout -k 10 400 bash tools/rocprof_rdreq.sh r05rd_c5 $C5 --no-solve --no-apply-bench --steps 8 --warmup 2 > gpurun_out/r05/rdreq_c5_share.txt 2>&1
echo "rdreq rc=$?"
grep "k_cg1b\|^kernel" gpurun_out/r05/rdreq_c5_share.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r05_c5_write -- python3 $GRAFT_REPO_ROOT/bench.py $C5 --no-solve --no-apply-bench --no-cpu-baseline --no-variants --steps 8 --warmup 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r05/c5_write.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_r05_c5_write/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(lambda:[0.0,0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]!="WRITE_SIZE": continue
    k=r["Kernel_Name"].split("(")[0]
    acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
for k,(v,n) in sorted(acc.items(),key=lambda kv:-kv[1][0])[:4]:
    print("WRITE_SIZE %-70s launches %4d  %.0f bytes/launch"%(k[:70],n,v/n*1024))
PY
find gpurun_out/prof_r05_c5_write -name "*.db" -delete; find gpurun_out/prof_r05_c5_write -name "*.csv" -size +2M -delete
