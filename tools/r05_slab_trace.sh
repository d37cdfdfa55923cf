# device timeline of the split-launch + one-rank RCCL path on a 128-plane slab
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_r05_slabtrace -- python3 $R/bench.py --force-dist --nz 128 --steps 60 --warmup 10 --no-solve --no-cpu-baseline --no-apply-bench --no-variants > $R/gpurun_out/r05/slab_trace_bench.json 2> $R/gpurun_out/r05/slab_trace.err
cd $R
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_r05_slabtrace/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0].replace("void ","")[:60],r.get("Stream_Id","?"),r.get("Queue_Id","?")) for r in csv.DictReader(open(f))]
rows.sort()
# last 5 iterations: find the last k_finalize1 launches
idx=[i for i,r in enumerate(rows) if 'k_finalize1' in r[2]]
a=idx[-6]; t0=rows[a][0]
for s,e,k,st,q in rows[a:idx[-1]+3]:
    print(f"{(s-t0)/1e3:9.1f} us  +{(e-s)/1e3:8.1f} us  q{q:>3s} s{st:>3s}  {k}")
PY
