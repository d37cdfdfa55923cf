#!/bin/bash
# Shader-side counters of the CG kernel (is it issue-bound, latency-bound, LDS-bound?): a few --pmc passes of a short
# bench run (run on the GPU box through gpurun).  usage: tools/sq_counters.sh <tag> [bench.py args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/sq_$tag
mkdir -p $out
n=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES_EQ_64 TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  n=$((n+1))
  if [ -n "$OI_SQ_SCRIPT" ]; then  # another workload (e.g. tools/mg_bench.py) and another kernel (OI_SQ_KERNEL)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$n -- python3 $R/$OI_SQ_SCRIPT "$@" > $out/bench_$n.json 2> $out/p$n.err || { tail -5 $out/p$n.err; continue; }
  else
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$n -- python3 $R/bench.py --no-solve --no-cpu-baseline --no-variants --no-apply-bench --steps 20 --warmup 4 "$@" > $out/bench_$n.json 2> $out/p$n.err || { tail -5 $out/p$n.err; continue; }
  fi
  echo pass-$n-done
done
python3 - $out <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        want = os.environ.get("OI_SQ_KERNEL", "k_cg1").split(";")   # (kernel names hold commas: ';' separates several)
        hit = [w for w in want if w in k]
        if not hit: continue
        acc[hit[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    for c, v in sorted(d.items()):
        v = v[len(v)//4:]  # drop the placement-search / warm-up launches' share
        print(f"{k:24s} {c:34s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
PY
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete; find $out -name "*kernel_trace*" -delete
find $out -name "*.csv" -size +2M -delete   # (gpurun brings back 64 MiB at most: the summary above is what is kept)
