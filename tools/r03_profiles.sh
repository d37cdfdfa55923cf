#!/bin/bash
# Round-3 profile set (run on the GPU box through gpurun): bench trace + PMC passes (tools/rocprof.sh), the multigrid-PCG solve's
# kernel stats, and the shader counters of its fine-level kernels.   usage: tools/r03_profiles.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
bash $R/tools/rocprof.sh $tag > $R/gpurun_out/rocprof_$tag.log 2>&1 || tail -5 $R/gpurun_out/rocprof_$tag.log
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_${tag}_mg
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/mg_bench.py > $out/mg_bench.txt 2> $out/err.txt || tail -5 $out/err.txt
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/kernel_stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("%-60s calls %5s avg %10.1f us total %9.1f ms %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void oi::", "").replace("oi::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete; find $out -name "*kernel_trace*" -delete
OI_SQ_SCRIPT=tools/mg_bench.py OI_SQ_KERNEL="k_cg1b<2, 8, double, false, true, 1>;k_cg1b<2, 8, double, false, true, 2>;k_cg1b<2, 8, double, false, true, 3>" bash $R/tools/sq_counters.sh ${tag}_mg > $R/gpurun_out/sq_${tag}_mg.txt 2>&1 || tail -5 $R/gpurun_out/sq_${tag}_mg.txt
find $R/gpurun_out/prof_${tag} $R/gpurun_out/prof_${tag}_mg $R/gpurun_out/sq_${tag}_mg -name "*.csv" -size +2M -delete 2>/dev/null
du -sh $R/gpurun_out
