#!/bin/bash
# rocprofv3 kernel stats of the homogenization path at 1024^3 (tools/homog_bench.py): tools/homog_stats.sh <tag>
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/homog_bench.py 1024 "$@" > $out/homog_bench.txt 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/kernel_stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:20]:
    print("%-60s calls %5s avg %10.1f us total %9.1f ms %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void oi::", "").replace("oi::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
cat $out/homog_bench.txt
find $out -name "*.db" -delete
find $out -name "*.csv" -size +2M -delete
