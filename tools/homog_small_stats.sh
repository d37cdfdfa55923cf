#!/bin/bash
# rocprofv3 kernel stats of homogenization solves on a small volume: tools/homog_small_stats.sh <n>
set -e
n=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_homog_small_$n
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/homog_bench.py $n > $out/homog_bench.txt 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["Calls"]) for r in rows)
print("launches", tot)
for r in rows[:16]:
    print("%-60s calls %5s avg %8.1f us total %7.2f ms" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void oi::", "").replace("oi::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cat $out/homog_bench.txt
find $out -name "*.db" -delete
