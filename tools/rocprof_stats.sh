#!/bin/bash
# kernel-trace + stats of an arbitrary bench.py invocation: tools/rocprof_stats.sh <tag> [bench args]
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py "$@" > $out/bench.json 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/kernel_stats.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print("%-44s calls %5s avg %10.1f us total %9.1f ms %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void oi::", "").replace("oi::", "")[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
find $out -name "*.db" -delete
