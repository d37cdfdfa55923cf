#!/bin/bash
# SQ counters of the CG kernel: where do the wave cycles go?  tools/rocprof_pmc.sh <tag> "<counter list>" [ab_bench args]
set -e
tag=$1; shift
ctr=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/run -- python3 $R/tools/ab_bench.py "$@" > $out/log.txt 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
f=$(find $out/run -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/summary.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void oi::", "")
    if not name.startswith("k_cg1"): continue
    a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print("%-28s per launch %.4g  (%d launches)" % (k, v / n, n))
PY
find $out -name "*.db" -delete
