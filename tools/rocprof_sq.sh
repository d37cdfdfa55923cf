#!/bin/bash
# SQ instruction / cycle counters per kernel of a multigrid-PCG solve: tools/rocprof_sq.sh <tag> "<counters>" [mg_bench args]
set -e
tag=$1; shift
ctr=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/run -- python3 $R/tools/mg_bench.py "$@" > $out/log.txt 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
f=$(find $out/run -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/summary.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void oi::", "").replace("oi::", "")
    if not (name.startswith("k_mg_fine") or name.startswith("k_fused_a") or name.startswith("k_cg1") or name.startswith("k_update_r")): continue
    a = acc[name][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c, (v, n) in sorted(acc[k].items()):
        print("   %-26s per launch %.4g  (%d launches)" % (c, v / n, n))
PY
find $out -name "*.db" -delete
