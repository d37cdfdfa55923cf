#!/bin/bash
# per-launch timeline of ONE multigrid-PCG iteration (the 12th) at 1024^3: tools/mg_cycle_trace.sh
set -e
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/mg_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/run -- python3 $R/tools/mg_bench.py --reps 1 > $out/log.txt 2> $out/err.txt || { tail -20 $out/err.txt; exit 1; }
f=$(find $out/run -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' | tee $out/cycle.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void oi::", "") for r in rows]
# an iteration of the rotated form starts at k_cg1b<.., FORM 2> (residual update + pre-smoother + restriction)
starts = [i for i, n in enumerate(names) if n.startswith("k_cg1b<") and n.rstrip().endswith(", 2>")]
a, b = starts[11], starts[12]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = {}
for i in range(a, b):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    gx = rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", "?"))
    print("%9.1f us  +gap %6.1f  dur %8.1f us  grid %10s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, gx, names[i][:70]))
    prev_end = e
    k = names[i].split("<")[0]
    tot[k] = tot.get(k, 0) + (e - s) / 1e3
print("iteration span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("  %-28s %9.1f us" % (k, v))
PY
find $out -name "*.db" -delete; find $out -name "*.csv" -size +2M -delete
