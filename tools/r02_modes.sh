#!/bin/bash
# Round 2: the timing modes of k_cg1 (same binary, consecutive processes on ONE box).  Each process prints its ms per
# launch and the device addresses of its vectors; processes 1-6 run plain, then each of three processes runs under
# rocprofv3 with memory-side stall counters of the L2 (TCC), to compare a fast and a slow placement.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/modes_r02
mkdir -p $out
for p in 1 2 3 4 5 6; do python3 $R/tools/ab_bench.py 1024 1 32 >> $out/plain.txt 2>&1; done
for p in 1 2 3 4; do
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum --output-format csv -d $out/pmc_a$p -- python3 $R/tools/ab_bench.py 1024 1 32 >> $out/pmc_a.txt 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum --output-format csv -d $out/pmc_b$p -- python3 $R/tools/ab_bench.py 1024 1 32 >> $out/pmc_b.txt 2>&1
done
python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    ks = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not fs: continue
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(fs[0])):
        if "k_cg1" not in r["Kernel_Name"]: continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    dur = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(ks[0])) if "k_cg1" in r["Kernel_Name"]] if ks else []
    print(os.path.basename(d), "k_cg1 avg ms %.3f" % (sum(dur) / max(1, len(dur))), {k: "%.4g" % (v / n) for k, (v, n) in sorted(acc.items())})
PY
find $out -name "*.db" -delete; find $out -name "*agent_info*" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -size +2M -delete
