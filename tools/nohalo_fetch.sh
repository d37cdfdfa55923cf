#!/bin/bash
# FETCH_SIZE of k_cg1b per launch, shipped build vs the timing-only build that never reads the halo (tools/variant_build.sh nohalo3
# -DOI_TIMING_NOHALO=3): how much of the fetch above the algorithmic bytes IS the halo.  Run on the GPU box through gpurun.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default nohalo3; do
  out=$R/gpurun_out/nohalo_fetch_$v
  mkdir -p $out
  if [ $v = default ]; then unset OI_LIB_PATH; else export OI_LIB_PATH=$R/tools/libopenimpala_amd_$v.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out -- python3 $R/tools/ab_bench.py 1024 1 > $out/run.txt 2> $out/err.txt || tail -5 $out/err.txt
  python3 - "$out" "$v" <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
tot, n = 0.0, 0
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("void oi::k_cg1b") or r["Kernel_Name"].startswith("oi::k_cg1b") or "k_cg1b<" in r["Kernel_Name"]:
        if r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
print("%s: k_cg1b launches %d, FETCH_SIZE x 2 per launch %.2f GB" % (sys.argv[2], n, 2 * tot * 1024 / max(n, 1) / 1e9))
PY
  cat $out/run.txt
  find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
done
