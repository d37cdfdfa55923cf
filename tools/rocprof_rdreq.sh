#!/bin/bash
# HBM READ traffic by request size (gfx950: TCC_EA0_RDREQ_32B / _64B / _128B; FETCH_SIZE tallies every non-32-byte request at 64 bytes, so
# its x2 rule is right only where all requests are 128-byte ones): ONE counter pass over a short bench.py run that holds the timed CG
# region, the stand-alone apply and the multigrid solve.  bytes = 32 n32 + 64 n64 + 128 n128, calibrated on the copy streams of the same
# run (k_stream_3r3w reads exactly 3 vectors).  Run on the GPU box through gpurun.
# usage: tools/rocprof_rdreq.sh <tag> [bench.py args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
A="--no-cg-solve --no-cpu-baseline --no-variants --steps 10 --warmup 3"
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/pmc_rdreq -- python3 $R/bench.py $A "$@" > $out/bench_rdreq.json 2> $out/rdreq.err || { tail -20 $out/rdreq.err; exit 1; }
echo rdreq-done
python3 $R/tools/rdreq_summary.py $out > $out/rdreq_summary.txt
cat $out/rdreq_summary.txt
find $out -name "*.db" -delete
find $out -name "*agent_info*" -delete
find $out -name "*.csv" -size +2M -delete
