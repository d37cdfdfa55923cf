#!/bin/bash
# Profiling recipe of the DEFAULT solver path (multigrid-preconditioned CG) at the bench configuration: kernel trace + stats, then the
# two HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950), each over one bench.py run that holds the
# timed CG region (short) and the multigrid solve to 1e-6.  Run on the GPU box through gpurun.
# usage: tools/rocprof_mg.sh <tag> [bench.py args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
A="--no-cg-solve --no-cpu-baseline --no-variants --no-apply-bench --steps 10 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py $A "$@" > $out/bench_trace.json 2> $out/trace.err || { tail -20 $out/trace.err; exit 1; }
echo trace-done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py $A "$@" > $out/bench_fetch.json 2> $out/fetch.err || { tail -20 $out/fetch.err; exit 1; }
echo fetch-done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py $A "$@" > $out/bench_write.json 2> $out/write.err || { tail -20 $out/write.err; exit 1; }
echo write-done
python3 $R/tools/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
# keep only the small files
find $out -name "*.db" -delete
find $out -name "*agent_info*" -delete
find $out -name "*.csv" -size +2M -delete   # (the summary is what is kept; gpurun brings back 64 MiB at most)
