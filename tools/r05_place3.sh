# rounds of: a process that writes the whole card and exits, then three fresh handle-building processes
set -x
mkdir -p gpurun_out/r05
out=gpurun_out/r05/place_after_hog_${2:-a}.txt
: > $out
for i in $(seq 1 ${1:-10}); do
  echo "== round $i" >> $out
  timeout -k 10 120 python tools/place_after_hog.py hog 0.95 >> $out 2>&1 || exit 1
  for j in 1 2 3; do timeout -k 10 120 python tools/place_after_hog.py one >> $out 2>&1 || exit 1; done
done
grep -c "second set" $out
