# sets of allocations at different places of ONE process's memory: seven 1024^3 handles built and kept, no cache, no retry -> what each set's search finds
set -x
mkdir -p gpurun_out/r05
out=gpurun_out/r05/place_sets_${2:-a}.txt
: > $out
for i in $(seq 1 ${1:-6}); do
  echo "== process $i" >> $out
  OI_PLACE_CACHE=0 OI_PLACE_RETRY_GBPS=${3:-0} timeout -k 10 200 python tools/place_probe.py --trials 48 --handles 7 2>&1 | grep -E "vector placement|^handle|rror" >> $out || exit 1
done
grep "vector placement" $out | awk '{print $(NF-9), $(NF-11)}' | sort -n | uniq -c
