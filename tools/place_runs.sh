#!/bin/bash
# N consecutive bench processes (defaults): kernel ms, placement search figures, set-up cost of each -> one line per process
# usage: tools/place_runs.sh N outfile [extra bench args]
N=$1; OUT=$2; shift 2
: > "$OUT"
for i in $(seq 1 "$N"); do
  python bench.py --steps 40 --warmup 10 --no-solve --no-variants --no-cpu-baseline "$@" > /tmp/place_run.json 2>/tmp/place_run.err || { echo "run $i failed" >> "$OUT"; tail -3 /tmp/place_run.err >> "$OUT"; exit 1; }
  python - "$i" >> "$OUT" <<'PY'
import json, sys
d = json.load(open("/tmp/place_run.json"))
p, r, s = d["vector_placement"], d["roofline"], d["setup_s"]
st = r.get("stream_3r3w_same_vectors", {})
print(f"run {sys.argv[1]} value {d['value']:.0f} kernel {r['ms_per_launch']:.3f} frac {r['frac']:.3f} in-order {p['ms_in_order']:.3f} kept {p['ms_kept']:.3f} "
      f"tried {p['tried']} stream {st.get('GBps', 0):.0f} {st.get('placement_class')} create {s['create_total']:.2f} alloc {p['alloc_s']:.2f} search {p['search_s']:.2f} "
      f"device_GB {d['device_bytes'] / 1e9:.1f} peak_GB {p['peak_device_bytes'] / 1e9:.1f}", flush=True)
PY
done
