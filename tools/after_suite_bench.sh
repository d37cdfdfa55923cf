#!/bin/bash
# What the round-end driver does on one box: a process that dirties most of the card (here: 240 GB written, tools/place_alloc_times.py
# dirty), then `python bench.py`.  Prints the placement and set-up part of the bench line.   usage: tools/after_suite_bench.sh out.json
python tools/place_alloc_times.py dirty ${2:-28}
python bench.py --no-solve --no-variants --no-cpu-baseline --no-apply-bench --steps 40 --warmup 10 > "$1" 2>/dev/null
python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
p = d["vector_placement"]
print("value", round(d["value"]), "kernel", round(d["roofline"]["ms_per_launch"], 3), "nslots", p["nslots"], "in-order", round(p["ms_in_order"], 3), "kept", round(p["ms_kept"], 3),
      "alloc_s", round(p["alloc_s"], 2), "search_s", round(p["search_s"], 2), "create", round(d["setup_s"]["create_total"], 2),
      d["roofline"]["stream_3r3w_same_vectors"].get("placement_class"))
PY
