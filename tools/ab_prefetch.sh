#!/bin/bash
# same-box A/B of the cg1 look-ahead touch: layout arg 1 = compact + prefetch, 3 = compact, no prefetch
for rep in 1 2 3; do for c in 1 3; do timeout -k 10 100 python tools/ab_bench.py ${1:-1024} $c 0 || exit 1; done; done
