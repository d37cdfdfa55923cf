# A/B of an environment switch on the default bench workload: tools/ab_env.sh VAR [bench args]
V=$1; shift
F="--steps 200 --warmup 20 --no-solve --no-cpu-baseline --no-apply-bench --no-variants $@"
for rep in 1 2 3; do
  python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default ', round(d['value']), round(d['ms_per_step'],3))"
  env $V=1 python bench.py $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=1', round(d['value']), round(d['ms_per_step'],3))"
done
