# A/B of an environment switch on the multigrid-PCG solve (third solve of a process): tools/mg_ab_env.sh VAR
for rep in 1 2; do
  python tools/mg_ab.py 1024 | tail -1
  env $1=1 python tools/mg_ab.py 1024 | tail -1 | sed "s/^/$1=1 /"
done
