# where do the reads of the single-kernel CG iteration beyond the algorithmic bytes come from?  Request-size counters of the timed
# region with one ingredient removed / changed each (EXPERIMENT library tools/libopenimpala_amd_exp.so: cg1_nt = 2 shipped policy,
# 4 no halo-row reads, 5 no halo-column reads, 6 neither -- wrong results, timing and traffic only), and with longer chunks
set -x
mkdir -p gpurun_out/r05
B="--no-solve --no-apply-bench"
for v in 2 4 5 6; do
  OI_LIB_PATH=$PWD/tools/libopenimpala_amd_exp.so timeout -k 10 300 bash tools/rocprof_rdreq.sh r05rd_e_nt$v $B --option cg1_nt=$v > gpurun_out/r05/rdreq_e_nt$v.txt 2>&1
  echo "nt $v rc=$?"; grep "k_cg1b<2, 8, double, false, true, 0" gpurun_out/r05/rdreq_e_nt$v.txt
done
for z in 48 96; do
  timeout -k 10 300 bash tools/rocprof_rdreq.sh r05rd_e_z$z $B --option zchunk=$z > gpurun_out/r05/rdreq_e_z$z.txt 2>&1
  echo "zchunk $z rc=$?"; grep "k_cg1b<2, 8, double, false, true, 0" gpurun_out/r05/rdreq_e_z$z.txt
done
for f in gpurun_out/prof_r05rd_e_*/bench_rdreq.json; do python - $f <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d['roofline']['ms_per_launch'])
PY
done
