"""One process, the library named by OI_LIB_PATH: the multigrid-PCG solve of the 1024^3 sphere pack (warm-up solve + `reps` timed ones) with
per-kernel HIP-event times, then the single-kernel CG iteration (FORM 0) on its own handle; a digest of each solution (bit-identity across
builds).  usage: forms_ab.py [n=1024] [reps=2] [cg1=1]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cg1 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tag = os.path.basename(os.environ.get("OI_LIB_PATH", "default")).replace("libopenimpala_amd", "").replace(".so", "") or "product"
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=1000)
for rep in range(reps + 1):
    t.reset_kernel_times(); t.sync()
    t0 = time.perf_counter()
    t.cg_begin(); launches = t.cg_run(1000); t.cg_finish()
    dt = time.perf_counter() - t0
    its, rel, done = t.cg_state()
    mk = t.kernel_times()
    g = lambda k: mk[k][0] / max(1, mk[k][1])
    if rep:
        print(f"{tag:>10} mg: {dt:.4f} s {its} it {launches} launches | FORM1 {g('fused_a'):.3f} FORM2 {g('update_r'):.3f} cycle {g('mg_vcycle'):.3f} (post {g('mg_post'):.3f}) "
              f"sum {g('fused_a') + g('update_r') + g('mg_vcycle'):.3f} ms", flush=True)
print(f"{tag:>10} mg digest {hashlib.sha256(t.solution().tobytes()).hexdigest()[:16]} tau {t.value():.12f}", flush=True)
t.destroy()
if cg1:
    oia.set_option("place_trials", 24)
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, precond=oia.PRECOND_NONE, profile=1, maxiter=100000)
    t.cg_begin(); t.cg_iterate(10); t.sync(); t.reset_kernel_times()
    t.cg_iterate(60); t.sync()
    kt = t.kernel_times()
    its, rel, done = t.cg_state()
    print(f"{tag:>10} cg1: {kt['cg1'][0] / kt['cg1'][1]:.3f} ms | relres after {its} iterations {rel!r}", flush=True)
    t.cg_finish(); t.destroy()
