"""Per-launch times of the k_cg1b forms for a (timing-only) build named by OI_LIB_PATH: FORM 0 (single-kernel iteration) and the
multigrid-PCG iteration's FORM 1 / FORM 2 / cycle (coarse levels + FORM 3), 1024^3 spheres, fixed iteration counts."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
name = os.path.basename(os.environ.get("OI_LIB_PATH", "default"))
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync(); t.reset_kernel_times()
t.cg_iterate(40); t.sync()
kt = t.kernel_times()
f0 = kt["cg1"][0] / kt["cg1"][1]
t.destroy()
tm = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, precond=oia.PRECOND_MG, profile=1, maxiter=100000)
tm.cg_begin(); tm.cg_iterate(4); tm.sync(); tm.reset_kernel_times()
tm.cg_iterate(12); tm.sync()
mk = tm.kernel_times()
g = lambda k: mk[k][0] / max(1, mk[k][1])
print(f"{name:36s} FORM 0 {f0:.3f} ms | FORM 1 {g('fused_a'):.3f} | FORM 2 {g('update_r'):.3f} | cycle (coarse + FORM 3) {g('mg_vcycle'):.3f}", flush=True)
tm.destroy()
