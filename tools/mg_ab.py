"""Multigrid-PCG solve timed on its SECOND run in one process (no first-touch page clearing in the kernel times).
usage: OI_LIB_PATH=... python tools/mg_ab.py [size] [layout]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2:
    oia.set_cg1_layout(int(sys.argv[2]))
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, max(4, 24 * n // 1024), 0.4)
t = oia.TortuosityMI355(buf, 1, 0, 0.0, 1.0, shape=shape, eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=2000)
buf.free()
for run in range(3):
    t.reset_kernel_times()
    t0 = time.perf_counter()
    t.cg_begin()
    done, its = False, 0
    while not done and its < 2000:
        t.cg_iterate(4)
        its, rel, done = t.cg_state()
    t.cg_finish()
    dt = time.perf_counter() - t0
    k = {a: round(v[0] / max(1, v[1]), 3) for a, v in t.kernel_times().items() if v[1] and a in ("fused_a", "update_r", "mg_vcycle")}
    print(os.path.basename(os.environ.get("OI_LIB_PATH", "default")), "run", run, "iterations", its, "seconds", round(dt, 4), k, flush=True)
