"""Multigrid-PCG solve at n^3 with the search direction stored as fp64 (default) against fp32 (opts.p_storage): iterations, time,
tau.  python tools/mg_pf32_ab.py [n=1024] [eps=1e-6]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
shape = (n, n, n)
for rep in range(2):
    for ps in (oia.STORE_F64, oia.STORE_F32):
        buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
        t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=eps, precond=oia.PRECOND_MG, profile=1, maxiter=1000, p_storage=ps)
        buf.free()
        for w in range(2):
            t.reset_kernel_times(); t.sync()
            t0 = time.perf_counter(); t.cg_begin(); done, its = False, 0
            while not done and its < 1000:
                t.cg_iterate(4); its, rel, done = t.cg_state()
            t.cg_finish(); dt = time.perf_counter() - t0
        mk = t.kernel_times(); g = lambda k: mk[k][0] / max(1, mk[k][1])
        tau = t.value()
        print(f"p stored as {'fp32' if ps else 'fp64'}: {dt:.4f} s, {its} iterations, relres {rel:.3e}, tau {tau:.10f} | FORM 1 {g('fused_a'):.3f} FORM 2 {g('update_r'):.3f} "
              f"cycle {g('mg_vcycle'):.3f} ms, {t.stats().bytes_device / 1e9:.1f} GB", flush=True)
        t.destroy()
