"""Multigrid-PCG at 1024^3 down to relres 1e-13 (X, Y, Z): iterations, final relres, tau, flux balance -- the solver must keep
converging far below the default tolerance with every round-4 kernel in the cycle.  python tools/mg_deep_convergence.py [n=1024]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
for d in range(3):
    for eps in (1e-9, 1e-13):
        for ps in (oia.STORE_F64, oia.STORE_F32):
            t = oia.TortuosityMI355(buf, 1, d, shape=shape, eps=eps, precond=oia.PRECOND_MG, maxiter=400, p_storage=ps)
            t0 = time.perf_counter(); tau = t.value(); dt = time.perf_counter() - t0
            st = t.stats()
            print(f"direction {'XYZ'[d]} eps {eps:g} p {'fp32' if ps else 'fp64'}: {st.iterations} iterations, relres {st.final_relres:.3e}, tau {tau:.12f}, "
                  f"flux in/out {st.flux_in:.9e} / {st.flux_out:.9e} (rel diff {abs(st.flux_in - st.flux_out) / abs(st.flux_in):.2e}), {dt:.3f} s", flush=True)
            t.destroy()
