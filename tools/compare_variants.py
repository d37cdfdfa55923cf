"""Iteration counts / answers of the CG variants on a large hard case (development aid)."""
import sys, time
sys.path.insert(0, ".")
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "b"
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
shape = (n, n, n)
buf = oia.gen_bernoulli_device(shape, 20251031, 0.5) if kind == "b" else oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
for name, kw in (("unfused", dict(kernel=oia.KERNEL_UNFUSED)), ("fused", dict(kernel=oia.KERNEL_FUSED)),
                 ("fused1", dict(kernel=oia.KERNEL_FUSED1)), ("fused1+p32", dict(kernel=oia.KERNEL_FUSED1, p_storage=1)),
                 ("fused1+jac", dict(kernel=oia.KERNEL_FUSED1, precond=1)), ("fused+jac", dict(kernel=oia.KERNEL_FUSED, precond=1)),
                 ("mg-pcg", dict(precond=oia.PRECOND_MG, profile=1))):
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=eps, maxiter=200000, **kw)
    t0 = time.time(); tau = t.value(); dt = time.time() - t0
    st = t.stats()
    print("%-11s tau %.10f status %d iters %6d relres %.3e flux %.8f / %.8f  solve %.2fs setup %.2fs" % (
        name, tau, t.status, st.iterations, st.final_relres, st.flux_in, st.flux_out, st.t_solve_s, st.t_setup_s), flush=True)
    if name == "mg-pcg":
        print("   ", {k: (round(v[0] / v[1], 3), v[1]) for k, v in t.kernel_times().items()})
    t.destroy()
