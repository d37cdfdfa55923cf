"""Multigrid PCG on the bundled 100^3 sample (the reference's own runnable case): launch-bound, so the
hipGraph capture of the cycle matters here.  usage: python tools/mg_small.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
ph = oia.TiffReader(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "SampleData_2Phase_stack_3d_1bit.tif")).threshold(0.5)
for precond, name in ((oia.PRECOND_MG, "multigrid PCG"), (oia.PRECOND_JACOBI, "Jacobi CG (cg1)")):
    best = None
    for rep in range(3):
        t = oia.TortuosityMI355(ph, 1, 0, -1.0, 1.0, eps=1e-9, precond=precond)
        t0 = time.perf_counter(); tau = t.value(); dt = time.perf_counter() - t0
        it = t.getSolverIterations(); t.destroy()
        best = dt if best is None else min(best, dt)
    print("%-16s graph=%s: tau %.9f, %d iterations, value() %.1f ms" % (name, os.environ.get("OI_MG_GRAPH", "1"), tau, it, 1e3 * best), flush=True)
