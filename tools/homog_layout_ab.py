"""Homogenization at n^3 on compacted vectors with the rotated iteration (default for whole-tile volumes) against dense vectors with the
unrotated one (`mg_norot` = 1): create, three solves, tensor; alternating handles.  python tools/homog_layout_ab.py [n=1024] [eps=1e-9]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-9
buf = oia.gen_spheres_device((n, n, n), 20251031, max(4, 24 * n // 1024), 0.4)
ph = buf.download(np.uint8, (n, n, n)); buf.free()
D = {}
for rep in range(3):
    for norot in (0, 1):
        oia.set_option("mg_norot", norot)
        t0 = time.perf_counter()
        h = oia.EffectiveDiffusivityMI355(ph, 1, eps=eps, precond=oia.PRECOND_MG)
        t1 = time.perf_counter()
        res = [h.solve(k) for k in range(3)]
        t2 = time.perf_counter()
        D[norot] = h.tensor()
        lay = h.layout()
        h.destroy()
        its = [r[1] for r in res]
        print(f"mg_norot = {norot} (compacted {lay[0]}, {lay[1] * 8 / 1e9:.2f} GB per vector): create {t1 - t0:.3f} s, 3 solves {t2 - t1:.3f} s, "
              f"iterations {its}, {1e3 * (t2 - t1) / max(1, sum(its)):.2f} ms per iteration, converged {[r[0] for r in res]}", flush=True)
oia.set_option("mg_norot", 0)
print("tensor difference (max abs):", float(np.abs(D[0] - D[1]).max()), " diag", np.round(np.diag(D[0]), 8))
