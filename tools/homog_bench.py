"""Homogenization (three periodic cell problems + tensor) timing on one GPU: python tools/homog_bench.py [size] [mg|cg]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
buf = oia.gen_spheres_device((n, n, n), 20251031, max(4, 24 * n // 1024), 0.4)
ph = buf.download(np.uint8, (n, n, n)); buf.free()
t0 = time.perf_counter()
pre = sys.argv[2] if len(sys.argv) > 2 else 'mg'
h = oia.EffectiveDiffusivityMI355(ph, 1, eps=1e-9, precond=oia.PRECOND_MG if pre == 'mg' else oia.PRECOND_NONE)
t1 = time.perf_counter()
res = [h.solve(k) for k in range(3)]
t2 = time.perf_counter()
D = h.tensor()
print(pre, "%d^3: create %.2f s, 3 solves %.2f s, iterations %s, D diag %s" % (n, t1 - t0, t2 - t1, [r[1] for r in res], np.round(np.diag(D), 6)), flush=True)
print("ms per iteration: %.2f" % (1e3 * (t2 - t1) / max(1, sum(r[1] for r in res))))
