"""Nontemporal stores in the dense-layout kernels of the homogenization path (option dense_nt, a bit mask read when a handle is built):
one handle per mask in ONE process, the three cell problems solved in turn, ms per iteration.  usage: python tools/homog_nt_ab.py [size] mask ..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
masks = [int(v) for v in sys.argv[2:]] or [0, 31]
buf = oia.gen_spheres_device((n, n, n), 20251031, max(4, 24 * n // 1024), 0.4)
ph = buf.download(np.uint8, (n, n, n)); buf.free()
res = {m: [] for m in masks}
ref = None
for rep in range(2):
    for m in masks:
        oia.set_option("dense_nt", m)
        h = oia.EffectiveDiffusivityMI355(ph, 1, eps=1e-9, precond=oia.PRECOND_MG)
        t0 = time.perf_counter()
        r = [h.solve(k) for k in range(3)]
        dt = time.perf_counter() - t0
        D = h.tensor().copy()
        h.destroy()
        if ref is None: ref = D
        assert np.array_equal(D, ref), "the cache policy changed the tensor"
        res[m].append(1e3 * dt / sum(x[1] for x in r))
oia.set_option("dense_nt", 0)
for m in masks:
    print(f"dense_nt = {m:2d}: " + " ".join(f"{x:.2f}" for x in res[m]) + " ms per iteration", flush=True)
