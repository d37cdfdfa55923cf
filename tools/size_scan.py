"""Per-iteration time of the default (multigrid-PCG) flow-through solve and of the homogenization solves over a range of cubic sizes:
looks for cliffs (a size that falls off a fast path).  python tools/size_scan.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
for n in [32, 48, 64, 80, 96, 100, 128, 160, 200, 256, 300, 384, 500, 512, 640]:
    shape = (n, n, n)
    buf = oia.gen_spheres_device(shape, 20251031, max(4, 24 * n // 1024), 0.4)
    ph = buf.download(np.uint8, shape)
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-9, precond=oia.PRECOND_MG, maxiter=500)
    buf.free()
    t.value()  # warm
    t0 = time.perf_counter(); tau = t.value(refresh=True); dt = time.perf_counter() - t0
    st = t.stats(); t.destroy()
    h = oia.EffectiveDiffusivityMI355(ph, 1, eps=1e-9, precond=oia.PRECOND_MG)
    h.solve(0)
    t0 = time.perf_counter(); r = h.solve(1); dh = time.perf_counter() - t0
    h.destroy()
    print(f"n {n:4d}: flow-through {st.iterations:3d} iterations {1e3 * dt:8.2f} ms ({1e3 * dt / max(1, st.iterations):7.3f} ms each, {n**3 * st.iterations / dt / 1e6:9.0f} Mvoxel-it/s) | "
          f"homogenization {r[1]:3d} iterations {1e3 * dh:8.2f} ms ({1e3 * dh / max(1, r[1]):7.3f} ms each)", flush=True)
