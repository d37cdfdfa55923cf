"""A volume that is narrow in x (the rows of its coarse levels shorter than a tile): multigrid-PCG with the tiled coarse kernels (default)
against the one-cell kernels (mg_coarse_tiled = 0).  python tools/narrow_volume_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
for shape in [(768, 768, 96), (1024, 1024, 60), (512, 512, 200)]:
    buf = oia.gen_spheres_device(shape, 20251031, 12, 0.4)
    for rep in range(2):
        for v in (1, 0):
            oia.set_option("mg_coarse_tiled", v)
            t = oia.TortuosityMI355(buf, 1, 1, shape=shape, eps=1e-9, precond=oia.PRECOND_MG, maxiter=500)
            t.value()
            t0 = time.perf_counter(); tau = t.value(refresh=True); dt = time.perf_counter() - t0
            st = t.stats(); t.destroy()
            print(f"(nz, ny, nx) = {shape} mg_coarse_tiled = {v}: {st.iterations} iterations, {1e3 * dt:.1f} ms, {1e3 * dt / st.iterations:.3f} ms each, tau {tau:.9f}", flush=True)
    buf.free()
oia.set_option("mg_coarse_tiled", 1)
