"""Homogenization on compacted vectors over many small shapes, handles created back to back (each inherits the arrays the previous one
released, full of ITS vectors): every solve must converge.  python tools/homog_oddnx_probe.py [alternate]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
alternate = len(sys.argv) > 1
rng = np.random.default_rng(3)
bad = []
for nz, ny in ((8, 16), (6, 10), (5, 7)):
    for nx in list(range(5, 44)) + [63, 65, 95, 97, 125, 127, 129, 131, 133, 255, 257]:
        shape = (nz, ny, nx)
        sub = (rng.random(shape) < 0.7).astype(np.uint8)
        if alternate:
            oia.set_option("mg_norot", 1)
            h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG, maxiter=60)
            h.solve(0); h.destroy()
            oia.set_option("mg_norot", 0)
        h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG, maxiter=60)
        res = [h.solve(k) for k in range(3)]
        lay = h.layout()[0]
        h.destroy()
        ok = all(r[0] for r in res)
        if not ok: bad.append(shape)
        print(shape, "compacted", lay, [r[1] for r in res], "ok" if ok else "FAIL", flush=True)
print("failing:", bad)
