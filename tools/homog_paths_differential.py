"""Differential run of the two homogenization paths over many small random volumes: compacted vectors + rotated iteration (default)
against dense vectors + unrotated iteration (mg_norot = 1): every solve converges, iteration counts within one, tensors equal to 1e-9.
python tools/homog_paths_differential.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
rng = np.random.default_rng(7)
worst, bad, n = 0.0, [], 0
for trial in range(120):
    shape = tuple(int(v) for v in rng.integers(4, 70, size=3))
    if trial % 4 == 0: shape = (shape[0], shape[1], int(rng.integers(120, 280)))
    sub = (rng.random(shape) < rng.uniform(0.45, 0.9)).astype(np.uint8)
    out = {}
    for norot in (0, 1):
        oia.set_option("mg_norot", norot)
        h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG, maxiter=300)
        res = [h.solve(k) for k in range(3)]
        out[norot] = (h.layout()[0], res, h.tensor().copy())
        h.destroy()
    conv = all(r[0] for r in out[0][1]) and all(r[0] for r in out[1][1])
    dit = max(abs(a[1] - b[1]) for a, b in zip(out[0][1], out[1][1]))
    dD = float(np.abs(out[0][2] - out[1][2]).max())
    worst = max(worst, dD)
    n += 1
    if not conv or dit > 1 or dD > 1e-9 or not out[0][0] or out[1][0]:
        bad.append((shape, conv, dit, dD, out[0][0], out[1][0]))
        print("MISMATCH", shape, conv, dit, dD, flush=True)
oia.set_option("mg_norot", 0)
print(f"{n} volumes, worst tensor difference {worst:.2e}, mismatches: {len(bad)}")
sys.exit(1 if bad else 0)
