"""k_cg1b tile shapes on ONE handle (set_tuning + retune): 16-row tiles of 8 waves (default), 32-row tiles of 16 waves, 8-row tiles;
chunk lengths 24 / 48.  python tools/cg1_tile_ab.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync()
cases = [(0, 0), (32, 0), (32, 16), (32, 32), (8, 0), (0, 0)]
res = {i: [] for i in range(len(cases))}
for rep in range(3):
    for ci, (rows, zc) in enumerate(cases):
        oia.set_tuning(cg1_tile_rows=rows); oia.set_option("zchunk", zc); t.retune()
        t.cg_iterate(4); t.sync(); t.reset_kernel_times()
        t.cg_iterate(30); t.sync()
        kt = t.kernel_times()
        res[ci].append(kt["cg1"][0] / kt["cg1"][1])
for ci, (rows, zc) in enumerate(cases):
    print(f"tile rows {rows or 'auto(16)'} zchunk {zc or 'auto'}: " + " ".join(f"{x:.3f}" for x in res[ci]) + f" | median {sorted(res[ci])[1]:.3f} ms", flush=True)
oia.set_tuning(); oia.set_option("zchunk", 0)
t.destroy()
