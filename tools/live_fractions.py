"""Fraction of aligned x-runs of g voxels that hold a free voxel (what a compacted layout of that granularity stores).
usage: python tools/live_fractions.py [size] [structure]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
structure = sys.argv[2] if len(sys.argv) > 2 else "spheres"
shape = (n, n, n)
if structure == "spheres":
    buf = oia.gen_spheres_device(shape, 20251031, max(4, 24 * n // 1024), 0.4)
elif structure == "blobs":
    buf = oia.gen_blobs_device(shape, 20251031, max(4, 16 * n // 1024), 0.2)
else:
    buf = oia.gen_bernoulli_device(shape, 20251031, 0.5)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape)
st = t.stats()
print(structure, n, "free fraction", st.n_free / n ** 3, "active", st.active_vf)
for g in (2, 4, 8, 16, 32, 64):
    print("group", g, "live fraction", t.live_fraction(g))
