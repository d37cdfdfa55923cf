"""Round 3 A/B in ONE process (same vectors, same placement): k_cg1b with line slots from segment records (scalar loads)
against the per-line slot table, and the z-chunk length, ms per launch at 1024^3 (or --size N)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import openimpala_amd as oia
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--nz", type=int, default=0)
ap.add_argument("--structure", default="spheres")
ap.add_argument("--chunks", default="0")
a = ap.parse_args()
n = a.size
nz = a.nz or n
buf, desc = bench.make_structure(oia, a, (nz, n, n), 0, nz)
t = oia.TortuosityMI355(buf, 1, oia.Direction.X, 0.0, 1.0, shape=(nz, n, n), eps=1e-300, kernel=oia.KERNEL_FUSED1,
                        precond=oia.PRECOND_NONE, maxiter=100000, profile=1)
buf.free()
print(desc, t.placement(), flush=True)
t.cg_begin(); t.cg_iterate(10); t.sync()
def run(label):
    t.retune()
    t.cg_iterate(4); t.sync()
    k0 = t.kernel_times()["cg1"]
    t.cg_iterate(40); t.sync()
    k1 = t.kernel_times()["cg1"]
    print(f"{label:40s} {(k1[0] - k0[0]) / (k1[1] - k0[1]):.3f} ms", flush=True)
for rep in range(3):
    for zc in [int(c) for c in a.chunks.split(",")]:
        for seg in (0, 1):
            oia.set_option("cg1_seg", seg); oia.set_option("zchunk", zc)
            run(f"seg {seg} zchunk {zc}")
oia.set_option("cg1_seg", 1); oia.set_option("zchunk", 0)
