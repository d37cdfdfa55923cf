"""Spread of the single-kernel iteration time over vector placements, in THIS process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import openimpala_amd as oia
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--trials", type=int, default=32)
ap.add_argument("--structure", default="spheres")
ap.add_argument("--nz", type=int, default=0)
ap.add_argument("--slots", type=int, default=12)
ap.add_argument("--handles", type=int, default=1, help="handles built one after the other and KEPT (each lands elsewhere in HBM)")
a = ap.parse_args()
oia.set_option("place_trials", a.trials)
oia.set_option("place_slots", a.slots)
n = a.size
nz = a.nz or n
buf, desc = bench.make_structure(oia, a, (nz, n, n), 0, nz)
keep = []
for k in range(a.handles):
    t = oia.TortuosityMI355(buf, 1, oia.Direction.X, 0.0, 1.0, shape=(nz, n, n), eps=1e-300, kernel=oia.KERNEL_FUSED1,
                            precond=oia.PRECOND_NONE, maxiter=1000, verbose=3)
    keep.append(t)
    print("handle", k, t.placement(), flush=True)
