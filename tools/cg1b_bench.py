"""k_cg1b against k_cg1, ms per launch at 1024^3 in ONE process (same vectors, same placement), over tile shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import openimpala_amd as oia
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--structure", default="spheres")
ap.add_argument("--trials", type=int, default=0)
a = ap.parse_args()
oia.set_option("place_trials", a.trials)
n = a.size
buf, desc = bench.make_structure(oia, a, (n, n, n), 0, n)
t = oia.TortuosityMI355(buf, 1, oia.Direction.X, 0.0, 1.0, shape=(n, n, n), eps=1e-300, kernel=oia.KERNEL_FUSED1,
                        precond=oia.PRECOND_NONE, maxiter=100000, profile=1)
print(t.placement())
t.cg_begin(); t.cg_iterate(10); t.sync()
def run(label):
    t.retune()
    t.cg_iterate(4); t.sync()
    k0 = t.kernel_times()["cg1"]
    t.cg_iterate(40); t.sync()
    k1 = t.kernel_times()["cg1"]
    print(f"{label:40s} {(k1[0] - k0[0]) / (k1[1] - k0[1]):.3f} ms", flush=True)
import itertools
res = []
for rep in range(2):
    for kern, rows, zc, tm in itertools.chain([(1, 32, 0, 2)], itertools.product((0,), (8, 16, 32), (0, 16, 32, 64), (0, 2))):
        oia.set_option("cg1_kernel", kern); oia.set_option("tile_map", tm); oia.set_tuning(16, zc, 32, rows)
        run(f"kernel {'k_cg1 ' if kern else 'k_cg1b'} rows {rows} zchunk {zc} tile_map {tm}")
oia.set_tuning(); oia.set_option("cg1_kernel", 0); oia.set_option("tile_map", 2)
