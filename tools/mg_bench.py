"""Multigrid-PCG at 1024^3: per-launch times of the cycle and the two-kernel iteration, solve time to 1e-6 (two solves in one process)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import openimpala_amd as oia
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--structure", default="spheres")
ap.add_argument("--nu", default="0", help="comma-separated mg_nu_coarse values, alternated (0 = by the structure)")
ap.add_argument("--wmask", default="-1", help="comma-separated mg_wmask values (-1 = default)")
ap.add_argument("--zchunk", default="0", help="comma-separated z-chunk lengths of the marching kernels (0 = automatic)")
ap.add_argument("--scale", default="0", help="comma-separated mg_scale values (0 = default 1.5)")
ap.add_argument("--reps", type=int, default=2, help="passes over the combinations")
ap.add_argument("--opt", action="append", default=[], help="key=v1,v2,...: any oi_tort_set_option key, alternated like the others")
a = ap.parse_args()
n = a.size
buf, desc = bench.make_structure(oia, a, (n, n, n), 0, n)
import itertools
extra = [(o.split("=")[0], o.split("=")[1].split(",")) for o in a.opt]
combos = list(itertools.product(a.nu.split(","), a.wmask.split(","), a.zchunk.split(","), a.scale.split(","), *[v for _, v in extra]))
for rep in range(a.reps * len(combos)):
    nu, wm, zc, sc = combos[rep % len(combos)][:4]
    ex = combos[rep % len(combos)][4:]
    for (key, _), v in zip(extra, ex):
        oia.set_option(key, float(v))
    nu, wm, zc, sc = int(nu), int(wm), int(zc), float(sc)
    oia.set_option("mg_nu_coarse", nu); oia.set_option("mg_wmask", wm); oia.set_option("zchunk", zc); oia.set_option("mg_scale", sc)
    tm = oia.TortuosityMI355(buf, 1, oia.Direction.X, 0.0, 1.0, shape=(n, n, n), eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=1000)
    t0 = time.perf_counter()
    tm.cg_begin()
    done, its, rel = False, 0, 1.0
    while not done and its < 1000:
        tm.cg_iterate(4)
        its, rel, done = tm.cg_state()
    tm.cg_finish()
    dt = time.perf_counter() - t0
    mk = tm.kernel_times()
    print(" ".join(f"{key} {v}" for (key, _), v in zip(extra, ex)), f"nu_coarse {nu} wmask {wm} zchunk {zc} scale {sc} solve {rep}: {its} iterations, relres {rel:.3e}, {dt:.3f} s,",
          {k: round(v[0] / max(1, v[1]), 3) for k, v in mk.items() if v[1]}, flush=True)
    tm.destroy()
