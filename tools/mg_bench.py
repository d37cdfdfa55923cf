"""Time-to-tolerance of the multigrid-preconditioned CG on one GPU.
usage: python tools/mg_bench.py [size] [eps] [structure: spheres|blobs|bernoulli] [direction]"""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
structure = sys.argv[3] if len(sys.argv) > 3 else "spheres"
d = int(sys.argv[4]) if len(sys.argv) > 4 else 0
shape = (n, n, n)
if "OI_LAYOUT" in os.environ:  # 0 = dense CG vectors, 1 = compacted lines (default)
    oia.set_cg1_layout(int(os.environ["OI_LAYOUT"]))
if structure == "spheres":
    buf = oia.gen_spheres_device(shape, 20251031, max(4, 24 * n // 1024), 0.4)
elif structure == "blobs":
    buf = oia.gen_blobs_device(shape, 20251031, max(4, 16 * n // 1024), 0.2)
else:
    buf = oia.gen_bernoulli_device(shape, 20251031, 0.5)
t = oia.TortuosityMI355(buf, 1, d, 0.0, 1.0, shape=shape, eps=eps, precond=oia.PRECOND_MG, profile=1, maxiter=2000)
buf.free()
t0 = time.perf_counter()
t.cg_begin()
done, its = False, 0
while not done and its < 2000:
    t.cg_iterate(4)
    its, rel, done = t.cg_state()
t.cg_finish()
dt = time.perf_counter() - t0
k = {a: round(v[0] / max(1, v[1]), 3) for a, v in t.kernel_times().items() if v[1]}
print(json.dumps({"size": n, "structure": structure, "eps": eps, "wmask": os.environ.get("OI_MG_WMASK", "default"), "scale": os.environ.get("OI_MG_SCALE", "default"),
                  "iterations": its, "relres": rel, "seconds": round(dt, 4), "ms_per_iteration": round(1e3 * dt / max(1, its), 2),
                  "setup_s": round(t.stats().t_setup_s, 3), "device_GB": round(t.stats().bytes_device / 1e9, 2), "tau": t.stats().tau, "kernels_ms": k}), flush=True)
t.destroy()
