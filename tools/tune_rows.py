"""Tile-row sweep of the two-kernel (fused) and capacity (lean) iterations and the multigrid fine path at one size.
usage: python tools/tune_rows.py [size]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, max(4, 24 * n // 1024), 0.4)
for kernel, name in ((oia.KERNEL_FUSED, "fused"), (oia.KERNEL_LEAN, "lean")):
    for frows in (8, 16, 32):
        for arows in (8, 16, 32):
            if kernel == oia.KERNEL_FUSED and arows != 32: continue
            if kernel == oia.KERNEL_LEAN and frows != 16: continue
            oia.set_tuning(frows, 0, arows)
            t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=kernel, profile=1, maxiter=100000)
            t.cg_begin(); t.cg_iterate(6); t.sync(); t.reset_kernel_times()
            t0 = time.time(); t.cg_iterate(40); t.sync(); dt = time.time() - t0
            kt = {k: round(v[0] / v[1], 3) for k, v in t.kernel_times().items() if v[1]}
            print(name, "fused_rows", frows, "apply_rows", arows, "| %.3f ms/iter |" % (dt / 40 * 1e3), kt, flush=True)
            t.destroy()
oia.set_tuning()
