"""In-process sweep of the cg1 tile shape on one handle (alternating batches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pf32 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000, p_storage=pf32)
t.cg_begin(); t.cg_iterate(10); t.sync()
res = {}
for rep in range(3):
    for rows in (8, 16, 32, 33):
        for zc in (0, 32, 128):
            oia.set_tuning(16, zc, 32, rows)
            t.reset_kernel_times(); t.cg_iterate(30); t.sync()
            kt = t.kernel_times()
            res.setdefault((rows, zc), []).append(kt["cg1"][0] / kt["cg1"][1])
oia.set_tuning()
for k, v in sorted(res.items(), key=lambda kv: sorted(kv[1])[1]):
    print("rows %2d zchunk %3d:" % k, " ".join("%.3f" % x for x in v), flush=True)
