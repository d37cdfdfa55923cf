"""Round 2: in-process sweep of the block -> tile order (tile_map) and the z-chunk length on ONE handle
(1024^3 spheres, alternating batches): k_cg1 per launch and the stand-alone apply (both kernels)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync()
res, resa = {}, {}
maps = (0, 1, 2)
for rep in range(2):
    for tm in maps:
        for rows in (16, 32):
            for zc in (0, 8, 16, 32):
                oia.set_tuning(16, zc, 32, rows)
                oia.set_option("tile_map", tm)
                t.retune()
                t.reset_kernel_times(); t.cg_iterate(20); t.sync()
                kt = t.kernel_times()
                res.setdefault((tm, rows, zc), []).append(kt["cg1"][0] / kt["cg1"][1])
for k, v in sorted(res.items(), key=lambda kv: min(kv[1])):
    print("cg1   map %d rows %2d zchunk %3d:" % k, " ".join("%.3f" % x for x in v), "ms", flush=True)
t.cg_finish()
for rep in range(2):
    for ak in (0, 1):
        for tm in maps:
            for zc in (0, 8, 16, 32, 64):
                oia.set_tuning(16, zc, 32, 0)
                oia.set_option("tile_map", tm); oia.set_option("apply_kernel", ak)
                t.retune()
                resa.setdefault((ak, tm, zc), []).append(t.bench_apply(10))
oia.set_tuning(); oia.set_option("tile_map", 2); oia.set_option("apply_kernel", 0)
for k, v in sorted(resa.items(), key=lambda kv: min(kv[1])):
    print("apply kernel %d map %d zchunk %3d:" % k, " ".join("%.3f" % x for x in v), "ms   frac of 8 TB/s %.3f" % (17.0 * n ** 3 / min(v) / 1e6 / 8000.0), flush=True)
