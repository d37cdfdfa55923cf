"""A/B timing of the single-kernel CG iteration on ONE box: python tools/ab_bench.py <size> [compact 0|1] [rows] [zchunk]
(the library under test comes from OI_LIB_PATH)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]); compact = int(sys.argv[2]) if len(sys.argv) > 2 else 1; rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
zchunk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
oia.set_cg1_layout(compact)
oia.set_tuning(16, zchunk, 32, rows)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync(); t.reset_kernel_times()
t0 = time.time(); t.cg_iterate(60); t.sync(); dt = time.time() - t0
kt = t.kernel_times()
print("%s compact %d rows %2d zchunk %d | %.3f ms/iter | cg1 %.3f ms | bytes %.1f GB | state %s" % (
    os.path.basename(os.environ.get("OI_LIB_PATH", "default")), compact, rows, zchunk, dt / 60 * 1e3, kt["cg1"][0] / kt["cg1"][1],
    t.stats().bytes_device / 1e9, t.cg_state()), flush=True)
