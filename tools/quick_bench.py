"""Scratch A/B timing helper used during development (not part of the contract).
usage: quick_bench.py N [b|s] : sweeps tile rows / p storage on one structure."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import openimpala_amd as oia

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "s"
shape = (n, n, n)
buf = oia.gen_bernoulli_device(shape, 20251031, 0.5) if kind == "b" else oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
configs = [(16, 0, 0), (16, 0, 1), (8, 0, 0)]
for rows, zchunk, pf32 in configs:
    oia.set_tuning(rows, zchunk, 32)
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED, profile=1, p_storage=pf32, maxiter=100000)
    ms_apply = t.bench_apply(10, oia.KERNEL_FUSED)
    t.cg_begin(); t.cg_iterate(10); t.sync(); t.reset_kernel_times()
    t0 = time.time(); t.cg_iterate(50); t.sync(); dt = time.time() - t0
    kt = t.kernel_times()
    fa = kt["fused_a"][0] / kt["fused_a"][1]; ur = kt["update_r"][0] / kt["update_r"][1]
    bpv = 41 if pf32 else 49
    print("rows %2d zchunk %4d p_f32 %d | apply %.3f ms (%.0f GB/s) | fused_a %.3f ms (%.0f GB/s @%dB) update_r %.3f ms | %.2f ms/iter %.0f Mvox-it/s" % (
        rows, zchunk, pf32, ms_apply, 17 * n ** 3 / ms_apply / 1e6, fa, bpv * n ** 3 / fa / 1e6, bpv, ur, dt / 50 * 1e3, n ** 3 * 50 / dt / 1e6), flush=True)
    t.cg_finish(); t.destroy()
