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
configs = [("cg1", 8, 0), ("cg1", 16, 0), ("cg1", 32, 0), ("cg1", 33, 0), ("cg1", 16, 1), ("cg1", 32, 1)]
for kern, rows, pf32 in configs:
    oia.set_tuning(16, 0, 32, rows)
    k = oia.KERNEL_FUSED1 if kern == "cg1" else oia.KERNEL_FUSED
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=k, profile=1, p_storage=pf32, maxiter=100000)
    t.cg_begin(); t.cg_iterate(10); t.sync(); t.reset_kernel_times()
    t0 = time.time(); t.cg_iterate(50); t.sync(); dt = time.time() - t0
    kt = t.kernel_times()
    print("%-5s rows %2d p_f32 %d | %.2f ms/iter %.0f Mvox-it/s | " % (kern, rows, pf32, dt / 50 * 1e3, n ** 3 * 50 / dt / 1e6),
          {kk: round(v[0] / v[1], 3) for kk, v in kt.items()}, "| state", t.cg_state(), flush=True)
    t.cg_finish(); t.destroy()
