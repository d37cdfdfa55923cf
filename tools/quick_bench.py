"""Scratch timing helper used during development (not part of the contract)."""
import sys, time
import numpy as np
import openimpala_amd as oia

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
shape = (n, n, n)
t0 = time.time()
buf = oia.gen_bernoulli_device(shape, 20251031, 0.5) if (len(sys.argv) < 3 or sys.argv[2] == "b") else oia.gen_spheres_device(shape, 7, 24 if n >= 512 else 8, 0.4)
print("gen %.2fs" % (time.time() - t0), flush=True)
for variant in (oia.KERNEL_FUSED, oia.KERNEL_UNFUSED):
    t0 = time.time()
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-6, kernel=variant, profile=1)
    st = t.stats()
    print("variant", variant, "create %.2fs mask %.2fs sweeps %d vf %.4f avf %.4f nfree %d mem %.1f GB" % (time.time() - t0, st.t_mask_s, st.flood_sweeps, st.vf, st.active_vf, st.n_free, st.bytes_device / 1e9), flush=True)
    for v in (oia.KERNEL_FUSED, oia.KERNEL_UNFUSED):
        ms = t.bench_apply(20, v)
        print("  apply variant %d: %.3f ms  %.1f GB/s (17 B/voxel)" % (v, ms, 17 * n ** 3 / ms / 1e6), flush=True)
    t.reset_kernel_times()
    t.cg_begin(); t.cg_iterate(20); t.sync()
    t.reset_kernel_times()
    t0 = time.time()
    t.cg_iterate(100); t.sync()
    dt = time.time() - t0
    print("  100 iters %.3fs -> %.1f Mvoxel-iters/s ; state" % (dt, n ** 3 * 100 / dt / 1e6), t.cg_state(), flush=True)
    for k, (ms, cnt) in t.kernel_times().items():
        print("    %-12s %8.3f ms/launch x %d" % (k, ms / cnt, cnt))
    t.cg_finish()
    t.destroy()
