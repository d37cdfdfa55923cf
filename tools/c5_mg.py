"""Multigrid-PCG to relres 1e-6 on config 5's per-GPU share (4096 x 4096 x 512 blobs at porosity 0.2), one GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openimpala_amd as oia
shape = (512, 4096, 4096)
buf = oia.gen_blobs_device(shape, 20251031, 24, 0.2)
for nu in (1, 2):
    oia.set_option("mg_nu_coarse", nu)
    t0 = time.perf_counter()
    tm = oia.TortuosityMI355(buf, 1, 0, 0.0, 1.0, shape=shape, eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=1000)
    tm.sync(); t1 = time.perf_counter()
    tm.cg_begin()
    done, its, rel = False, 0, 1.0
    while not done and its < 1000:
        tm.cg_iterate(4)
        its, rel, done = tm.cg_state()
    tm.cg_finish(); tm.sync(); t2 = time.perf_counter()
    mk = tm.kernel_times()
    print(f"nu_coarse {nu}: create {t1 - t0:.2f} s, {its} iterations, relres {rel:.3e}, solve {t2 - t1:.3f} s, device {tm.stats().bytes_device / 1e9:.1f} GB,",
          {k: round(v[0] / max(1, v[1]), 2) for k, v in mk.items() if v[1] and k in ("fused_a", "update_r", "mg_vcycle")}, flush=True)
    tm.destroy()
