"""k_mg_post4t against k_mg_post4: bit-identical solves (flow-through and homogenization, ragged shapes, option mg_coarse_tiled = 2 / 0),
then the time of a 1024^3 multigrid-PCG solve with the option on / off.  python tools/mg_coarse_tiled_check.py [n=1024]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
rng = np.random.default_rng(5)
ok = True
for shape in [(40, 72, 256), (33, 70, 132), (64, 64, 64), (20, 130, 260), (9, 17, 68), (48, 96, 384), (12, 20, 134), (10, 18, 130), (16, 24, 268), (24, 40, 300)]:
    ph = (rng.random(shape) < 0.75).astype(np.uint8)
    res = {}
    for v in (2, 0):
        oia.set_option("mg_coarse_tiled", v)
        t = oia.TortuosityMI355(ph, 1, 0, eps=1e-10, precond=oia.PRECOND_MG)
        tau = t.value(); st = t.stats(); x = t.solution(); t.destroy()
        h = oia.EffectiveDiffusivityMI355(ph, 1, eps=1e-10, precond=oia.PRECOND_MG)
        hs = [h.solve(k) for k in range(3)]; chi = h.getChiSolution(1); D = h.tensor(); h.destroy()
        res[v] = (tau, st.iterations, x, hs, chi, D)
    same = res[2][0] == res[0][0] and res[2][1] == res[0][1] and np.array_equal(res[2][2], res[0][2]) and res[2][3] == res[0][3] \
        and np.array_equal(res[2][4], res[0][4]) and np.array_equal(res[2][5], res[0][5])
    ok = ok and same
    print(shape, "flow-through", res[2][1], "iterations, homogenization", [r[1] for r in res[2][3]], "bit-identical:", same, flush=True)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if n:
    shape = (n, n, n)
    for rep in range(3):
        for v in (1, 0):
            oia.set_option("mg_coarse_tiled", v)
            buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
            t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=1000)
            buf.free()
            for w in range(2):
                t.reset_kernel_times(); t.sync()
                t0 = time.perf_counter(); t.cg_begin(); done, its = False, 0
                while not done and its < 1000:
                    t.cg_iterate(4); its, rel, done = t.cg_state()
                t.cg_finish(); dt = time.perf_counter() - t0
            mk = t.kernel_times(); g = lambda k: mk[k][0] / max(1, mk[k][1])
            print(f"mg_coarse_tiled = {v}: {dt:.4f} s, {its} iterations | FORM 1 {g('fused_a'):.3f} FORM 2 {g('update_r'):.3f} cycle {g('mg_vcycle'):.3f} ms", flush=True)
            t.destroy()
oia.set_option("mg_coarse_tiled", 1)
