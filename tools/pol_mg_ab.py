"""EXPERIMENT build (-DOI_EXP_CACHEPOL): cache policies of FORM 1 / 2 / 3 of k_cg1b (multigrid-PCG iteration), ONE handle, alternating.
Policy id i selects entry i of each form's own (load mask, store mask) list in oi_k_cg1b.hip."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = 1024
shape = (n, n, n)
f1 = ["-", "st p", "st q", "st p q", "st p q, ld z", "st p, ld z", "ld z", "st p, ld p"]
f2 = ["-", "st r", "st r, ld q", "st r, ld r", "st r, ld r q", "ld q", "ld r q", "ld r"]
f3 = ["-", "st z", "st x", "st z x", "st z x, ld x", "st z x, ld x p", "st x, ld x p", "ld x p"]
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
tm = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, precond=oia.PRECOND_MG, profile=1, maxiter=100000)
tm.cg_begin(); tm.cg_iterate(4); tm.sync()
res = {v: [] for v in range(8)}
for rep in range(3):
    for v in range(8):
        oia.set_option("cg1_nt", v); tm.retune()
        tm.cg_iterate(2); tm.sync(); tm.reset_kernel_times()
        tm.cg_iterate(8); tm.sync()
        mk = tm.kernel_times()
        g = lambda k: mk[k][0] / max(1, mk[k][1])
        res[v].append((g("fused_a"), g("update_r"), g("mg_vcycle")))
for i, (name, lst) in enumerate((("FORM 1", f1), ("FORM 2", f2), ("cycle (FORM 3)", f3))):
    for v in range(8):
        xs = [r[i] for r in res[v]]
        print(f"{name:14s} {v} {lst[v]:18s} " + " ".join(f"{x:.3f}" for x in xs) + f"  | median {sorted(xs)[1]:.3f} ms", flush=True)
oia.set_option("cg1_nt", 0)
tm.destroy()
