"""A/B of an option that is frozen when a multigrid-PCG handle is built or its CG run begins (mg_xpair, mg_blocked, ...): two handles,
value 1 against value 0, alternating batches; ms per launch of FORM 1 / FORM 2 / the cycle (coarse levels + FORM 3).
usage: python tools/mg_two_handles_ab.py <option>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
KEY = sys.argv[1]
n = 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
hs = {}
for v in (1, 0):
    oia.set_option(KEY, v)
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, precond=oia.PRECOND_MG, profile=1, maxiter=100000)
    t.cg_begin(); t.cg_iterate(4); t.sync()
    hs[v] = t
oia.set_option(KEY, 1)
res = {1: [], 0: []}
for rep in range(4):
    for v in (1, 0):
        t = hs[v]
        t.reset_kernel_times(); t.cg_iterate(12); t.sync()
        mk = t.kernel_times()
        g = lambda k: mk[k][0] / max(1, mk[k][1])
        res[v].append((g("fused_a"), g("update_r"), g("mg_vcycle")))
for v in (1, 0):
    for i, name in enumerate(("FORM 1", "FORM 2", "cycle ")):
        xs = [r[i] for r in res[v]]
        print(f"{name}  {KEY} = {v}: " + " ".join(f"{x:.3f}" for x in xs) + f"  | median {sorted(xs)[len(xs) // 2]:.3f} ms", flush=True)
for t in hs.values(): t.destroy()
