"""In-process sweep of one named library option on ONE multigrid-PCG handle (same vectors, alternating batches):
ms per launch of FORM 1 / FORM 2 / the cycle (coarse levels + FORM 3).  usage: python tools/opt_sweep_mg.py <option> <v1> <v2> ..."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
key, vals = sys.argv[1], [float(v) for v in sys.argv[2:]]
n = 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
tm = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, precond=oia.PRECOND_MG, profile=1, maxiter=100000)
tm.cg_begin(); tm.cg_iterate(4); tm.sync()
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        oia.set_option(key, v); tm.retune()
        tm.cg_iterate(2); tm.sync(); tm.reset_kernel_times()
        tm.cg_iterate(10); tm.sync()
        mk = tm.kernel_times()
        g = lambda k: mk[k][0] / max(1, mk[k][1])
        res[v].append((g("fused_a"), g("update_r"), g("mg_vcycle")))
for v in vals:
    for i, name in enumerate(("FORM 1", "FORM 2", "cycle ")):
        xs = [r[i] for r in res[v]]
        print(f"{name}  {key} = {v:g}: " + " ".join(f"{x:.3f}" for x in xs) + f"  | median {sorted(xs)[1]:.3f} ms", flush=True)
oia.set_option(key, vals[0])
tm.destroy()
