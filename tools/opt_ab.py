"""In-process A/B of one named library option (oi_tort_set_option + oi_tort_retune) on ONE handle per solver path: same vectors,
same placement, alternating batches.  usage: python tools/opt_ab.py <option> <value A> <value B> [size]
Prints ms per launch of the single-kernel CG iteration (FORM 0) and of the multigrid-PCG iteration's kernels."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
key, va, vb = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync()
res = {va: [], vb: []}
for rep in range(4):
    for v in (va, vb):
        oia.set_option(key, v); t.retune()
        t.cg_iterate(4); t.sync(); t.reset_kernel_times()
        t.cg_iterate(40); t.sync()
        kt = t.kernel_times()
        res[v].append(kt["cg1"][0] / kt["cg1"][1])
for v in (va, vb):
    print(f"FORM 0  {key} = {v:g}: " + " ".join(f"{x:.3f}" for x in res[v]) + f"  | median {sorted(res[v])[len(res[v]) // 2]:.3f} ms", flush=True)
t.destroy()
oia.set_option(key, va)
tm = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, precond=oia.PRECOND_MG, profile=1, maxiter=100000)
tm.cg_begin(); tm.cg_iterate(4); tm.sync()
resm = {va: [], vb: []}
for rep in range(4):
    for v in (va, vb):
        oia.set_option(key, v); tm.retune()
        tm.cg_iterate(2); tm.sync(); tm.reset_kernel_times()
        tm.cg_iterate(10); tm.sync()
        mk = tm.kernel_times()
        g = lambda k: mk[k][0] / max(1, mk[k][1])
        resm[v].append((g("fused_a"), g("update_r"), g("mg_vcycle")))
for v in (va, vb):
    for i, name in enumerate(("FORM 1", "FORM 2", "cycle ")):
        xs = [r[i] for r in resm[v]]
        print(f"{name}  {key} = {v:g}: " + " ".join(f"{x:.3f}" for x in xs) + f"  | median {sorted(xs)[len(xs) // 2]:.3f} ms", flush=True)
tm.destroy()
oia.set_option(key, va)
