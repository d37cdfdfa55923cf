"""EXPERIMENT build (-DOI_EXP_CACHEPOL): cache policies of the vector loads / stores of k_cg1b FORM 0, ONE handle, alternating."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = 1024
shape = (n, n, n)
names = {0: "st -", 1: "st r q p x", 2: "st r p, ld x", 3: "st r p x, ld x", 4: "st r, ld x", 5: "st r q p, ld x", 6: "st r p, ld x p(k-2)", 7: "st all, ld x", 8: "st r p",
         9: "st r x, ld x", 10: "st p, ld x", 11: "st -, ld x", 12: "st p x, ld x", 13: "st r p x, ld x p(k-2)", 14: "st r p, ld p(k-2)"}
vals = [int(v) for v in sys.argv[1:]] or sorted(names)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync()
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        oia.set_option("cg1_nt", v); t.retune()
        t.cg_iterate(4); t.sync(); t.reset_kernel_times()
        t.cg_iterate(30); t.sync()
        kt = t.kernel_times()
        res[v].append(kt["cg1"][0] / kt["cg1"][1])
for v in vals:
    print(f"{v:2d} {names[v]:24s} " + " ".join(f"{x:.3f}" for x in res[v]) + f"  | median {sorted(res[v])[1]:.3f} ms", flush=True)
oia.set_option("cg1_nt", 0)
t.destroy()
