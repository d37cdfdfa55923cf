"""EXPERIMENT build (-DOI_EXP_CACHEPOL): cache policies of the vector loads / stores of k_cg1b FORM 0, ONE handle, alternating."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = 1024
shape = (n, n, n)
# (ids of the table in launch_cg1b_k, oi_k_cg1b.hip, as it stands at the end of round 4; the earlier sweeps -- every load / store policy,
# per stored array, combinations -- used other tables: their records in profiles/r04/k_cg1b_cache_policies_sweep1..5*.txt name the entries)
names = {0: "default policy everywhere", 1: "shipped: r p x stored nt, x p(k-2) loaded nt", 2: "shipped, own instantiation", 3: "VAR 1: whole-line flag constant",
         4: "VAR 2: no halo ROW reads (wrong results)", 5: "VAR 4: no halo COLUMN reads (wrong results)", 6: "VAR 6: neither (wrong results)",
         7: "shipped + r loaded nt", 8: "shipped + q loaded nt", 9: "shipped + p loaded nt"}
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
    print(f"{v:2d} {names[v]:48s} " + " ".join(f"{x:.3f}" for x in res[v]) + f"  | median {sorted(res[v])[1]:.3f} ms", flush=True)
oia.set_option("cg1_nt", 0)
t.destroy()
