"""In-process sweep of one named library option (oi_tort_set_option + oi_tort_retune) on ONE single-kernel CG handle: same vectors,
same placement, alternating batches, repeatable to 0.1 %.  usage: python tools/opt_sweep.py <option> <v1> <v2> ... [--size N] [--nz planes] [--structure spheres|blobs]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
args = sys.argv[1:]
n, structure = 1024, "spheres"
if "--size" in args: i = args.index("--size"); n = int(args[i + 1]); del args[i:i + 2]
if "--structure" in args: i = args.index("--structure"); structure = args[i + 1]; del args[i:i + 2]
nz = 0
if "--nz" in args: i = args.index("--nz"); nz = int(args[i + 1]); del args[i:i + 2]
key, vals = args[0], [float(v) for v in args[1:]]
shape = (nz or n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4) if structure == "spheres" else oia.gen_blobs_device(shape, 20251031, 12, 0.2)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync()
res = {v: [] for v in vals}
for rep in range(3):
    for v in vals:
        oia.set_option(key, v); t.retune()
        t.cg_iterate(4); t.sync(); t.reset_kernel_times()
        t.cg_iterate(30); t.sync()
        kt = t.kernel_times()
        res[v].append(kt["cg1"][0] / kt["cg1"][1])
for v in vals:
    print(f"FORM 0 {structure} {n}x{n}x{shape[0]}  {key} = {v:g}: " + " ".join(f"{x:.3f}" for x in res[v]) + f"  | median {sorted(res[v])[1]:.3f} ms", flush=True)
oia.set_option(key, vals[0])
t.destroy()
