"""In-process A/B harness for the single-kernel CG iteration: one process, alternating batches (process-to-process
noise on these boxes is +-10 %, in-process batches repeat to 0.2 %).
Compares  dense layout with pair-granular predication | compacted lines, pair-granular | compacted lines, whole lines.
Earlier experiments run through it and rejected: halo code prefetch (no gain), look-ahead touch of plane gz+2 (-5 %)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
pf32 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
kw = dict(shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000, p_storage=pf32)
oia.set_cg1_layout(0)
td = oia.TortuosityMI355(buf, 1, 0, **kw)
oia.set_cg1_layout(1)
tc = oia.TortuosityMI355(buf, 1, 0, **kw)
for t in (td, tc):
    t.cg_begin(); t.cg_iterate(10); t.sync()
res = {"dense, pairs": [], "dense, whole lines": [], "compact, pairs": [], "compact, whole lines": []}
for rep in range(5):
    for name, t, mode in (("dense, pairs", td, 2), ("dense, whole lines", td, 0), ("compact, pairs", tc, 3), ("compact, whole lines", tc, 1)):
        oia.set_cg1_layout(mode)   # (layout bit only matters at creation; bit 1 switches the access granularity per launch)
        t.reset_kernel_times(); t.cg_iterate(40); t.sync()
        kt = t.kernel_times()
        res[name].append(kt["cg1"][0] / kt["cg1"][1])
oia.set_cg1_layout(1)
for name, v in res.items():
    print("%-22s" % name, " ".join("%.3f" % x for x in v), "| median %.3f ms" % sorted(v)[len(v) // 2], flush=True)
print("device bytes: dense %.1f GB, compact %.1f GB" % (td.stats().bytes_device / 1e9, tc.stats().bytes_device / 1e9))
