"""In-process A/B harness for cg1 kernel experiments on ONE handle (alternating batches; process-to-process noise on
these boxes is +-10 %, in-process batches repeat to 0.2 %).  Last used for a look-ahead touch of plane gz+2 (a
fire-and-forget dword load per line during the compute phase): 9.10 ms with, 8.67 ms without -- rejected."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24 if n >= 512 else 8, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
t.cg_begin(); t.cg_iterate(10); t.sync()
res = {1: [], 3: []}
for rep in range(6):
    for mode in (1, 3):
        oia.set_cg1_layout(mode)
        t.reset_kernel_times(); t.cg_iterate(40); t.sync()
        kt = t.kernel_times()
        res[mode].append(kt["cg1"][0] / kt["cg1"][1])
oia.set_cg1_layout(1)
for mode, name in ((1, "touch on "), (3, "touch off")):
    v = res[mode]
    print(name, " ".join("%.3f" % x for x in v), "| median %.3f ms" % sorted(v)[len(v) // 2], flush=True)
