"""Eight in-process slabs (the driver's N = 8 decomposition, 64 planes each) against the single-domain solve:
default single-kernel CG and the default multigrid PCG.  usage: python tests/experiments/slabs8_check.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
from test_gpu_slabs import run_slabs

shape = (512, 256, 256)
buf = oia.gen_spheres_device(shape, 20251031, 10, 0.4)
ph = buf.download(np.uint8, shape); buf.free()
for name, kw in (("cg1", dict(precond=oia.PRECOND_NONE)), ("multigrid", dict(precond=oia.PRECOND_MG))):
    t0 = time.time()
    s = oia.TortuosityMI355(ph, 1, 2, eps=1e-8, **kw)
    tau1, it1 = s.value(), s.getSolverIterations(); s.destroy()
    t1 = time.time()
    outs = run_slabs(ph, 1, 2, 8, eps=1e-8, **kw)
    t2 = time.time()
    taus = {r["tau"] for r in outs}
    print(name, "single tau %.10f its %d (%.1f s) | 8 slabs tau %s its %d (%.1f s) spans %s" % (
        tau1, it1, t1 - t0, sorted(taus), outs[0]["st"]["iterations"], t2 - t1, [r["span"] for r in outs][:3]), flush=True)
    assert len(taus) == 1 and abs(taus.pop() - tau1) <= 1e-7 * tau1 and abs(outs[0]["st"]["iterations"] - it1) <= 3
print("ok")
