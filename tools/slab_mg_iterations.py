"""Iteration counts of the multigrid-preconditioned CG when the cycle is slab-local
(block-Jacobi across z-slabs): one GPU, n host threads with the in-process transport.
usage: python tools/slab_mg_iterations.py [size] [direction]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import openimpala_amd as oia
from test_gpu_slabs import run_slabs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 0
buf = oia.gen_spheres_device((n, n, n), 20251031, 24 * n // 1024 if n >= 512 else 12, 0.4)
phase = buf.download(np.uint8, (n, n, n))
for precond, name in ((oia.PRECOND_MG, "mg"),):
    t0 = time.time()
    t = oia.TortuosityMI355(phase, 1, d, -1.0, 1.0, eps=1e-6, precond=precond)
    t.value()
    print(f"{n}^3 dir {d} single {name}: iterations {t.getSolverIterations()} solve {t.stats().t_solve_s:.3f}s", flush=True)
    t.destroy()
    for nr in (2, 4, 6):
        outs = run_slabs(phase, 1, d, nr, -1.0, 1.0, eps=1e-6, precond=precond)
        print(f"{n}^3 dir {d} {nr} slabs {name}: iterations {outs[0]['st']['iterations']} status {outs[0]['status']}", flush=True)
