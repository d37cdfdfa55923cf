"""What one sample of a REV study costs: homogenization handles on sub-volumes of n^3 created, solved (three cell problems + tensor) and
destroyed back to back, with and without the pool of released arrays.  python tools/rev_cost.py [sizes=64,96,128,192]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 96, 128, 192]
rng = np.random.default_rng(2)
for pool in (0, 1):
    oia.dev_pool(pool, 0.0)
    for n in sizes:
        subs = [(rng.random((n, n, n)) < 0.6).astype(np.uint8) for _ in range(4)]
        tc = ts = td = 0.0
        reps = 12
        its = 0
        for r in range(reps + 2):
            sub = subs[r % 4]
            t0 = time.perf_counter()
            h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-9, precond=oia.PRECOND_MG)
            t1 = time.perf_counter()
            res = [h.solve(k) for k in range(3)]
            D = h.tensor()
            t2 = time.perf_counter()
            h.destroy()
            t3 = time.perf_counter()
            if r >= 2:
                tc += t1 - t0; ts += t2 - t1; td += t3 - t2; its += sum(x[1] for x in res)
        print(f"pool {pool} n {n}: create {1e3 * tc / reps:.2f} ms, three solves + tensor {1e3 * ts / reps:.2f} ms ({its / reps:.0f} iterations, "
              f"{1e3 * ts / its:.3f} ms each), destroy {1e3 * td / reps:.2f} ms", flush=True)
oia.dev_pool(0, 0.0)
