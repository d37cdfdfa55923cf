"""A/B on ONE multigrid-PCG handle of an option that oi_tort_retune takes over and cg_begin freezes: the bench's solve to 1e-6
(cg_begin, cg_iterate(4) + cg_state until converged, cg_finish), alternating values, then the solutions compared bit for bit.
(Written for `mg_xstream`, profiles/r04/mg_x_update_on_its_own_stream_discarded.txt; options retune PRESERVES per handle --
mg_xpair, mg_blocked, the cycle shape -- need tools/mg_two_handles_ab.py.)
usage: python tools/mg_solve_ab.py <option> [n=1024] [values=0,1,0,1,...]"""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import openimpala_amd as oia
KEY = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
values = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 0, 1, 0, 1, 0, 1]
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=1000)
buf.free()
digest = {}
for rep, v in enumerate([values[0]] + values):  # (the first solve warms the handle up and is not reported)
    oia.set_option(KEY, v)
    t.retune()
    t.reset_kernel_times()
    t.sync()
    t0 = time.perf_counter()
    t.cg_begin()
    done, its = False, 0
    while not done and its < 1000:
        t.cg_iterate(4)
        its, rel, done = t.cg_state()
    t.cg_finish()
    dt = time.perf_counter() - t0
    mk = t.kernel_times()
    g = lambda k: mk[k][0] / max(1, mk[k][1])
    if rep:
        print(f"{KEY} = {v}: {dt:.4f} s, {its} iterations, relres {rel:.4e} | FORM 1 {g('fused_a'):.3f} FORM 2 {g('update_r'):.3f} "
              f"cycle {g('mg_vcycle'):.3f} ms", flush=True)
    if n <= 512 or rep in (len(values) - 1, len(values)):
        digest.setdefault(v, set()).add(hashlib.sha256(t.solution().tobytes()).hexdigest()[:16])
print("solution digests by value:", {k: sorted(s) for k, s in digest.items()})
vals = set().union(*digest.values()) if digest else set()
print("bit-identical across values:", len(vals) == 1)
oia.set_option(KEY, values[0])
t.destroy()
