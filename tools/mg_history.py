import sys
sys.path.insert(0, "/root/repo")
import openimpala_amd as oia
n = 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-12, precond=oia.PRECOND_MG, maxiter=1000)
t.cg_begin()
prev = None
for k in range(1, 34):
    t.cg_iterate(1); t.sync()
    its, rel, done = t.cg_state()
    print(k, its, "%.4e" % rel, ("ratio %.3f" % (rel / prev)) if prev else "", flush=True)
    prev = rel
t.cg_finish(); t.destroy()
