"""Design experiment (round 4, CPU only): would SMOOTHED aggregation (the tentative prolongator smoothed by one damped-Jacobi step, Galerkin
coarse operators) beat the shipped plain-aggregation cycle?  PCG iteration counts to 1e-6 / 1e-9 and nonzeros per row of the first four
levels.  Output kept in profiles/r04/smoothed_aggregation_prototype.txt."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
import importlib.util
# reuse build / agg from the prototype without running its main part
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mg_prototype.py')).read().split("for rep in (1,2):")[0]
ns = {'__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mg_prototype.py')}
exec(compile(src, 'proto', 'exec'), ns)
build, agg, pcg = ns['build'], ns['agg'], ns['pcg']
o = ns['o']; ph0 = ns['ph0']

def hierarchy(A, coords, shape, smooth, minsize=600, omega_p=2.0/3.0):
    L = [dict(A=A, D=A.diagonal())]
    while A.shape[0] > minsize and len(L) < 10:
        P, coords, shape = agg(coords, shape)
        if smooth:
            Dinv = sp.diags(1.0 / A.diagonal())
            P = (P - omega_p * (Dinv @ (A @ P))).tocsr()
        A = (P.T @ A @ P).tocsr()
        L[-1]['P'] = P
        L.append(dict(A=A, D=A.diagonal()))
    return L

def cyc(L, l, b, omega, scale, wmask, ncoarse):
    lev = L[l]; A, D = lev['A'], lev['D']
    if 'P' not in lev:
        x = omega * b / D
        for _ in range(ncoarse - 1): x += omega * (b - A @ x) / D
        return x
    x = omega * b / D
    r = b - A @ x
    bc = lev['P'].T @ r
    xc = cyc(L, l + 1, bc, omega, scale, wmask, ncoarse)
    if (wmask >> (l + 1)) & 1:
        rc = bc - L[l + 1]['A'] @ xc
        xc = xc + cyc(L, l + 1, rc, omega, scale, wmask, ncoarse)
    x += scale * (lev['P'] @ xc)
    x += omega * (b - A @ x) / D
    return x

cases = [('sample p1 x', ph0, 1, 0, -1., 1.), ('spheres 128 r6', o.gen_spheres((128, 128, 128), 20251031, 6, 0.4), 1, 0, 0., 1.),
         ('blobs 128', o.gen_blobs((128, 128, 128), 7, 12, 23996), 1, 0, 0., 1.)]
for name, ph, pid, d, vlo, vhi in cases:
    A, b, coords, bn = build(ph, pid, d, vlo, vhi)
    out = []
    for label, smooth, scale, wmask in (("plain agg, scale 1.5, W on levels 1-3 (shipped)", False, 1.5, 14), ("smoothed agg, scale 1.0, V", True, 1.0, 0),
                                        ("smoothed agg, scale 1.0, W 1-3", True, 1.0, 14), ("smoothed agg, scale 1.2, V", True, 1.2, 0)):
        t = time.time()
        L = hierarchy(A, coords, ph.shape, smooth)
        nnz = [l['A'].nnz / l['A'].shape[0] for l in L]
        its = []
        for eps in (1e-6, 1e-9):
            x, it = pcg(A, b, bn, eps, lambda r: cyc(L, 0, r, 0.9, scale, wmask, 16))
            its.append(it)
        out.append((label, its, [round(v, 1) for v in nnz[:4]]))
    print(name, 'n', A.shape[0])
    for row in out: print('   ', row, flush=True)
