"""Design experiment (round 5, CPU only; usage: kcycle_prototype.py <n> <sphere radius> [rb]; output kept in profiles/r05/kcycle_rbgs_prototype.txt): does Krylov acceleration help the shipped plain-aggregation cycle?
 (a) flexible CG outside (Polak-Ribiere beta), (b) a K-cycle at the first coarse level(s): two inner FCG steps preconditioned by the rest of the cycle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np, scipy.sparse as sp
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mg_prototype.py')).read().split("for rep in (1,2):")[0]
ns = {'__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mg_prototype.py')}
exec(compile(src, 'proto', 'exec'), ns)
build, agg = ns['build'], ns['agg']
o = ns['o']

def hierarchy(A, coords, shape, minsize=600):
    L = [dict(A=A, D=A.diagonal())]
    while A.shape[0] > minsize and len(L) < 10:
        P, coords, shape = agg(coords, shape)
        A = (P.T @ A @ P).tocsr()
        L[-1]['P'] = P
        L.append(dict(A=A, D=A.diagonal()))
    return L

def cyc(L, l, b, omega, scale, wmask, ncoarse, kmask=0):
    lev = L[l]; A, D = lev['A'], lev['D']
    if 'P' not in lev:
        x = omega * b / D
        for _ in range(ncoarse - 1): x += omega * (b - A @ x) / D
        return x
    x = omega * b / D
    r = b - A @ x
    bc = lev['P'].T @ r
    Ac = L[l + 1]['A']
    if (kmask >> (l + 1)) & 1:
        # K-cycle on level l+1: two steps of flexible CG on Ac xc = bc, preconditioned by the cycle below (Notay)
        c1 = cyc(L, l + 1, bc, omega, scale, wmask, ncoarse, kmask)
        v1 = Ac @ c1
        rho1 = c1 @ v1; a1 = c1 @ bc
        xc = (a1 / rho1) * c1
        rc = bc - (a1 / rho1) * v1
        c2 = cyc(L, l + 1, rc, omega, scale, wmask, ncoarse, kmask)
        v2 = Ac @ c2
        g = c2 @ v1; bta = c2 @ v2; a2 = c2 @ rc
        rho2 = bta - g * g / rho1
        xc = (a1 / rho1 - g * a2 / (rho1 * rho2)) * c1 + (a2 / rho2) * c2
        s = 1.0   # the Krylov step has found its own step length
    else:
        xc = cyc(L, l + 1, bc, omega, scale, wmask, ncoarse, kmask)
        if (wmask >> (l + 1)) & 1:
            rc = bc - Ac @ xc
            xc = xc + cyc(L, l + 1, rc, omega, scale, wmask, ncoarse, kmask)
        s = scale
    x += s * (lev['P'] @ xc)
    x += omega * (b - A @ x) / D
    return x

def pcg(A, b, bn, eps, M, flexible=False, maxit=500):
    x = np.zeros_like(b); r = b.copy(); z = M(r); p = z.copy(); rz = r @ z; it = 0; hist = []
    while np.sqrt(r @ r) / bn > eps and it < maxit:
        q = A @ p; a = rz / (p @ q); x += a * p
        rold = r.copy()
        r -= a * q; z = M(r)
        rzn = r @ z
        beta = (rzn - rold @ z) / rz if flexible else rzn / rz
        p = z + beta * p; rz = rzn; it += 1
        hist.append(np.sqrt(r @ r) / bn)
    return x, it, hist

if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    rad = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    ph = o.gen_spheres((n, n, n), 20251031, rad, 0.4)
    A, b, coords, bn = build(ph, 1, 0, 0., 1.)
    L = hierarchy(A, coords, ph.shape)
    print('spheres', n, 'r', rad, 'n', A.shape[0], 'levels', [l['A'].shape[0] for l in L], flush=True)
    for label, kw, flex in (("shipped cycle, PCG", dict(wmask=14, kmask=0), False), ("shipped cycle, flexible CG", dict(wmask=14, kmask=0), True),
                            ("K-cycle on level 1, FCG", dict(wmask=14, kmask=2), True), ("K-cycle on levels 1-2, FCG", dict(wmask=14, kmask=6), True),
                            ("K-cycle on levels 1-3, FCG", dict(wmask=0, kmask=14), True), ("W on level 1 too (wmask 15), PCG", dict(wmask=15, kmask=0), False)):
        t = time.time()
        x, it, hist = pcg(A, b, bn, 1e-9, lambda r: cyc(L, 0, r, 0.9, 1.5, kw['wmask'], 16, kw['kmask']), flexible=flex)
        it6 = next(i + 1 for i, h in enumerate(hist) if h <= 1e-6)
        tail = hist[it6 - 1] / hist[max(0, it6 - 6)]
        print('  %-34s iterations to 1e-6 / 1e-9: %3d / %3d   mean factor over the last five before 1e-6: %.3f   (%.0f s)' % (label, it6, it, tail ** 0.2, time.time() - t), flush=True)

def rb_split(coords):
    i, j, k = coords
    return ((i + j + k) & 1) == 0

def cyc_rb(L, l, b, omega, scale, wmask, ncoarse, rbmask, levels_red):
    """the shipped cycle with a red-black Gauss-Seidel smoother (red then black before the correction, black then red after it) on the levels in rbmask"""
    lev = L[l]; A, D = lev['A'], lev['D']
    if 'P' not in lev:
        x = omega * b / D
        for _ in range(ncoarse - 1): x += omega * (b - A @ x) / D
        return x
    rb = (rbmask >> l) & 1
    if rb:
        red = levels_red[l]; blk = ~red
        x = np.zeros_like(b)
        x[red] = b[red] / D[red]
        x[blk] = (b - A @ x)[blk] / D[blk]
    else:
        x = omega * b / D
    r = b - A @ x
    bc = lev['P'].T @ r
    Ac = L[l + 1]['A']
    xc = cyc_rb(L, l + 1, bc, omega, scale, wmask, ncoarse, rbmask, levels_red)
    if (wmask >> (l + 1)) & 1:
        rc = bc - Ac @ xc
        xc = xc + cyc_rb(L, l + 1, rc, omega, scale, wmask, ncoarse, rbmask, levels_red)
    x += scale * (lev['P'] @ xc)
    if rb:
        red = levels_red[l]; blk = ~red
        x[blk] += (b - A @ x)[blk] / D[blk]
        x[red] += (b - A @ x)[red] / D[red]
    else:
        x += omega * (b - A @ x) / D
    return x

if __name__ == '__main__' and len(sys.argv) > 3:
    # red-black colouring per level (by the cells' coordinates)
    coords_l = [coords]; shp = ph.shape; c = coords
    for l in range(len(L) - 1):
        P_, c, shp = agg(c, shp)
        coords_l.append(c)
    reds = [rb_split(c) for c in coords_l]
    for label, rbmask, scale in (("RB-GS on the fine level only, scale 1.5", 1, 1.5), ("RB-GS on the fine level only, scale 1.3", 1, 1.3), ("RB-GS on every level, scale 1.5", 255, 1.5),
                                 ("RB-GS on every level, scale 1.3", 255, 1.3), ("RB-GS on the coarse levels only, scale 1.5", 254, 1.5)):
        t = time.time()
        x, it, hist = pcg(A, b, bn, 1e-9, lambda r: cyc_rb(L, 0, r, 0.9, scale, 14, 16, rbmask, reds))
        it6 = next(i + 1 for i, h in enumerate(hist) if h <= 1e-6)
        print('  %-44s iterations to 1e-6 / 1e-9: %3d / %3d   (%.0f s)' % (label, it6, it, time.time() - t), flush=True)
