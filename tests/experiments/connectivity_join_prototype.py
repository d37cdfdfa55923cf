"""Design experiment (round 5, CPU only; usage: connectivity_join_prototype.py <n> <radius> [s|b|sample]; profiles/r05/connectivity_aggregation_prototype.txt), option B: the minor components of a 2x2x2 block JOIN the aggregate of a face-neighbour block they are linked to (no extra coarse
unknowns); couplings of the Galerkin operator that leave the 7-point pattern are cut (the edge removed from the coarse graph)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import connected_components
import kcycle_prototype as kc
build, o, pcg, cyc, agg = kc.build, kc.o, kc.pcg, kc.cyc, kc.agg

def first_level(A, coords, shape, mode):
    """mode: 'plain' | 'split' (extra unknowns) | 'join' (option B, full Galerkin) | 'join7' (option B, 7-point truncation)"""
    i, j, k = coords; ci, cj, ck = i // 2, j // 2, k // 2
    cs = ((shape[2] + 1) // 2, (shape[1] + 1) // 2, (shape[0] + 1) // 2)
    key = ci + cs[0] * (cj + cs[1] * ck)
    C = A.tocoo()
    off = (C.row != C.col) & (C.data != 0)
    m = off & (key[C.row] == key[C.col])
    G = sp.coo_matrix((np.ones(m.sum()), (C.row[m], C.col[m])), shape=A.shape)
    ncomp, lab = connected_components(G, directed=False)
    size = np.bincount(lab)
    order = np.lexsort((lab, -size[lab], key))
    firstpos = np.ones(order.size, bool); firstpos[1:] = key[order][1:] != key[order][:-1]
    best = np.full(key.max() + 1, -1, np.int64); best[key[order][firstpos]] = lab[order][firstpos]
    primary = lab == best[key]
    akey = key.copy()                      # aggregate (block key) of every cell
    stats = {}
    if mode in ('join', 'join7'):
        # links from minor cells to PRIMARY cells of other blocks
        r, c = C.row[off], C.col[off]
        sel = (~primary[r]) & primary[c] & (key[r] != key[c])
        comp = lab[r[sel]]; tgt = key[c[sel]]
        # per minor component: the neighbour block with the most links
        pair = comp.astype(np.int64) * (key.max() + 1) + tgt
        up, cnt = np.unique(pair, return_counts=True)
        ucomp = up // (key.max() + 1); utgt = up % (key.max() + 1)
        o2 = np.lexsort((-cnt, ucomp))
        fp = np.ones(o2.size, bool); fp[1:] = ucomp[o2][1:] != ucomp[o2][:-1]
        choice = np.full(ncomp, -1, np.int64); choice[ucomp[o2][fp]] = utgt[o2][fp]
        minor = ~primary
        newk = choice[lab]
        ok = minor & (newk >= 0)
        akey[ok] = newk[ok]
        stats['minor cells'] = int(minor.sum()); stats['joined'] = int(ok.sum()); stats['left in their block'] = int((minor & ~ok).sum())
    if mode == 'split':
        u, inv = np.unique(lab, return_inverse=True)
        firstc = np.zeros(u.size, np.int64); firstc[inv] = np.arange(inv.size)
        P = sp.csr_matrix((np.ones(inv.size), (np.arange(inv.size), inv)), shape=(inv.size, u.size))
        return P, (P.T @ A @ P).tocsr(), (ci[firstc], cj[firstc], ck[firstc]), (cs[2], cs[1], cs[0]), stats
    u, inv = np.unique(akey, return_inverse=True)
    P = sp.csr_matrix((np.ones(inv.size), (np.arange(inv.size), inv)), shape=(inv.size, u.size))
    ck2 = u // (cs[0] * cs[1]); rr = u % (cs[0] * cs[1]); cj2 = rr // cs[0]; ci2 = rr % cs[0]
    Ac = (P.T @ A @ P).tocoo()
    if mode == 'join7':
        d = np.abs(ci2[Ac.row] - ci2[Ac.col]) + np.abs(cj2[Ac.row] - cj2[Ac.col]) + np.abs(ck2[Ac.row] - ck2[Ac.col])
        far = d > 1
        stats['couplings cut'] = int(far.sum() // 2); stats['of'] = int((Ac.row != Ac.col).sum() // 2)
        diag_fix = np.zeros(u.size); np.add.at(diag_fix, Ac.row[far], Ac.data[far])   # row I loses the edge to J: diag += (negative) offdiag
        keep = ~far
        Ac = sp.coo_matrix((Ac.data[keep], (Ac.row[keep], Ac.col[keep])), shape=Ac.shape).tocsr() + sp.diags(diag_fix)
    return P, Ac.tocsr(), (ci2, cj2, ck2), (cs[2], cs[1], cs[0]), stats

def hierarchy(A, coords, shape, mode, minsize=600):
    L = [dict(A=A, D=A.diagonal())]
    P, A1, coords, shape, stats = first_level(A, coords, shape, mode)
    L[-1]['P'] = P; L.append(dict(A=A1, D=A1.diagonal())); A = A1
    while A.shape[0] > minsize and len(L) < 10:
        P, coords, shape = agg(coords, shape)
        A = (P.T @ A @ P).tocsr()
        L[-1]['P'] = P
        L.append(dict(A=A, D=A.diagonal()))
    return L, stats

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rad = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else 's'
ph = o.gen_spheres((n, n, n), 20251031, rad, 0.4) if kind == 's' else (kc.ns['ph0'] if kind == 'sample' else o.gen_blobs((n, n, n), 7, 12, 23996))
A, b, coords, bn = build(ph, 1, 0, -1. if kind == 'sample' else 0., 1.)
for mode in ('plain', 'split', 'join', 'join7'):
    L, stats = hierarchy(A, coords, ph.shape, mode)
    if any((l['D'] <= 0).any() for l in L):
        print(kind, n, rad, '%-6s' % mode, 'SINGULAR: cutting the far couplings isolates', int(sum((l['D'] <= 0).sum() for l in L)), 'coarse cells', stats, flush=True)
        continue
    x, it, hist = pcg(A, b, bn, 1e-9, lambda r: cyc(L, 0, r, 0.9, 1.5, 14, 16, 0))
    it6 = next(i + 1 for i, h in enumerate(hist) if h <= 1e-6)
    print(kind, n, rad, '%-6s' % mode, 'level sizes', [l['A'].shape[0] for l in L][:4], 'iterations', it6, it, stats, flush=True)
