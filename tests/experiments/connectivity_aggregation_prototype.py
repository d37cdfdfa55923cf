"""Design experiment (round 5, CPU only; usage: connectivity_aggregation_prototype.py <n> <sphere radius> [s|b]; output kept in
profiles/r05/connectivity_aggregation_prototype.txt): 2x2x2 aggregates SPLIT into their connected components (cells of one block that are not linked inside the block get their own coarse unknown)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import connected_components
import kcycle_prototype as kc
build, o, pcg, cyc = kc.build, kc.o, kc.pcg, kc.cyc

def agg_conn(A, coords, shape, split, drop=False):
    i, j, k = coords; ci, cj, ck = i // 2, j // 2, k // 2
    cs = ((shape[2] + 1) // 2, (shape[1] + 1) // 2, (shape[0] + 1) // 2)
    key = ci + cs[0] * (cj + cs[1] * ck)
    if split:
        C = A.tocoo()
        m = (C.row != C.col) & (key[C.row] == key[C.col]) & (C.data != 0)
        G = sp.coo_matrix((np.ones(m.sum()), (C.row[m], C.col[m])), shape=A.shape)
        ncomp, lab = connected_components(G, directed=False)
        if drop:
            # keep the LARGEST component of every block (ties: the one with the lowest label); the other cells get no coarse unknown
            size = np.bincount(lab)
            order = np.lexsort((lab, -size[lab], key))           # by block, then size descending, then label
            firstpos = np.ones(order.size, bool); firstpos[1:] = key[order][1:] != key[order][:-1]
            best = np.zeros(key.max() + 1, np.int64); best[key[order][firstpos]] = lab[order][firstpos]
            keep = lab == best[key]
            u, inv_k = np.unique(key[keep], return_inverse=True)
            rows = np.nonzero(keep)[0]
            P = sp.csr_matrix((np.ones(rows.size), (rows, inv_k)), shape=(key.size, u.size))
            ck2 = u // (cs[0] * cs[1]); r = u % (cs[0] * cs[1]); cj2 = r // cs[0]; ci2 = r % cs[0]
            print('   dropped cells', int((~keep).sum()), 'of', key.size, flush=True)
            return P, (ci2, cj2, ck2), (cs[2], cs[1], cs[0])
        u, inv = np.unique(lab, return_inverse=True)
        # coordinates of a component = its block's
        first = np.zeros(u.size, np.int64); first[inv] = np.arange(inv.size)
        P = sp.csr_matrix((np.ones(inv.size), (np.arange(inv.size), inv)), shape=(inv.size, u.size))
        return P, (ci[first], cj[first], ck[first]), (cs[2], cs[1], cs[0])
    u, inv = np.unique(key, return_inverse=True)
    P = sp.csr_matrix((np.ones(inv.size), (np.arange(inv.size), inv)), shape=(inv.size, u.size))
    ck2 = u // (cs[0] * cs[1]); r = u % (cs[0] * cs[1]); cj2 = r // cs[0]; ci2 = r % cs[0]
    return P, (ci2, cj2, ck2), (cs[2], cs[1], cs[0])

def hierarchy(A, coords, shape, split, minsize=600, drop=False, only_first=False):
    L = [dict(A=A, D=A.diagonal())]
    while A.shape[0] > minsize and len(L) < 10:
        P, coords, shape = agg_conn(A, coords, shape, split and (not only_first or len(L) == 1), drop)
        A = (P.T @ A @ P).tocsr()
        L[-1]['P'] = P
        L.append(dict(A=A, D=A.diagonal()))
    return L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rad = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else 's'
ph = o.gen_spheres((n, n, n), 20251031, rad, 0.4) if kind == 's' else o.gen_blobs((n, n, n), 7, 12, 23996)
A, b, coords, bn = build(ph, 1, 0, 0., 1.)
for split, drop, only_first in ((False, False, False), (True, False, False), (True, False, True), (True, True, True), (True, True, False)):
    L = hierarchy(A, coords, ph.shape, split, drop=drop, only_first=only_first)
    x, it, hist = pcg(A, b, bn, 1e-9, lambda r: cyc(L, 0, r, 0.9, 1.5, 14, 16, 0))
    it6 = next(i + 1 for i, h in enumerate(hist) if h <= 1e-6)
    print(kind, n, rad, ('split' + (' DROP minor components' if drop else '') + (' first level only' if only_first else ' every level')) if split else 'plain', 'levels', [l['A'].shape[0] for l in L], 'nnz/row', [round(l['A'].nnz / l['A'].shape[0], 1) for l in L[:5]], 'iterations', it6, it, flush=True)
