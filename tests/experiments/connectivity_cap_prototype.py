"""Design experiment (round 5, CPU only; usage: connectivity_cap_prototype.py <n> <sphere radius> [s|b]; profiles/r05/connectivity_aggregation_prototype.txt):
how many cells per block does the split need?  2x2x2 aggregates SPLIT into their connected components (cells of one block that are not linked inside the block get their own coarse unknown)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
from scipy.sparse.csgraph import connected_components
import kcycle_prototype as kc
build, o, pcg, cyc = kc.build, kc.o, kc.pcg, kc.cyc

def agg_conn(A, coords, shape, split, cap=0):
    i, j, k = coords; ci, cj, ck = i // 2, j // 2, k // 2
    cs = ((shape[2] + 1) // 2, (shape[1] + 1) // 2, (shape[0] + 1) // 2)
    key = ci + cs[0] * (cj + cs[1] * ck)
    if split:
        C = A.tocoo()
        m = (C.row != C.col) & (key[C.row] == key[C.col]) & (C.data != 0)
        G = sp.coo_matrix((np.ones(m.sum()), (C.row[m], C.col[m])), shape=A.shape)
        ncomp, lab = connected_components(G, directed=False)
        if cap:
            # at most `cap` cells per block: the cap - 1 largest components keep their own cell, the rest share the last one
            size = np.bincount(lab)
            order = np.lexsort((lab, -size[lab], key))
            ko = key[order]; lo = lab[order]
            newblock = np.ones(order.size, bool); newblock[1:] = ko[1:] != ko[:-1]
            newcomp = np.ones(order.size, bool); newcomp[1:] = (lo[1:] != lo[:-1]) | newblock[1:]
            rank = np.cumsum(newcomp) - 1
            first_rank_of_block = np.maximum.accumulate(np.where(newblock, rank, 0))
            r_in_block = np.minimum(rank - first_rank_of_block, cap - 1)
            lab2 = np.empty_like(lab); lab2[order] = ko.astype(np.int64) * cap + r_in_block
            lab = lab2
        u, inv = np.unique(lab, return_inverse=True)
        # coordinates of a component = its block's
        first = np.zeros(u.size, np.int64); first[inv] = np.arange(inv.size)
        P = sp.csr_matrix((np.ones(inv.size), (np.arange(inv.size), inv)), shape=(inv.size, u.size))
        return P, (ci[first], cj[first], ck[first]), (cs[2], cs[1], cs[0])
    u, inv = np.unique(key, return_inverse=True)
    P = sp.csr_matrix((np.ones(inv.size), (np.arange(inv.size), inv)), shape=(inv.size, u.size))
    ck2 = u // (cs[0] * cs[1]); r = u % (cs[0] * cs[1]); cj2 = r // cs[0]; ci2 = r % cs[0]
    return P, (ci2, cj2, ck2), (cs[2], cs[1], cs[0])

def hierarchy(A, coords, shape, split, minsize=600, cap=0):
    L = [dict(A=A, D=A.diagonal())]
    while A.shape[0] > minsize and len(L) < 10:
        P, coords, shape = agg_conn(A, coords, shape, split, cap)
        A = (P.T @ A @ P).tocsr()
        L[-1]['P'] = P
        L.append(dict(A=A, D=A.diagonal()))
    return L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rad = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kind = sys.argv[3] if len(sys.argv) > 3 else 's'
ph = o.gen_spheres((n, n, n), 20251031, rad, 0.4) if kind == 's' else o.gen_blobs((n, n, n), 7, 12, 23996)
A, b, coords, bn = build(ph, 1, 0, 0., 1.)
for split, cap in ((False, 0), (True, 0), (True, 2), (True, 3), (True, 4)):
    L = hierarchy(A, coords, ph.shape, split, cap=cap)
    x, it, hist = pcg(A, b, bn, 1e-9, lambda r: cyc(L, 0, r, 0.9, 1.5, 14, 16, 0))
    it6 = next(i + 1 for i, h in enumerate(hist) if h <= 1e-6)
    print(kind, n, rad, ('split, at most %d cells per block' % cap if cap else 'split, unbounded') if split else 'plain', 'levels', [l['A'].shape[0] for l in L], 'nnz/row', [round(l['A'].nnz / l['A'].shape[0], 1) for l in L[:5]], 'iterations', it6, it, flush=True)
