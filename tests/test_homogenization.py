"""Homogenization path (the reference's default calculation_method).

The reference pins no number for it (tests/inputs/tEffectiveDiffusivity.inputs only asks for
convergence, symmetry to 1e-7 and non-negative diagonals), so parity here is against the
oracle's restatement of effdiff_fillmtx + the Diffusion.cpp tensor integrator (parity
unpinned beyond those invariants)."""
import numpy as np
import pytest

from oracle import oracle as o


def test_oracle_rows_restate_effdiff_fillmtx(sample_phase):
    sub = np.ascontiguousarray(sample_phase[:14, :12, :16])
    for k in range(3):
        a, rhs = o.effdiff_fillmtx(sub, 1, k)
        act = sub.reshape(-1) == 1
        assert np.all(a[act, 0] == 6.0) and np.all(a[~act, 0] == 1.0) and np.all(a[~act, 1:] == 0.0)
        assert np.all(rhs[~act] == 0.0) and set(np.unique(rhs)) <= {-0.5, 0.0, 0.5}
        # symmetric: -1 towards a neighbour iff both cells are active
        nz, ny, nx = sub.shape
        idx = np.arange(sub.size).reshape(sub.shape)
        for s, (ax, sh) in enumerate([(2, 1), (2, -1), (1, 1), (1, -1), (0, 1), (0, -1)]):
            nb = np.roll(idx, sh, axis=ax).reshape(-1)      # periodic neighbour (-x,+x,-y,+y,-z,+z)
            assert np.array_equal(a[:, 1 + s] == -1.0, act & act[nb])
        # the matrix-free CG solves exactly these rows
        chi, ok, it, rel = o.effdiff_solve(sub, 1, k, eps=1e-12)
        assert ok
        x = chi.reshape(-1)
        res = a[:, 0] * x - rhs
        for s, (ax, sh) in enumerate([(2, 1), (2, -1), (1, 1), (1, -1), (0, 1), (0, -1)]):
            res += a[:, 1 + s] * x[np.roll(idx, sh, axis=ax).reshape(-1)]
        assert np.abs(res).max() < 1e-10


def test_oracle_tensor_invariants(sample_phase):
    D, its = o.effective_diffusivity(np.ones((8, 10, 12), np.uint8), 1)
    np.testing.assert_allclose(D, np.eye(3), atol=1e-14)       # no interfaces: chi = 0, D = I
    D, its = o.effective_diffusivity(np.ascontiguousarray(sample_phase[:40, :40, :40]), 1, eps=1e-10)
    assert np.abs(D - D.T).max() < 1e-7 and np.all(np.diag(D) >= 0)   # tEffectiveDiffusivity's checks
    lam = np.zeros((8, 8, 8), np.uint8)
    lam[:, ::2, :] = 1                                                 # layers normal to y
    D, _ = o.effective_diffusivity(lam, 1)
    assert D[0, 0] == pytest.approx(0.5) and D[2, 2] == pytest.approx(0.5)  # along the layers: chi = 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape,pid", [((40, 36, 44), 1), ((24, 30, 33), 0), ((16, 16, 130), 1)])
def test_gpu_homogenization_matches_oracle(sample_phase, shape, pid):
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 1, 2))[:nz, :ny, :nx])
    want, its = o.effective_diffusivity(sub, pid, eps=1e-10)
    h = oia.EffectiveDiffusivityMI355(sub, pid, eps=1e-10, precond=oia.PRECOND_NONE)   # plain CG: the oracle's iteration counts
    for k in range(3):
        conv, it, rel = h.solve(k)
        assert conv and rel <= 1e-10 and abs(it - its[k]) <= 2
        chi, ok, _, _ = o.effdiff_solve(sub, pid, k, eps=1e-10)
        np.testing.assert_allclose(h.getChiSolution(k), chi, rtol=0, atol=1e-9)
    D = h.tensor()
    np.testing.assert_allclose(D, want, rtol=0, atol=1e-9)
    assert np.abs(D - D.T).max() < 1e-7 and np.all(np.diag(D) >= 0)
    h.destroy()


@pytest.mark.gpu
def test_gpu_homogenization_edge_cases():
    import openimpala_amd as oia
    h = oia.EffectiveDiffusivityMI355(np.ones((6, 8, 10), np.uint8), 1)
    assert all(h.solve(k)[0] for k in range(3))
    np.testing.assert_allclose(h.tensor(), np.eye(3), atol=1e-14)
    h = oia.EffectiveDiffusivityMI355(np.zeros((6, 8, 10), np.uint8), 1)   # no active cell
    assert all(h.solve(k) == (True, 0, 0.0) for k in range(3))
    np.testing.assert_allclose(h.tensor(), np.zeros((3, 3)), atol=0)
    s = o.gen_spheres((32, 32, 32), 3, 5, 0.5)
    h = oia.EffectiveDiffusivityMI355(s, 1, maxiter=2)                     # not converged -> chi = 0
    conv, it, rel = h.solve(0)
    assert not conv and it == 2 and np.all(h.getChiSolution(0) == 0.0)
    with pytest.raises(oia.OpenImpalaError):
        h.tensor()                                                          # chi_y, chi_z missing


@pytest.mark.gpu
@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_gpu_homogenization_on_a_ring_of_slabs(sample_phase, nranks):
    """z-slabs over a ring of ranks (in-process transport: host threads on one GPU): periodic z crosses
    the first/last rank, every rank reports the global tensor, chi slabs concatenate to the single-domain field."""
    import threading
    import openimpala_amd as oia
    sub = np.ascontiguousarray(sample_phase[:37, :28, :45])
    nz = sub.shape[0]
    want, its = o.effective_diffusivity(sub, 1, eps=1e-10)
    comms = oia.Comm.create_local(nranks)
    out, err = [None] * nranks, []

    def work(r):
        try:
            z0, z1 = oia.slab_range(nz, nranks, r)
            h = oia.EffectiveDiffusivityMI355(np.ascontiguousarray(sub[z0:z1]), 1, eps=1e-10, precond=oia.PRECOND_NONE, comm=comms[r], z0=z0,
                                              nz_global=nz)
            res = [h.solve(k) for k in range(3)]
            out[r] = dict(res=res, D=h.tensor(), chi=[h.getChiSolution(k) for k in range(3)])
            h.destroy()
        except Exception as e:
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th), "slab threads hung"
    assert not err, err
    for c in comms:
        c.destroy()
    for r in range(nranks):
        assert all(conv for conv, _, _ in out[r]["res"])
        assert [it for _, it, _ in out[r]["res"]] == [it for _, it, _ in out[0]["res"]]
        assert np.array_equal(out[r]["D"], out[0]["D"])
    assert all(abs(it - its[k]) <= 2 for k, (_, it, _) in enumerate(out[0]["res"]))
    np.testing.assert_allclose(out[0]["D"], want, rtol=0, atol=1e-9)
    for k in range(3):
        chi, ok, _, _ = o.effdiff_solve(sub, 1, k, eps=1e-10)
        np.testing.assert_allclose(np.concatenate([out[r]["chi"][k] for r in range(nranks)]), chi, rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,pid", [((40, 36, 44), 1), ((25, 31, 33), 0), ((16, 18, 130), 1), ((64, 64, 64), 1), ((12, 40, 300), 1),
                                       ((25, 31, 36), 0), ((17, 40, 132), 1), ((40, 19, 44), 1)])
def test_gpu_homogenization_multigrid(sample_phase, shape, pid):
    """OI_PRECOND_MG on the periodic cell problems: the aggregation hierarchy wraps in x, y (by index) and z (ghost
    planes), odd sizes leave single-child aggregates next to the wrap.  Same chi and tensor, far fewer iterations.
    Even extents run the flow-through path's LDS-tiled kernels and fused two-kernel iteration with x / y wrapped by
    index (domain ends inside a tile: 18 and 40 rows over 16-row tiles, 130 and 300 columns over 128-column tiles), also
    with odd extents (round 3: odd ny / nz -- the last three shapes -- and odd nx, (25, 31, 33))."""
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 1, 2))[:nz, :ny, :nx])
    want, its = o.effective_diffusivity(sub, pid, eps=1e-10)
    h = oia.EffectiveDiffusivityMI355(sub, pid, eps=1e-10, precond=oia.PRECOND_MG)
    # (round 4: all of them run on compacted vectors with the rotated iteration -- the wrapped row / column staged inside a tile for
    # 36, 40, 18, 31 and 19 rows, 44, 300, 130, 36, 132 and 33 columns; odd ny / nz: (25, 31, 36), (17, 40, 132), (40, 19, 44); odd nx,
    # the pair (nx - 1, pad) with the wrapped column 0 in its second LDS slot: (25, 31, 33))
    assert h.layout()[0]
    for k in range(3):
        conv, it, rel = h.solve(k)
        assert conv and rel <= 1e-10 and it < max(12, its[k] // 2), (k, it, its[k])
        chi, ok, _, _ = o.effdiff_solve(sub, pid, k, eps=1e-10)
        np.testing.assert_allclose(h.getChiSolution(k), chi, rtol=0, atol=2e-9)
    np.testing.assert_allclose(h.tensor(), want, rtol=0, atol=2e-9)
    h.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 64), (40, 36, 44), (25, 31, 33)])
def test_gpu_homogenization_coarsest_level_in_one_launch(sample_phase, oi_option, shape):
    """Round 4: the periodic coarsest level runs its sixteen sweeps in ONE launch too (k_mg_coarsest with x, y, z neighbours wrapped
    by index inside the LDS iterate) instead of sixteen launches with two image-plane copies each: same arithmetic in the same
    order -- the solves are bit-identical with `mg_coarsest_plain` = 1 (even and odd extents, single-cell wraps)."""
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 1, 2))[:nz, :ny, :nx])

    def run():
        h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG)
        res = [h.solve(k) for k in range(3)]
        out = res, [h.getChiSolution(k).copy() for k in range(3)], h.tensor().copy()
        h.destroy()
        return out

    one = run()
    oi_option("mg_coarsest_plain", 1)
    many = run()
    for k in range(3):
        assert one[0][k] == many[0][k] and one[0][k][0]
        assert np.array_equal(one[1][k], many[1][k])
    assert np.array_equal(one[2], many[2])


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,nzs", [(2, 48), (3, 37), (4, 64)])
def test_gpu_homogenization_multigrid_on_a_ring(sample_phase, oi_option, nranks, nzs):
    """The same on a ring of z-slabs: slab-partitioned periodic levels (ring exchange of their ghost planes) and the
    replicated tail; same iteration counts as on one GPU."""
    import threading
    import openimpala_amd as oia
    oi_option("mg_replicate_cells", 600)
    sub = np.ascontiguousarray(sample_phase[:nzs, :28, :44])
    want, _ = o.effective_diffusivity(sub, 1, eps=1e-10)
    single = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG)
    its1 = [single.solve(k)[1] for k in range(3)]
    single.destroy()
    comms = oia.Comm.create_local(nranks)
    out, err = [None] * nranks, []

    def work(r):
        try:
            z0, z1 = oia.slab_range(nzs, nranks, r)
            h = oia.EffectiveDiffusivityMI355(np.ascontiguousarray(sub[z0:z1]), 1, eps=1e-10, precond=oia.PRECOND_MG,
                                              comm=comms[r], z0=z0, nz_global=nzs)
            res = [h.solve(k) for k in range(3)]
            out[r] = dict(res=res, D=h.tensor(), compacted=h.layout()[0])
            h.destroy()
        except Exception as e:
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th), "slab threads hung"
    assert not err, err
    for c in comms:
        c.destroy()
    assert all(conv for conv, _, _ in out[0]["res"])
    assert all(abs(it - i1) <= 2 for (_, it, _), i1 in zip(out[0]["res"], its1)), ([r[1] for r in out[0]["res"]], its1)
    np.testing.assert_allclose(out[0]["D"], want, rtol=0, atol=2e-9)
    # (round 4: a ring whose slabs all start on an even plane runs the compacted vectors and the rotated iteration too -- the line
    # lists of the first / last plane travel round the ring, exchange_ring_v -- the ring of 13 + 12 + 12 planes keeps dense vectors;
    # every rank takes the same decision)
    assert [o_["compacted"] for o_ in out] == [nranks != 3] * nranks


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(36, 52, 172), (35, 51, 132), (33, 40, 20), (40, 37, 64), (36, 52, 171), (33, 41, 67), (20, 24, 129), (16, 16, 127)])
def test_gpu_homogenization_tiled_and_plain_kernels_agree(sample_phase, oi_option, shape):
    """The two kernel families behind OI_PRECOND_MG on the cell problems (OI_MG_PLAIN selects the plain one): same
    iteration counts, chi fields equal to solver tolerance.  Round 3: odd extents run the tiled kernels too: a single column /
    row / plane in the last aggregates, image planes beyond an odd nz, and with an odd nx the x pair (nx - 1, pad) whose second
    LDS slot holds the wrapped column 0 (domain end inside a tile: 171, 67; one voxel past / before a tile boundary: 129, 127)."""
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 1, 2))[:nz, :ny, :nx])

    def run():
        h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-11, precond=oia.PRECOND_MG)
        res = [h.solve(k) for k in range(3)]
        out = res, [h.getChiSolution(k) for k in range(3)], h.tensor().copy()
        h.destroy()
        return out

    tiled = run()
    oi_option("mg_plain", 1)
    plain = run()
    for k in range(3):
        assert tiled[0][k][0] and plain[0][k][0] and abs(tiled[0][k][1] - plain[0][k][1]) <= 1
        np.testing.assert_allclose(tiled[1][k], plain[1][k], rtol=0, atol=1e-9)
    np.testing.assert_allclose(tiled[2], plain[2], rtol=0, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,pid", [((12, 32, 128), 1), ((20, 16, 256), 0), ((6, 48, 128), 1), ((34, 64, 384), 1), ((10, 30, 136), 1),
                                       ((8, 34, 120), 0), ((4, 4, 8), 1), ((22, 50, 260), 1), ((9, 33, 128), 1), ((33, 17, 24), 0), ((12, 20, 129), 1), ((10, 16, 127), 0), ((7, 9, 11), 1)])
def test_gpu_homogenization_on_compacted_vectors_matches_oracle(sample_phase, shape, pid):
    """Round 4: on one GPU the multigrid-preconditioned cell problems take the flow-through path's compacted vectors and rotated multigrid iteration
    (k_cg1b FORM 2 -> coarse levels -> FORM 3 -> FORM 1, their periodic instantiations) with the halo of the first row / column
    wrapped to the last one inside the kernel and z wrapped through image copies of the line lists.  Same chi and tensor as the
    oracle: planes of whole 128 x 16 tiles (one tile column / one tile row / several of each), the domain ending inside a tile (one
    line / two rows past a tile boundary: 136, 34; short of it: 120, 30; 260 x 50), a volume smaller than a tile, odd ny and nz (a single
    row / plane in the last aggregates, the image planes' parents in the coarse ghost planes), odd nx one voxel past / before a tile
    boundary (the voxel pair that straddles the seam)."""
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 1, 4))[:nz, :ny, :nx])
    want, its = o.effective_diffusivity(sub, pid, eps=1e-10)
    h = oia.EffectiveDiffusivityMI355(sub, pid, eps=1e-10, precond=oia.PRECOND_MG)
    compacted, nel = h.layout()
    assert compacted and nel < (nz + 2) * ny * nx
    for k in range(3):
        conv, it, rel = h.solve(k)
        assert conv and rel <= 1e-10 and it < max(12, its[k] // 2), (k, it, its[k])
        chi, ok, _, _ = o.effdiff_solve(sub, pid, k, eps=1e-10)
        np.testing.assert_allclose(h.getChiSolution(k), chi, rtol=0, atol=2e-9)
    np.testing.assert_allclose(h.tensor(), want, rtol=0, atol=2e-9)
    h.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 64, 256), (18, 32, 128), (64, 128, 128), (30, 90, 200)])
def test_gpu_homogenization_compacted_and_dense_vectors_agree(sample_phase, oi_option, shape):
    """The same volumes on dense vectors with the unrotated iteration (`mg_norot` = 1, what every other shape runs): same iteration
    counts to within one, chi equal to solver tolerance, and a solve that is repeated on one handle gives the same bits (every
    buffer of the rotated iteration is re-initialised by solve())."""
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 2, 3))[:nz, :ny, :nx])

    def run(expect):
        h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-11, precond=oia.PRECOND_MG)
        assert h.layout()[0] == expect
        res = [h.solve(k) for k in range(3)]
        chis = [h.getChiSolution(k) for k in range(3)]
        again = h.solve(1)
        assert again == res[1] and np.array_equal(h.getChiSolution(1), chis[1])
        out = res, chis, h.tensor().copy()
        h.destroy()
        return out

    lines = run(True)
    oi_option("mg_norot", 1)
    dense = run(False)
    for k in range(3):
        assert lines[0][k][0] and dense[0][k][0] and abs(lines[0][k][1] - dense[0][k][1]) <= 1
        np.testing.assert_allclose(lines[1][k], dense[1][k], rtol=0, atol=1e-9)
    np.testing.assert_allclose(lines[2], dense[2], rtol=0, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tiled", [((64, 64, 64), 1), ((40, 36, 44), 1), ((25, 31, 33), 1), ((48, 96, 384), 2), ((30, 50, 520), 2), ((2, 6, 8), 1)])
def test_gpu_homogenization_kernels_write_their_own_z_images(sample_phase, oi_option, shape, tiled):
    """Round 4: on a periodic level that lives whole on one rank every kernel of the cycle writes, with a cell of the first / last
    plane, its image in the ghost plane beyond the other end -- the image-plane copy launches after them (78 per cycle at 1024^3)
    are gone.  The copies moved, nothing else: with `mg_fused_images` = 0 (the copies run as well) the solves are the same bit for
    bit; one-cell, four-cell and LDS-tiled kernels (`mg_coarse_tiled` = 2), one and two sweeps per visit, a level of one plane."""
    import openimpala_amd as oia
    nz, ny, nx = shape
    sub = np.ascontiguousarray(np.tile(sample_phase, (1, 1, 6))[:nz, :ny, :nx])
    oi_option("mg_coarse_tiled", tiled, 1)
    out = {}
    for nu in (1, 2):
        oi_option("mg_nu_coarse", nu)
        for v in (1, 0):
            oi_option("mg_fused_images", v, 1)
            h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG)
            res = [h.solve(k) for k in range(3)]
            out[v] = res, [h.getChiSolution(k).copy() for k in range(3)], h.tensor().copy()
            h.destroy()
        assert out[1][0] == out[0][0] and all(r[0] for r in out[1][0])
        for k in range(3):
            assert np.array_equal(out[1][1][k], out[0][1][k])
        assert np.array_equal(out[1][2], out[0][2])

