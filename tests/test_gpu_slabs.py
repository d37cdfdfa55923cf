"""Slab-decomposed solves on ONE GPU: `n` host threads, each owning a z-slab and an
in-process communicator (device-to-device ghost-plane copies, fixed-order reductions).
Everything except the RCCL calls themselves is the code the multi-GPU run executes."""
import threading

import numpy as np
import pytest

import openimpala_amd as oia
from oracle import oracle as o

pytestmark = pytest.mark.gpu


def run_slabs(phase, phase_id, direction, nranks, vlo=0.0, vhi=1.0, **opts):
    nz = phase.shape[0]
    comms = oia.Comm.create_local(nranks)
    out = [None] * nranks
    err = []

    def work(r):
        try:
            z0, z1 = oia.slab_range(nz, nranks, r)
            t = oia.TortuosityMI355(np.ascontiguousarray(phase[z0:z1]), phase_id, direction, vlo, vhi,
                                    comm=comms[r], z0=z0, nz_global=nz, **opts)
            ok = t.checkMatrixProperties()
            tau = t.value()
            st = t.stats()
            out[r] = dict(tau=tau, status=t.status, ok=ok, st=st.asdict(), mask=t.mask(),
                          x=t.solution() if st.n_active else None, span=(z0, z1))
            t.destroy()
        except Exception as e:  # surface the failure in the main thread
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th), "slab threads hung"
    assert not err, err
    for c in comms:
        c.destroy()
    return out


@pytest.mark.parametrize("nranks", [2, 3, 4])
@pytest.mark.parametrize("direction", [0, 1, 2])
def test_slabs_match_single_domain(nranks, direction):
    phase = o.gen_spheres((36, 40, 44), 21, 5, 0.45)
    ref = o.tortuosity(phase, 1, direction, -1.0, 1.0, eps=1e-11)
    single = oia.TortuosityMI355(phase, 1, direction, -1.0, 1.0, eps=1e-11)
    tau1 = single.value()
    it1 = single.getSolverIterations()
    outs = run_slabs(phase, 1, direction, nranks, -1.0, 1.0, eps=1e-11)
    mask = np.concatenate([r["mask"] for r in outs])
    assert np.array_equal(mask, ref["mask"])
    for r in outs:
        assert r["ok"] and r["status"] == oia.OI_OK
        assert r["st"]["n_active"] == ref["n_active"] and r["st"]["n_free"] == ref["n_free"]
        assert r["tau"] == outs[0]["tau"]  # every rank reports the same global answer
        assert abs(r["st"]["iterations"] - it1) <= 2
    assert outs[0]["tau"] == pytest.approx(ref["tau"], rel=1e-8)
    assert outs[0]["tau"] == pytest.approx(tau1, rel=1e-9)
    x = np.concatenate([r["x"] for r in outs])
    np.testing.assert_allclose(x, ref["x"], rtol=0, atol=1e-8)


@pytest.mark.parametrize("variant,jacobi,pf32", [(oia.KERNEL_UNFUSED, False, 0), (oia.KERNEL_FUSED, True, 0),
                                                 (oia.KERNEL_FUSED, False, 1), (oia.KERNEL_FUSED1, False, 0),
                                                 (oia.KERNEL_FUSED1, True, 1), (oia.KERNEL_LEAN, True, 0)])
def test_slab_variants_on_sample(sample_phase, goldens, variant, jacobi, pf32):
    g = goldens["phases"]["1"]
    sub = np.ascontiguousarray(sample_phase)
    outs = run_slabs(sub, 1, 2, 4, g["vlo"], g["vhi"], eps=1e-10, kernel=variant, p_storage=pf32,
                     precond=oia.PRECOND_JACOBI if jacobi else oia.PRECOND_NONE)
    assert outs[0]["tau"] == pytest.approx(g["tau"][2], rel=1e-7)
    assert outs[0]["st"]["flux_in"] == pytest.approx(g["flux"][2], rel=1e-7)
    assert outs[0]["st"]["active_vf"] == pytest.approx(g["active_vf"], abs=5e-7)


@pytest.mark.parametrize("xs", [0, 1])
def test_slabs_split_launch_overlap_path(xs):
    """Short z-chunks force boundary + interior launches on every slab (the interior launch
    runs while the q planes travel on the comm stream)."""
    phase = o.gen_spheres((40, 36, 44), 8, 5, 0.5)
    ref = o.tortuosity(phase, 1, 2, -1.0, 1.0, eps=1e-11)
    assert ref["status"] == 0
    try:
        oia.set_tuning(16, 3, 32, 8, xs)  # 3-plane chunks: 4-5 chunks per slab
        outs = run_slabs(phase, 1, 2, 3, -1.0, 1.0, eps=1e-11)
    finally:
        oia.set_tuning()
    assert outs[0]["tau"] == pytest.approx(ref["tau"], rel=1e-8)
    x = np.concatenate([r["x"] for r in outs])
    np.testing.assert_allclose(x, ref["x"], rtol=0, atol=1e-8)


def test_slabs_thin_and_uneven():
    """nz not divisible by the rank count, slabs of one or two planes, flow along z."""
    phase = o.gen_bernoulli((7, 24, 30), 4, 0.7)
    ref = o.tortuosity(phase, 1, 2, eps=1e-11)
    assert ref["status"] == 0
    outs = run_slabs(phase, 1, 2, 5, eps=1e-11)
    assert outs[0]["tau"] == pytest.approx(ref["tau"], rel=1e-8)
    assert [r["span"] for r in outs] == [(0, 2), (2, 4), (4, 5), (5, 6), (6, 7)]


def test_slabs_no_percolation_is_collective():
    ph = np.ones((12, 12, 12), np.uint8)
    ph[6] = 0  # plane z = 6 blocks the z direction
    outs = run_slabs(ph, 1, 2, 3)
    assert all(np.isnan(r["tau"]) and r["status"] == oia.OI_NO_PERCOLATION for r in outs)


@pytest.mark.parametrize("kernel", [oia.KERNEL_AUTO, oia.KERNEL_UNFUSED])
@pytest.mark.parametrize("nranks,direction", [(2, 0), (3, 2), (4, 1)])
def test_slabs_multigrid_replicated_hierarchy(sample_phase, goldens, nranks, direction, kernel):
    """OI_PRECOND_MG on slabs that do not start on even planes (100 planes over 3 or 4 ranks) or are
    small: the first coarse level is summed across ranks and the hierarchy solved redundantly.  The
    cycle is the same linear map as on one GPU, so the iteration count is the same too."""
    g = goldens["phases"]["1"]
    single = oia.TortuosityMI355(sample_phase, 1, direction, g["vlo"], g["vhi"], eps=1e-10, precond=oia.PRECOND_MG,
                                 kernel=kernel)
    tau1, it1 = single.value(), single.getSolverIterations()
    single.destroy()
    outs = run_slabs(np.ascontiguousarray(sample_phase), 1, direction, nranks, g["vlo"], g["vhi"], eps=1e-10,
                     precond=oia.PRECOND_MG, kernel=kernel)
    for r in outs:
        assert r["status"] == oia.OI_OK and r["st"]["converged"]
        assert r["tau"] == outs[0]["tau"]
    assert outs[0]["tau"] == pytest.approx(g["tau"][direction], rel=1e-7)
    assert outs[0]["tau"] == pytest.approx(tau1, rel=1e-8)
    assert outs[0]["st"]["flux_in"] == pytest.approx(g["flux"][direction], rel=1e-7)
    assert abs(outs[0]["st"]["iterations"] - it1) <= 2, (outs[0]["st"]["iterations"], it1)


@pytest.mark.parametrize("nranks,direction,shape", [(2, 2, (64, 40, 52)), (4, 0, (64, 37, 45)), (3, 1, (96, 33, 40)),
                                                    (4, 2, (72, 48, 40))])
def test_slabs_multigrid_distributed_levels(oi_option, nranks, direction, shape):
    """Slab boundaries on multiples of 16 / 8 planes: the first coarse levels keep the slab partition
    (ghost-plane exchanges inside the cycle) until the slabs get thin or misaligned, the rest is
    replicated.  Same iteration count and answer as one GPU."""
    oi_option("mg_replicate_cells", 600)
    phase = o.gen_spheres(shape, 33, 5, 0.45)
    ref = o.tortuosity(phase, 1, direction, -1.0, 1.0, eps=1e-10)
    assert ref["status"] == 0
    single = oia.TortuosityMI355(phase, 1, direction, -1.0, 1.0, eps=1e-10, precond=oia.PRECOND_MG)
    tau1, it1 = single.value(), single.getSolverIterations()
    single.destroy()
    outs = run_slabs(phase, 1, direction, nranks, -1.0, 1.0, eps=1e-10, precond=oia.PRECOND_MG)
    assert all(r["status"] == oia.OI_OK for r in outs)
    assert outs[0]["tau"] == pytest.approx(ref["tau"], rel=1e-7)
    assert outs[0]["tau"] == pytest.approx(tau1, rel=1e-8)
    assert abs(outs[0]["st"]["iterations"] - it1) <= 2, (outs[0]["st"]["iterations"], it1)
    assert it1 < ref["iterations"] / 4
    x = np.concatenate([r["x"] for r in outs])
    np.testing.assert_allclose(x, ref["x"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("nranks,shape", [(2, (64, 48, 256)), (3, (96, 36, 264))])
def test_slabs_multigrid_tiled_coarse_kernels(oi_option, nranks, shape):
    """Round 4: the LDS-tiled coarse-level kernels (k_mg_post4t, k_mg_lin4t; forced onto every four-cell level with
    `mg_coarse_tiled` = 2) on SLAB-partitioned levels -- the tile's z faces are the ghost planes the neighbours sent -- for the
    flow-through problem, and on a ring of slabs for the periodic cell problems: same iteration counts as one GPU, same answers;
    with `mg_coarse_tiled` = 0 (k_mg_post4 / k_mg_lin4) the slab solves are the same bit for bit."""
    rng = np.random.default_rng(11)
    phase = (rng.random(shape) < 0.72).astype(np.uint8)
    oi_option("mg_replicate_cells", 600)
    res = {}
    for v in (2, 0):
        oi_option("mg_coarse_tiled", v, 1)
        single = oia.TortuosityMI355(phase, 1, 0, -1.0, 1.0, eps=1e-10, precond=oia.PRECOND_MG)
        tau1, it1 = single.value(), single.getSolverIterations()
        single.destroy()
        outs = run_slabs(phase, 1, 0, nranks, -1.0, 1.0, eps=1e-10, precond=oia.PRECOND_MG)
        assert all(r["status"] == oia.OI_OK for r in outs)
        assert outs[0]["tau"] == pytest.approx(tau1, rel=1e-8)
        assert abs(outs[0]["st"]["iterations"] - it1) <= 2, (outs[0]["st"]["iterations"], it1)
        res[v] = (outs[0]["tau"], outs[0]["st"]["iterations"], np.concatenate([r["x"] for r in outs]))
    assert res[2][0] == res[0][0] and res[2][1] == res[0][1] and np.array_equal(res[2][2], res[0][2])
    # the periodic cell problems on a ring of the same slabs
    nz = shape[0]
    tens = {}
    for v in (2, 0):
        oi_option("mg_coarse_tiled", v, 1)
        comms = oia.Comm.create_local(nranks)
        out, err = [None] * nranks, []

        def work(r):
            try:
                z0, z1 = oia.slab_range(nz, nranks, r)
                h = oia.EffectiveDiffusivityMI355(np.ascontiguousarray(phase[z0:z1]), 1, eps=1e-10, precond=oia.PRECOND_MG, comm=comms[r], z0=z0,
                                                  nz_global=nz)
                rs = [h.solve(k) for k in range(3)]
                out[r] = (rs, h.tensor().copy())
                h.destroy()
            except Exception as e:
                err.append((r, repr(e)))

        th = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
        [t.start() for t in th]
        [t.join(timeout=300) for t in th]
        assert not any(t.is_alive() for t in th) and not err, err
        for c in comms:
            c.destroy()
        assert all(x[0] for x in out[0][0])
        tens[v] = (out[0][0], out[0][1])
    assert tens[2][0] == tens[0][0] and np.array_equal(tens[2][1], tens[0][1])
    h1 = oia.EffectiveDiffusivityMI355(phase, 1, eps=1e-10, precond=oia.PRECOND_MG)
    r1 = [h1.solve(k) for k in range(3)]
    np.testing.assert_allclose(tens[2][1], h1.tensor(), rtol=0, atol=1e-9)
    assert all(abs(a[1] - b[1]) <= 1 for a, b in zip(tens[2][0], r1))
    h1.destroy()


@pytest.mark.parametrize("nranks", [2, 3])
def test_slabs_multigrid_compacted_lines(sample_phase, goldens, nranks):
    """The multigrid-preconditioned iteration on compacted lines across slabs: ghost planes of r and z travel as
    line lists.  Two slabs of 50 planes start on even planes and compact; with three one slab starts on plane 67,
    which the fused fine-level kernels do not take, and ALL slabs then agree on dense vectors."""
    g = goldens["phases"]["1"]
    single = oia.TortuosityMI355(sample_phase, 1, 2, g["vlo"], g["vhi"], eps=1e-10, precond=oia.PRECOND_MG)
    tau1, it1 = single.value(), single.getSolverIterations()
    single.destroy()
    outs = run_slabs(np.ascontiguousarray(sample_phase), 1, 2, nranks, g["vlo"], g["vhi"], eps=1e-10,
                     precond=oia.PRECOND_MG)
    for r in outs:
        assert r["status"] == oia.OI_OK and r["st"]["converged"]
        assert r["tau"] == outs[0]["tau"]
    assert outs[0]["tau"] == pytest.approx(g["tau"][2], rel=1e-7)
    assert outs[0]["tau"] == pytest.approx(tau1, rel=1e-8)
    assert abs(outs[0]["st"]["iterations"] - it1) <= 2


def test_serpentine_path_across_many_thin_slabs_reaches_the_fixed_point():
    """A single pore channel that snakes up and down in z while it advances in x crosses every slab boundary once per
    wall; reachability moves across a boundary only once per flood round, so 12 slabs x 200 walls need ~2200 rounds --
    more than the 4 (nx + ny + nz) + 64 = 1776 a capped fill would allow.  The fill runs to its fixed point: the whole
    channel is active on every slab, and the solve sees one long 1-voxel-wide conductor."""
    nx, ny, nz, nranks = 401, 3, 24, 12
    phase = np.zeros((nz, ny, nx), np.uint8)
    phase[:, 1, 0::2] = 1                      # vertical shafts at even x
    for m, i in enumerate(range(1, nx, 2)):    # joined alternately at the top and at the bottom
        phase[0 if m % 2 else nz - 1, 1, i] = 1
    n_path = int(phase.sum())
    single = oia.TortuosityMI355(phase, 1, 0, eps=1e-9, precond=oia.PRECOND_JACOBI, kernel=oia.KERNEL_FUSED1)
    assert single.stats().n_active == n_path and np.array_equal(single.mask(), phase)
    single.destroy()
    outs = run_slabs(phase, 1, 0, nranks, eps=1e-9, precond=oia.PRECOND_JACOBI, kernel=oia.KERNEL_FUSED1, maxiter=20)
    mask = np.concatenate([r["mask"] for r in outs])
    assert np.array_equal(mask, phase)
    assert all(r["st"]["n_active"] == n_path for r in outs)
    assert all(r["st"]["flood_sweeps"] > 8 * 1776 for r in outs)


@pytest.mark.parametrize("nranks,box", [(2, 32), (3, 16), (4, 8)])
def test_remspot_on_slabs_equals_the_single_domain(nranks, box):
    """preconditionPhaseFab runs on every rank of the reference; boxes that straddle a slab boundary relax across it
    (ghost planes of the relaxing field travel every sweep).  Same mask and answer as the single-domain handle and as
    the oracle's box-wise pass."""
    phase = o.gen_bernoulli((44, 40, 52), 17, 0.62)
    clean = o.remspot(phase, 2, box_size=box)
    ref = o.tortuosity(clean, 1, 2, eps=1e-10)
    outs = run_slabs(phase, 1, 2, nranks, eps=1e-10, remspot_passes=2, remspot_box_size=box, precond=oia.PRECOND_JACOBI,
                     kernel=oia.KERNEL_FUSED1)
    mask = np.concatenate([r["mask"] for r in outs])
    assert np.array_equal(mask, ref["mask"])
    assert all(r["status"] == oia.OI_OK for r in outs)
    assert outs[0]["tau"] == pytest.approx(ref["tau"], rel=1e-8)


def test_placement_search_on_slabs_with_neighbours(oi_option):
    """The placement search of the single-kernel iteration on z-slabs WITH neighbours: the decision to search, the number of
    trials, the second-set decision and the cache verdict are all-reduced (each of them guards launches that have
    collectives inside).  `place_min_mb` = 0 makes these small slabs search; two handles per rank, so that the second one
    goes through the cache path with its all-reduces too.  Same answer as the single domain."""
    oi_option("place_min_mb", 0, reset=256)
    oi_option("place_trials", 6, reset=48)
    oi_option("place_slots", 8, reset=12)
    oi_option("place_spacer_gb", 0.0, reset=48.0)
    oi_option("place_alloc_budget_s", 1e9, reset=0.5)
    phase = o.gen_spheres((48, 40, 44), 21, 5, 0.45)
    ref = o.tortuosity(phase, 1, 2, -1.0, 1.0, eps=1e-10)
    nranks, nz = 3, phase.shape[0]
    comms = oia.Comm.create_local(nranks)
    out, err = [None] * nranks, []

    def work(r):
        try:
            z0, z1 = oia.slab_range(nz, nranks, r)
            res = []
            for rep in range(2):
                t = oia.TortuosityMI355(np.ascontiguousarray(phase[z0:z1]), 1, 2, -1.0, 1.0, comm=comms[r], z0=z0, nz_global=nz,
                                        eps=1e-10, kernel=oia.KERNEL_FUSED1, precond=oia.PRECOND_NONE)
                pl = t.placement()
                res.append((t.value(), pl, t.stats().bytes_device))
                t.destroy()
            out[r] = res
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
        (tau0, pl0, b0), (tau1, pl1, b1) = out[r]
        assert tau0 == pytest.approx(ref["tau"], rel=1e-7) and tau1 == tau0
        assert pl0["nslots"] == 8 and pl0["tried"] in (6, 12) and not pl0["cache_hit"]
        assert pl1["tried"] in (1, 6, 12) and (pl1["tried"] == 1) == pl1["cache_hit"]
        assert b0 == b1
    # every rank took the same decisions
    assert len({out[r][1][1]["cache_hit"] for r in range(nranks)}) == 1
