"""CPU tests that pin the oracle (oracle/oi_oracle.c) before anything trusts it.

Pins: BASELINE.md section-2 goldens of the bundled 100^3 TIFF, the analytic known
answers of the reference's discretisation, the reference's own
checkMatrixProperties invariants (TortuosityHypre.cpp:896-982) on materialised
rows, its flux-conservation bound (TortuosityHypre.cpp:794-804) and its reader
assertions (tTiffReader.cpp:98-103: 100^3, 1 bit per sample).
"""
import math
import os

import numpy as np
import pytest

from oracle import oracle as o

from conftest import GOLDEN

TIF = os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif")


def test_tiff_reader_dims_and_bps(goldens):
    raw, bps = o.read_tiff_stack(TIF)
    assert list(raw.shape) == goldens["dims"]
    assert bps == goldens["bits_per_sample"]
    assert set(np.unique(raw)) == {0.0, 1.0}


def test_volume_fraction_goldens(sample_phase, goldens):
    for pid in ("0", "1"):
        pc, tc = o.volume_fraction(sample_phase, int(pid))
        assert tc == 100 ** 3
        assert pc == goldens["phases"][pid]["voxels"]
        assert pc / tc == pytest.approx(goldens["phases"][pid]["volume_fraction"], abs=1e-12)


@pytest.mark.parametrize("pid", [0, 1])
@pytest.mark.parametrize("direction", [0, 1, 2])
def test_sample_goldens_matrix_free(sample_phase, goldens, pid, direction):
    g = goldens["phases"][str(pid)]
    r = o.tortuosity(sample_phase, pid, direction, g["vlo"], g["vhi"], eps=1e-11)
    assert r["status"] == 0 and r["converged"]
    assert r["active_vf"] == pytest.approx(g["active_vf"], abs=5e-7)
    assert r["n_free"] == g["free_unknowns"][direction]
    # goldens are printed to 8-10 significant digits
    assert r["flux_in"] == pytest.approx(g["flux"][direction], rel=2e-9)
    assert r["flux_out"] == pytest.approx(g["flux"][direction], rel=2e-9)
    assert r["deff"] == pytest.approx(g["deff"][direction], rel=2e-9)
    assert r["tau"] == pytest.approx(g["tau"][direction], rel=1e-9)
    # the reference's own acceptance test
    fi, fo = abs(r["flux_in"]), abs(r["flux_out"])
    assert abs(fi - fo) / (0.5 * (fi + fo)) <= 1e-6


def test_sample_assembled_system_matches_matrix_free(sample_phase, goldens):
    """The reference-shaped assembled system (fillmtx rows + identity-row elimination)
    and the matrix-free operator are the same linear system."""
    g = goldens["phases"]["1"]
    ra = o.tortuosity(sample_phase, 1, 0, g["vlo"], g["vhi"], eps=1e-11, assembled=True)
    assert ra["tau"] == pytest.approx(g["tau"][0], rel=1e-9)
    rj = o.tortuosity(sample_phase, 1, 0, g["vlo"], g["vhi"], eps=1e-11, jacobi=True)
    assert rj["tau"] == pytest.approx(g["tau"][0], rel=1e-9)
    assert rj["iterations"] < ra["iterations"] * 1.05


def test_rtol_guidance(sample_phase, goldens):
    """rtol 1e-8 agrees to <= 1.3e-7 (the survey measured this with its own
    residual normalisation; with HYPRE's ||r||/||b_full|| normalisation used here the
    reference default 1e-9 is the parity setting)."""
    g = goldens["phases"]["0"]
    r = o.tortuosity(sample_phase, 0, 0, g["vlo"], g["vhi"], eps=1e-9)
    assert r["status"] == 0
    assert r["tau"] == pytest.approx(g["tau"][0], rel=1e-5)


@pytest.mark.parametrize("direction", [0, 1, 2])
def test_check_matrix_properties_invariants(sample_phase, direction):
    sub = np.ascontiguousarray(sample_phase[:40, :36, :44])
    mask, nact = o.active_mask(sub, 1, direction)
    assert nact > 0
    a, rhs, xinit = o.fillmtx(sub, mask, 1, direction, -1.0, 1.0)
    assert o.check_matrix(a, rhs, mask, direction, -1.0, 1.0) == 0
    # as assembled the matrix is NOT symmetric (Dirichlet columns left in place)
    nz, ny, nx = sub.shape
    code = o.codes(mask, direction)
    free = code.reshape(-1) > 0
    # matrix-free operator == assembled rows on a random vector that vanishes off the free set
    rng = np.random.default_rng(1)
    v = np.where(free, rng.standard_normal(free.size), 0.0)
    offs = [0, -1, 1, -nx, nx, -nx * ny, nx * ny]
    av = np.zeros_like(v)
    idx = np.nonzero(free)[0]
    for s, off in enumerate(offs):
        coef = a[idx, s]
        nb = idx + off
        ok = coef != 0
        av[idx[ok]] += coef[ok] * v[nb[ok]]
    q = o.apply(code, v.reshape(sub.shape)).reshape(-1)
    np.testing.assert_allclose(q[free], av[free], rtol=0, atol=1e-12)
    assert np.all(q[~free] == 0)
    # the diagonal of every free row is the neighbour count
    assert np.array_equal(a[idx, 0], code.reshape(-1)[idx].astype(float))
    # corrupting one coefficient trips the invariant checker
    a2 = a.copy()
    a2[idx[0], 0] += 0.5
    assert o.check_matrix(a2, rhs, mask, direction, -1.0, 1.0) > 0


def test_initial_field_is_reference_ramp(sample_phase):
    sub = np.ascontiguousarray(sample_phase[:20, :24, :28])
    for d in range(3):
        mask, _ = o.active_mask(sub, 0, d)
        a, rhs, xinit = o.fillmtx(sub, mask, 0, d, -1.0, 1.0)
        x0 = o.initial_field(mask, d, -1.0, 1.0).reshape(-1)
        skipped = (a[:, 0] == 1.0) & (np.abs(a[:, 1:]).sum(1) > 0)  # F90:233 leaves these unassigned
        np.testing.assert_allclose(x0[~skipped], xinit[~skipped], rtol=0, atol=3e-16)


def test_all_pore_cube_known_answer(goldens):
    n = 16
    ph = np.ones((n, n, n), np.uint8)
    for d in range(3):
        r = o.tortuosity(ph, 1, d, -1.0, 1.0, eps=1e-12)
        assert r["active_vf"] == 1.0
        assert r["tau"] == pytest.approx(goldens["analytic"]["all_pore_16"]["tau"], rel=1e-10)
        assert r["deff"] == pytest.approx(goldens["analytic"]["all_pore_16"]["deff"], rel=1e-10)


def test_channel_known_answer(goldens):
    n = 16
    ph = np.zeros((n, n, n), np.uint8)
    ph[6:10, 6:10, :] = 1  # 4x4 channel along x
    r = o.tortuosity(ph, 1, 0)
    assert r["active_vf"] == goldens["analytic"]["channel_4x4_16_along_x"]["active_vf"]
    assert r["tau"] == pytest.approx(goldens["analytic"]["channel_4x4_16_along_x"]["tau"], rel=1e-10)
    r = o.tortuosity(ph, 1, 1)
    assert r["active_vf"] == 0.0 and math.isnan(r["tau"]) and r["status"] == 2


def test_blocked_plane_and_dead_ends():
    n = 12
    ph = np.ones((n, n, n), np.uint8)
    ph[:, :, 5] = 0  # plane x = 5 blocks X
    assert math.isnan(o.tortuosity(ph, 1, 0)["tau"])
    r = o.tortuosity(ph, 1, 1, eps=1e-12)
    assert r["tau"] == pytest.approx((n - 1) / n, rel=1e-9)
    assert r["active_vf"] == pytest.approx((n - 1) / n)
    # a dead-end pocket touching only the lo face is not active
    ph2 = np.zeros((n, n, n), np.uint8)
    ph2[2:4, 2:4, :] = 1
    ph2[8, 8, 0:5] = 1
    mask, nact = o.active_mask(ph2, 1, 0)
    assert nact == 4 * n and mask[8, 8, 0] == 0


def test_tau_invariant_to_boundary_values(sample_phase):
    sub = np.ascontiguousarray(sample_phase[:32, :32, :32])
    t1 = o.tortuosity(sub, 0, 0, 0.0, 1.0, eps=1e-12)["tau"]
    t2 = o.tortuosity(sub, 0, 0, -1.0, 1.0, eps=1e-12)["tau"]
    t3 = o.tortuosity(sub, 0, 0, 2.0, -5.0, eps=1e-12)["tau"]
    assert t1 == pytest.approx(t2, rel=1e-9) and t1 == pytest.approx(t3, rel=1e-9)


def test_value_edge_cases():
    shape = (8, 8, 8)
    t, deff, st = o.tau(0.5, True, -1.0, -1.0, shape, 0, 1.0, 1.0)  # vlo == vhi
    assert math.isinf(t) and st == 0
    t, deff, st = o.tau(0.5, True, 0.0, 0.0, shape, 0, 0.0, 1.0)  # no flux
    assert math.isinf(t)
    t, deff, st = o.tau(0.5, True, -1.0, -1.00001, shape, 0, 0.0, 1.0)  # flux mismatch 1e-5
    assert math.isnan(t) and st == 4
    t, deff, st = o.tau(0.5, False, -1.0, -1.0, shape, 0, 0.0, 1.0)
    assert math.isnan(t) and st == 3
    t, deff, st = o.tau(0.0, True, -1.0, -1.0, shape, 0, 0.0, 1.0)
    assert math.isnan(t) and st == 2


def test_remspot_flips_isolated_voxels():
    ph = np.zeros((5, 5, 5), np.uint8)
    ph[2, 2, 2] = 1
    out = o.remspot(ph)
    assert out.sum() == 0
    ph = np.ones((4, 4, 4), np.uint8)
    ph[0, 0, 0] = 0  # corner voxel, neighbours outside the domain never match
    assert o.remspot(ph).sum() == 64


def test_generators_are_deterministic_and_percolate():
    a = o.gen_bernoulli((24, 20, 28), 20251031, 0.5)
    b = o.gen_bernoulli((24, 20, 28), 20251031, 0.5)
    assert np.array_equal(a, b) and 0.45 < a.mean() < 0.55
    assert not np.array_equal(a, o.gen_bernoulli((24, 20, 28), 7, 0.5))
    bl = o.gen_blobs((40, 44, 48), 5, 8, 24000)
    assert np.array_equal(bl, o.gen_blobs((40, 44, 48), 5, 8, 24000)) and 0.1 < bl.mean() < 0.35
    assert np.all(o.gen_blobs((8, 8, 8), 5, 8, 65535) == 1) and np.all(o.gen_blobs((8, 8, 8), 5, 8, 0) == 0)
    s = o.gen_spheres((48, 48, 48), 3, 6, 0.4)
    assert 0.25 < s.mean() < 0.55
    r = o.tortuosity(s, 1, 0, eps=1e-10)
    assert r["status"] == 0 and r["tau"] > 1.0
