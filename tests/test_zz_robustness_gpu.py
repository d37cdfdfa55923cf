"""Robustness of the C ABI on the GPU: misuse, concurrent handles from host threads, out-of-memory, release of device memory.

These are collected LAST (the file name sorts after every parity file, and tests/conftest.py moves them to the end of any
selection): a failure here can never hide a parity row under `pytest -x`.
"""
import math
import os

import numpy as np
import pytest

import openimpala_amd as oia

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TIF = os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif")


@pytest.fixture(scope="module")
def sample():
    return oia.TiffReader(TIF).threshold(0.5)


def test_api_misuse_is_reported_not_fatal(sample):
    """Calls out of order, bad arguments, use after destroy: OI_BAD_ARG with a message that names the call."""
    sub = np.ascontiguousarray(sample[:24, :28, :32])
    t = oia.TortuosityMI355(sub, 1, 0)
    for call in (lambda: t.cg_iterate(3), t.cg_finish, lambda: t.set_eps(0.0), lambda: t.set_eps(float("nan"))):
        with pytest.raises(oia.OpenImpalaError) as e:
            call()
        assert e.value.status == oia.OI_BAD_ARG and "oi_tort_" in str(e.value)
    assert t.cg_state() == (0, 0.0, False) and t.solution().shape == sub.shape
    tau = t.value()
    assert tau > 1.0 and t.value() == tau and t.value(refresh=True) == pytest.approx(tau, rel=1e-12)
    t.cg_begin()
    with pytest.raises(oia.OpenImpalaError):   # the stand-alone apply is refused while a CG run is open
        t.apply(np.zeros(sub.shape))
    with pytest.raises(oia.OpenImpalaError):
        t.cg_iterate(-5)
    t.cg_iterate(0); t.cg_finish()
    t.destroy(); t.destroy()
    with pytest.raises(oia.OpenImpalaError):
        t.value()
    for kw in (dict(kernel=99), dict(eps=-1.0)):
        with pytest.raises(oia.OpenImpalaError):
            oia.TortuosityMI355(sub, 1, 0, **kw)
    with pytest.raises(oia.OpenImpalaError):
        oia.TortuosityMI355(np.zeros((0, 4, 4), np.uint8), 1, 0)
    assert math.isnan(oia.TortuosityMI355(sub, 5, 0).value())            # phase id absent: no percolation
    assert math.isnan(oia.TortuosityMI355(sub, 1, 0, maxiter=1).value())  # not converged
    h = oia.EffectiveDiffusivityMI355(sub, 1)
    with pytest.raises(oia.OpenImpalaError):
        h.tensor()
    with pytest.raises(oia.OpenImpalaError) as e:
        h.solve(5)
    assert "oi_effdiff_solve" in str(e.value)
    h.destroy(); h.destroy()


def test_independent_handles_in_concurrent_host_threads(sample, goldens):
    """Distinct handles may be driven from distinct host threads at once (the reference runs one solver object per
    MPI rank; the app's gpu.ngpus mode and a parallel REV study do this): three directions, two preconditioners."""
    import threading
    g = goldens["phases"]["1"]
    out, err = {}, []

    def work(d, precond):
        try:
            t = oia.TortuosityMI355(sample, 1, d, g["vlo"], g["vhi"], eps=1e-10, precond=precond)
            out[(d, precond)] = (t.value(), t.stats().flux_in)
            t.destroy()
        except Exception as e:  # surface it in the main thread
            err.append((d, precond, repr(e)))

    th = [threading.Thread(target=work, args=(d, p)) for d in range(3) for p in (oia.PRECOND_MG, oia.PRECOND_JACOBI)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th) and not err, err
    for (d, _), (tau, flux) in out.items():
        assert tau == pytest.approx(g["tau"][d], rel=1e-7) and flux == pytest.approx(g["flux"][d], rel=1e-7)


def test_out_of_memory_is_an_error_not_a_crash():
    """A volume that cannot fit (34e9 voxels: x alone would take 275 GB next to 70 GB of phase and code bytes) fails
    with OI_HIP_ERROR, gives back what it had allocated (the ledger is where it was), and the library keeps working."""
    live0 = oia.dev_live()
    shape = (2048, 4096, 4096)
    buf = oia.gen_bernoulli_device(shape, 3, 0.6)
    with pytest.raises(oia.OpenImpalaError) as e:
        oia.TortuosityMI355(buf, 1, 0, shape=shape)
    assert e.value.status == oia.OI_HIP_ERROR and "out of memory" in str(e.value)
    buf.free()
    assert oia.dev_live() == live0
    ph = oia.gen_bernoulli_device((48, 48, 48), 3, 0.7)
    t = oia.TortuosityMI355(ph, 1, 0, shape=(48, 48, 48))
    assert t.value() > 1.0
    t.destroy(); ph.free()
    assert oia.dev_live() == live0


def test_an_allocation_that_fails_at_any_point_of_the_build_leaves_nothing_behind(sample):
    """The n-th device allocation of a handle's construction fails (oi_debug_fail_alloc_after), for every n until the build goes
    through: each failure is OI_HIP_ERROR, and the ledger is back at its starting point every time -- no array of a half-built
    handle (tortuosity single-kernel / multigrid, homogenization) outlives the error."""
    sub = np.ascontiguousarray(sample[:40, :44, :48])
    live0 = oia.dev_live()
    makers = (lambda: oia.TortuosityMI355(sub, 1, 0, eps=1e-8, kernel=oia.KERNEL_FUSED1),
              lambda: oia.TortuosityMI355(sub, 1, 2, eps=1e-8),
              lambda: oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-8, precond=oia.PRECOND_MG))
    try:
        for make in makers:
            for n in range(1, 400):
                oia.debug_fail_alloc_after(n)
                try:
                    h = make()
                except oia.OpenImpalaError as e:
                    assert e.status == oia.OI_HIP_ERROR, (n, str(e))
                    assert oia.dev_live() == live0, (n, oia.dev_live(), live0)
                    continue
                oia.debug_fail_alloc_after(0)
                h.destroy()
                assert oia.dev_live() == live0
                assert n > 10   # (a handle makes more than ten allocations: the hook did bite)
                break
            else:
                raise AssertionError("the build never went through")
    finally:
        oia.debug_fail_alloc_after(0)


def test_device_pool_keeps_and_returns_blocks(sample):
    """oi_dev_pool (off by default): with it ON the device arrays of a destroyed handle are kept for the next handle of that
    shape -- same results, zeroed blocks, hits counted -- and trim / switching it off hands everything back to the driver."""
    sub = np.ascontiguousarray(sample[:64, :72, :80])

    def solve_all():
        out = []
        for d in range(3):
            t = oia.TortuosityMI355(sub, 1, d, -1.0, 1.0, eps=1e-10)
            out.append((t.value(), t.getSolverIterations(), t.stats().final_relres, t.solution().copy()))
            t.destroy()
        h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-9, precond=oia.PRECOND_MG)
        out.append(tuple(h.solve(k) for k in range(3)) + (h.tensor().copy(),))
        h.destroy()
        return out

    assert not oia.dev_pool_stats()["enabled"]
    live0 = oia.dev_live()
    plain = solve_all()
    assert oia.dev_live() == live0
    try:
        oia.dev_pool(True)
        first = solve_all()
        st1 = oia.dev_pool_stats()
        again = solve_all()   # every array of this pass comes out of the pool
        st2 = oia.dev_pool_stats()
        assert st1["enabled"] and st1["held_blocks"] > 10 and st2["hits"] > st1["hits"] + 30 and st2["misses"] - st1["misses"] < 10, (st1, st2)
        assert oia.dev_live() == live0 and st2["held_bytes"] > 0   # (pooled blocks are not live; the pool accounts for them)
        for a, b, c in zip(plain[:3], first[:3], again[:3]):
            assert a[0] == b[0] == c[0] and a[1] == b[1] == c[1] and a[2] == b[2] == c[2]
            assert np.array_equal(a[3], b[3]) and np.array_equal(a[3], c[3])
        assert plain[3][:3] == first[3][:3] == again[3][:3] and np.array_equal(plain[3][3], again[3][3])
        released = oia.dev_pool_trim()
        assert released == st2["held_bytes"] and oia.dev_pool_stats()["held_blocks"] == 0
    finally:
        oia.dev_pool(False)
    assert oia.dev_live() == live0 and oia.dev_pool_stats()["held_bytes"] == 0


def test_an_allocation_the_driver_refuses_gets_the_pool_back_first(sample):
    """Pool ON and holding the arrays of a destroyed handle; the driver refuses an allocation of the NEXT handle that does not come
    from the pool (a vector slot of the single-kernel iteration: oi_tort.cpp vecset_alloc; oi_dev_malloc): the pool is handed back
    to the driver and the request repeated -- the build goes through where round 4 reported 'out of device memory'."""
    sub = np.ascontiguousarray(sample[:64, :72, :80])
    live0 = oia.dev_live()
    try:
        oia.dev_pool(True)
        t = oia.TortuosityMI355(sub, 1, 0, eps=1e-9)
        tau = t.value()
        t.destroy()
        held = oia.dev_pool_stats()["held_bytes"]
        assert held > 0 and oia.dev_live() == live0
        # the first driver allocations of a single-kernel handle built next (its record array, then the vector set, which is
        # never pooled) are refused once each, while the pool still holds the multigrid handle's vectors
        for n in (1, 2, 3):
            oia.dev_pool_trim()
            t = oia.TortuosityMI355(sub, 1, 0, eps=1e-9); t.destroy()       # refill the pool
            assert oia.dev_pool_stats()["held_bytes"] > 0
            oia.debug_fail_alloc_after(n)
            t1 = oia.TortuosityMI355(sub, 1, 0, eps=1e-9, kernel=oia.KERNEL_FUSED1)
            oia.debug_fail_alloc_after(0)
            assert t1.value() == pytest.approx(tau, rel=1e-7)
            t1.destroy()
        # with nothing to give back the refusal is the error it was
        oia.dev_pool_trim()
        oia.debug_fail_alloc_after(1)
        with pytest.raises(oia.OpenImpalaError):
            oia.TortuosityMI355(sub, 1, 0, eps=1e-9, kernel=oia.KERNEL_FUSED1)
        oia.debug_fail_alloc_after(0)
        oia.dev_pool_trim()
        assert oia.dev_live() == live0
        t = oia.TortuosityMI355(sub, 1, 0, eps=1e-9); t.destroy()
        buf = None
        oia.debug_fail_alloc_after(1)
        buf = oia.gen_bernoulli_device((32, 32, 32), 1, 0.5)    # oi_dev_malloc: the same path
        oia.debug_fail_alloc_after(0)
        buf.free()
    finally:
        oia.debug_fail_alloc_after(0)
        oia.dev_pool(False)
    assert oia.dev_live() == live0 and oia.dev_pool_stats()["held_bytes"] == 0


RELEASE_VARIANTS = (("mg", dict()), ("cg1", dict(kernel=oia.KERNEL_FUSED1)), ("lean_jacobi", dict(kernel=oia.KERNEL_LEAN, precond=oia.PRECOND_JACOBI)),
                    ("fused", dict(kernel=oia.KERNEL_FUSED, precond=oia.PRECOND_NONE)), ("unfused_mg", dict(kernel=oia.KERNEL_UNFUSED, precond=oia.PRECOND_MG)))


def test_handles_release_their_device_memory(sample):
    """A REV study creates and destroys hundreds of handles: every variant gives ALL its device memory back, and while it is alive
    it holds exactly what its stats account for.  Read from the library's allocation ledger (oi_dev_live_bytes / _blocks: every
    device allocation of the library is entered there when made and struck when freed), so the assertions are equalities -- the
    earlier form of this test sampled hipMemGetInfo, which reports freed arrays late, and needed 60 cycles and a 64 MiB window."""
    sub = np.ascontiguousarray(sample[:96, :90, :88])
    live0 = oia.dev_live()
    for name, kw in (("effdiff", None),) + RELEASE_VARIANTS:
        for cycle in range(2):
            if kw is None:
                h = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-8, precond=oia.PRECOND_MG)
                assert h.solve(0)[0]
                assert oia.dev_live()[0] > live0[0]
                h.destroy()
            else:
                t = oia.TortuosityMI355(sub, 1, 0, eps=1e-8, **kw)
                # alive, before the solve: the handle holds what it says it holds, to the byte
                assert oia.dev_live()[0] - live0[0] == int(t.stats().bytes_device), (name, oia.dev_live()[0] - live0[0], t.stats().bytes_device)
                assert t.value() > 1.0
                assert oia.dev_live()[0] - live0[0] == int(t.stats().bytes_device), name   # (the solve's scratch is gone again)
                t.solution(); t.mask(); t.codes()
                assert t.checkMatrixProperties()
                assert oia.dev_live()[0] - live0[0] == int(t.stats().bytes_device), name
                t.destroy()
            assert oia.dev_live() == live0, (name, cycle, oia.dev_live(), live0)


def test_searching_single_kernel_handles_release_their_arenas(oi_option):
    """A single-kernel CG handle whose vectors exceed the placement threshold builds slot arenas (and, for the search, maps
    spare slots and a distance between the two arena halves): all of it is given back -- also the spare slots the search
    did not choose, while the handle is still alive (ledger equalities)."""
    shape = (384, 384, 400)
    live0 = oia.dev_live()
    ph = oia.gen_bernoulli_device(shape, 7, 0.75)
    live_ph = oia.dev_live()
    assert live_ph[0] - live0[0] == int(np.prod(shape)) and live_ph[1] == live0[1] + 1
    oi_option("place_alloc_budget_s", 1e9, reset=0.5)   # (spare slots whatever the driver's clearing of them costs)
    for n in range(3):
        oia.dev_peak_bytes(reset=True)
        t = oia.TortuosityMI355(ph, 1, 0, shape=shape, kernel=oia.KERNEL_FUSED1, eps=1e-3, maxiter=50)
        pl = t.placement()
        # the search did run: in full for the first handle; the later ones may take its placement from the process cache
        # (one verification cycle) -- they allocate and free the same spare slots either way
        assert pl["nslots"] > 6 and (pl["tried"] > 1 or (n > 0 and pl["cache_hit"])), pl
        assert pl["peak_device_bytes"] > t.stats().bytes_device + 2 * 400e6, pl
        # the ledger saw that peak too (spare slots + spacer), and what is alive now is what the handle accounts for
        assert oia.dev_peak_bytes() - live_ph[0] >= pl["peak_device_bytes"]
        assert oia.dev_live()[0] - live_ph[0] == int(t.stats().bytes_device)
        t.cg_begin(); t.cg_iterate(4); t.cg_finish()
        t.destroy()
        assert oia.dev_live() == live_ph
    ph.free()
    assert oia.dev_live() == live0


def _slab_builds_with_a_refused_allocation(name, make, nranks, nz, n_fail):
    """`nranks` host threads build one slab each; the driver refuses allocation number `n_fail` (whichever slab asks for it).
    Returns (number of slabs that failed, their messages); asserts that nobody hangs and nothing leaks."""
    import threading
    live1 = oia.dev_live()
    comms = oia.Comm.create_local(nranks)
    out, err = [None] * nranks, [None] * nranks

    def work(r):
        try:
            z0, z1 = oia.slab_range(nz, nranks, r)
            h = make(comms[r], z0, z1)
            out[r] = h
        except oia.OpenImpalaError as e:
            err[r] = str(e)

    oia.debug_fail_alloc_after(n_fail)
    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]   # (daemon: a hang fails the test, not the exit)
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    oia.debug_fail_alloc_after(0)
    assert not any(t.is_alive() for t in th), f"{name}: slab threads hung with allocation {n_fail} refused"
    failed = [e for e in err if e is not None]
    # all or nothing: the slabs agree on the outcome of every set-up step
    assert len(failed) in (0, nranks), (n_fail, err)
    for h in out:
        if h is not None:
            h.destroy()
    for c in comms:
        c.destroy()
    assert oia.dev_live() == live1, (n_fail, oia.dev_live(), live1)   # (the transport's staging buffers went with it)
    return len(failed), failed


def test_a_slab_that_cannot_allocate_takes_its_neighbours_out_with_an_error(sample):
    """ADVICE r4 (oi_effdiff.hip ring set-up; the same pattern in the flow-through build and the multigrid hierarchy): an allocation
    that fails on ONE in-process rank used to leave its peers in the next collective for good.  Every rank-local stretch of a set-up
    now ends in an agreement (oi::agree_status): all slabs return an error -- the failing one its own, the others 'another slab
    failed' -- nobody waits, nothing is left allocated."""
    sub = np.ascontiguousarray(sample[:48, :40, :44])
    nz = sub.shape[0]
    live0 = oia.dev_live()
    makers = {
        "tortuosity mg": lambda c, z0, z1: oia.TortuosityMI355(np.ascontiguousarray(sub[z0:z1]), 1, 0, eps=1e-8, comm=c, z0=z0, nz_global=nz),
        "tortuosity cg1": lambda c, z0, z1: oia.TortuosityMI355(np.ascontiguousarray(sub[z0:z1]), 1, 2, eps=1e-8, kernel=oia.KERNEL_FUSED1, comm=c, z0=z0, nz_global=nz),
        "homogenization ring": lambda c, z0, z1: oia.EffectiveDiffusivityMI355(np.ascontiguousarray(sub[z0:z1]), 1, eps=1e-8, precond=oia.PRECOND_MG, comm=c, z0=z0, nz_global=nz),
    }
    for name, make in makers.items():
        seen_other = False
        for n_fail in list(range(1, 60, 3)) + list(range(60, 400, 17)):
            nfailed, msgs = _slab_builds_with_a_refused_allocation(name, make, 2, nz, n_fail)
            if nfailed == 0:
                break   # the build makes fewer allocations than that: every step has been covered
            seen_other = seen_other or any("another slab failed" in m for m in msgs)
        else:
            raise AssertionError(f"{name}: the build never went through")
        assert seen_other, name   # (at least once the failure was somebody else's)
    assert oia.dev_live() == live0
