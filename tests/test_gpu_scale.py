"""BASELINE.json configs at their real per-GPU sizes (no oracle can run there): C3 = 1024^3 spheres in Y and Z (X lives in
test_gpu_parity.py), C4's per-GPU share 2048 x 2048 x 256 as two z-slabs against the single-domain handle, C5's per-GPU
share 4096 x 4096 x 512 at porosity 0.2 with Jacobi-PCG.  Checks are size-independent properties: two solver paths built
from different kernels agree; the TRUE residual b - A x, recomputed with the stand-alone apply kernel (which no solver
variant uses inside its iteration), matches the recurrence residual; the reference's own acceptance tests hold (flux
conservation 1e-6, TortuosityHypre.cpp:794-804; checkMatrixProperties, :896-982)."""
import threading

import numpy as np
import pytest

import openimpala_amd as oia

pytestmark = pytest.mark.gpu


def flux_gap(st):
    return abs(abs(st.flux_in) - abs(st.flux_out)) / (0.5 * (abs(st.flux_in) + abs(st.flux_out)))


@pytest.mark.parametrize("direction", [1, 2])
def test_c3_1024_spheres_y_and_z(direction):
    """Multigrid-PCG against the single-kernel plain CG, both to the reference's default 1e-9: tau and fluxes to 1e-6
    (at this size CG stopped at 1e-8 still leaves 1.0e-6 in tau; measured)."""
    shape = (1024, 1024, 1024)
    buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
    t = oia.TortuosityMI355(buf, 1, direction, shape=shape, eps=1e-9, precond=oia.PRECOND_MG)
    assert t.checkMatrixProperties()
    tau = t.value()
    st = t.stats()
    assert t.status == oia.OI_OK and st.converged and st.iterations < 60 and tau > 1.0
    assert 0.35 < st.active_vf < 0.45
    assert flux_gap(st) <= 1e-6
    true_rel = t.true_residual()
    assert true_rel <= 2e-9 and true_rel == pytest.approx(st.final_relres, rel=0.05)  # recurrence residual = true residual
    t.destroy()
    t = oia.TortuosityMI355(buf, 1, direction, shape=shape, eps=1e-9, precond=oia.PRECOND_NONE, kernel=oia.KERNEL_FUSED1)
    tau_cg = t.value()
    s2 = t.stats()
    assert t.status == oia.OI_OK and s2.converged and s2.iterations > 1000
    assert tau_cg == pytest.approx(tau, rel=1e-6)
    assert s2.flux_in == pytest.approx(st.flux_in, rel=1e-6) and s2.flux_out == pytest.approx(st.flux_out, rel=1e-6)
    assert t.true_residual() <= 1.5e-9
    t.destroy(); buf.free()


def test_c4_share_2048x2048x256_two_slabs_equal_the_single_domain():
    """BASELINE config 4 puts 256 planes of 2048^2 on every GPU.  That share as ONE domain against the same volume as two
    z-slabs of 128 planes (two host threads, in-process transport: every line of the slab path except the RCCL calls):
    same multigrid-PCG iteration count, same tau and fluxes."""
    shape = (256, 2048, 2048)
    buf = oia.gen_spheres_device(shape, 20251031, 32, 0.4)
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-9, precond=oia.PRECOND_MG)
    tau = t.value()
    st = t.stats()
    assert t.status == oia.OI_OK and st.converged and flux_gap(st) <= 1e-6 and t.true_residual() <= 2e-9
    single = (tau, st.iterations, st.flux_in, st.flux_out, st.n_active, st.n_free)
    t.destroy(); buf.free()
    comms = oia.Comm.create_local(2)
    out, err = [None, None], []

    def work(r):
        try:
            z0, z1 = oia.slab_range(shape[0], 2, r)
            b = oia.gen_spheres_device(shape, 20251031, 32, 0.4, z0=z0, nz_local=z1 - z0)
            ts = oia.TortuosityMI355(b, 1, 0, shape=(z1 - z0, shape[1], shape[2]), comm=comms[r], z0=z0, nz_global=shape[0],
                                     eps=1e-9, precond=oia.PRECOND_MG)
            b.free()
            v = ts.value()
            s = ts.stats()
            out[r] = (v, s.iterations, s.flux_in, s.flux_out, s.n_active, s.n_free, ts.status, ts.true_residual())
            ts.destroy()
        except Exception as e:  # surface the failure in the main thread
            err.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    [x.start() for x in th]
    [x.join(timeout=400) for x in th]
    assert not any(x.is_alive() for x in th), "slab threads hung"
    assert not err, err
    for c in comms:
        c.destroy()
    for r in range(2):
        assert out[r][6] == oia.OI_OK and out[r][7] <= 2e-9
        assert out[r][1] == single[1] and out[r][4:6] == single[4:6]  # iterations, active and free counts (global)
        assert out[r][0] == pytest.approx(single[0], rel=1e-9)
        assert out[r][2] == pytest.approx(single[2], rel=1e-9) and out[r][3] == pytest.approx(single[3], rel=1e-9)
    assert out[0][:4] == out[1][:4]  # both ranks report the same global answer


def test_c5_share_4096x4096x512_porosity_02_jacobi_pcg():
    """BASELINE config 5 puts 4096 x 4096 x 512 voxels of a porosity-0.2 structure on every GPU (8.6e9 voxels, ~190-215 GB)
    and asks for Jacobi-PCG.  K iterations of the default single-kernel iteration (compacted lines) and of the capacity
    mode (x, r, p only) -- different kernels, same algorithm -- leave the same residual; the single-kernel iterate's TRUE
    residual equals its recurrence residual; multigrid-PCG then solves the share to 1e-9 and passes the flux gate."""
    shape = (512, 4096, 4096)
    K = 40
    buf = oia.gen_blobs_device(shape, 20251031, 24, 0.2)
    hist = {}
    for name, kern in (("cg1", oia.KERNEL_FUSED1), ("lean", oia.KERNEL_LEAN)):
        t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=100000, precond=oia.PRECOND_JACOBI, kernel=kern)
        st = t.stats()
        assert 0.15 < st.vf < 0.25 and 0.0 < st.active_vf <= st.vf
        # (launch k of the single-kernel iteration forms r_k and p_k: K + 1 launches leave the iterate x_K)
        t.cg_begin(); t.cg_iterate(K + 1 if name == "cg1" else K); t.sync()
        its, rel, conv = t.cg_state()
        t.cg_finish()
        assert its == K and not conv and 0.0 < rel < 1.0 and t.stats().final_relres == rel
        hist[name] = (rel, st.n_active, st.n_free, st.bytes_device)
        if name == "cg1":
            assert t.true_residual() == pytest.approx(rel, rel=1e-6)
        t.destroy()
    assert hist["cg1"][1:3] == hist["lean"][1:3]
    assert hist["cg1"][0] == pytest.approx(hist["lean"][0], rel=1e-6)
    assert hist["lean"][3] < 230e9 and hist["cg1"][3] < 230e9  # both fit the 288 GB with room for the driver
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-9, precond=oia.PRECOND_MG)
    tau = t.value()
    st = t.stats()
    assert t.status == oia.OI_OK and st.converged and st.iterations < 200 and tau > 2.0
    assert flux_gap(st) <= 1e-6
    t.destroy(); buf.free()


def test_final_sums_of_the_single_kernel_iteration_are_reproducible(oi_option):
    """Large grids sum their per-block partials in 64 slices; the block that draws the last ticket adds the slice sums
    (k_finalize1_sliced).  WHICH block that is depends on timing, what it computes must not: three runs of 40 iterations on a 512^3
    volume (4096 blocks, the smallest grid that takes the sliced form) leave the same residual bit for bit, and the one-block sums
    (`cg1_fin_sliced` = 0, another order of summation) agree to rounding."""
    shape = (512, 512, 512)
    buf = oia.gen_spheres_device(shape, 7, 24, 0.4)
    oi_option("place_trials", 0, reset=48)
    seen = []
    for sliced in (1, 1, 1, 0):
        oi_option("cg1_fin_sliced", sliced, reset=1)
        t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=5000, precond=oia.PRECOND_NONE, kernel=oia.KERNEL_FUSED1)
        t.cg_begin(); t.cg_iterate(41); t.sync()
        its, rel, conv = t.cg_state()
        t.cg_finish()
        assert its == 40 and not conv
        seen.append(rel)
        t.destroy()
    buf.free()
    assert seen[0] == seen[1] == seen[2]
    assert seen[3] == pytest.approx(seen[0], rel=1e-10) and seen[0] < 1.0


def test_vector_placement_search_leaves_the_iterates_alone(oi_option):
    """The single-kernel iteration allocates x and a set of vector-sized slots back to back, times a few placements of its six
    double-buffered vectors in them when the handle is built (on MI355X the iteration time depends on the distances between
    the arrays read and the arrays written) and frees the slots it does not use.  The choice is invisible to the arithmetic:
    the same residual history and the same tau with the search off, with six slots (orderings only) and with spare slots;
    after the search a handle holds the same memory whatever it searched; when the card is nearly full the spare slots are
    given up, not the handle; a second handle of the same shape takes the first one's placement from the process cache."""
    shape = (512, 512, 512)
    K = 30
    buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
    out = {}
    oi_option("place_cache", 0, reset=1)
    oi_option("place_alloc_budget_s", 1e9, reset=0.5)   # (spare slots whatever the driver's clearing costs: their number is asserted)
    for name, trials, slots, spacer in (("off", 0, 12, 48.0), ("six", 6, 6, 0.0), ("spare", 10, 12, 48.0), ("second", 6, 6, 0.0)):
        oi_option("place_trials", trials, reset=48)
        # "second": a rate no kernel reaches -> the first set of allocations counts as slow, a second one is allocated beside it and searched
        oi_option("place_retry_gbps", 1e9 if name == "second" else 0.0, reset=5150.0)
        oi_option("place_slots", slots, reset=12)
        oi_option("place_spacer_gb", spacer, reset=48.0)   # ("spare": a freed spacer between the two halves of the slots)
        t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=5000, precond=oia.PRECOND_NONE, kernel=oia.KERNEL_FUSED1)
        pl = t.placement()
        if name == "off":
            assert pl["tried"] == 0 and pl["ms_kept"] == 0.0 and pl["nslots"] == 0
        else:
            # (twice the trials when the first set of allocations turned out flat and slow and a second one was searched)
            assert pl["nslots"] == slots and pl["tried"] in ((2 * trials,) if name == "second" else (trials, 2 * trials))
            assert 0.0 < pl["ms_kept"] <= pl["ms_in_order"]
            assert len(set(pl["slots"])) == 6 and all(0 <= v < slots for v in pl["slots"]) and pl["x_after_slots"] == 0
            assert not pl["cache_hit"] and pl["search_s"] > 0.0
        t.cg_begin(); t.cg_iterate(K + 1); t.sync()
        its, rel, conv = t.cg_state()
        t.cg_finish()
        assert its == K and not conv
        out[name] = (rel, t.true_residual(), t.stats().bytes_device, pl)
        t.destroy()
    assert out["off"][0] == out["six"][0] == out["spare"][0] == out["second"][0]      # bit-identical recurrence residual
    assert out["off"][1] == out["six"][1] == out["spare"][1] == out["second"][1]      # and iterate (its true residual)
    assert abs(out["second"][2] - out["six"][2]) < (1 << 20)      # whichever set was kept, the other went back whole
    oi_option("place_retry_gbps", 0.0, reset=5150.0)
    # the spare slots went back: the handle holds x and six vectors however many slots it searched (the 2 MiB rounding of
    # the separate allocations aside), and its peak was six slots higher
    assert abs(out["spare"][2] - out["six"][2]) < (1 << 20) and abs(out["six"][2] - out["off"][2]) < (64 << 20)
    slot = out["spare"][3]["kept_vector_bytes"] / 8.05   # (x is about two vectors here)
    assert out["spare"][3]["peak_device_bytes"] > out["spare"][2] + 4 * slot
    # the process cache: the second handle of a shape verifies the first one's placement with one cycle
    oi_option("place_cache", 1, reset=1)
    oi_option("place_trials", 10, reset=48)
    seen = []
    for _ in range(2):
        t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=5000, precond=oia.PRECOND_NONE, kernel=oia.KERNEL_FUSED1)
        seen.append(t.placement())
        t.cg_begin(); t.cg_iterate(K + 1); t.sync()
        assert t.cg_state()[1] == out["off"][0]
        t.cg_finish(); t.destroy()
    assert not seen[0]["cache_hit"] and seen[0]["tried"] >= 10
    if seen[1]["cache_hit"]:   # (a miss -- the verification cycle slower than the cached figure -- searches again: legitimate)
        # (the cached placement is the one the first handle's FIRST set of allocations gave; had that set been flat and slow
        # and a second set been searched and kept, the first handle itself reports the second set's placement)
        assert seen[1]["tried"] == 1 and len(set(seen[1]["slots"])) == 6
    else:
        assert seen[1]["tried"] >= 10
    # a nearly full card: a handle that fits without spare slots is still built, with six slots
    oi_option("place_cache", 0, reset=1)
    oi_option("place_trials", 4, reset=48)
    free, total = oia.device_memory()
    hog = oia.DeviceBuffer(int(free - 0.26 * total))  # leaves 26 % of the card: spare slots would cut into the last quarter
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=5000, precond=oia.PRECOND_NONE, kernel=oia.KERNEL_FUSED1)
    pl = t.placement()
    assert pl["nslots"] == 6 and pl["tried"] in (4, 8)
    t.cg_begin(); t.cg_iterate(K + 1); t.sync()
    assert t.cg_state()[1] == out["off"][0]
    t.cg_finish()
    t.destroy(); hog.free(); buf.free()
