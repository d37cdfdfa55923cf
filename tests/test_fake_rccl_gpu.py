"""RcclComm (oi_comm.cpp: the transport of the one-process-per-GPU runs) with MORE THAN ONE RANK, on a one-GPU box.

No multi-GPU node has been available to any round and RCCL refuses two ranks on one device, so the RCCL legs of the slab paths had
only ever run with one rank.  Here a child process loads a TEST DOUBLE of the RCCL entry points (tests/csrc/fake_rccl.cpp:
in-process ranks = host threads, NCCL's matching rules for grouped send / receive enforced, a dead-locked call pattern turned into
an error after 30 s) with RTLD_GLOBAL BEFORE the product library, whose calls into librccl then resolve to it.  The UNCHANGED
RcclComm code -- grouped ghost-plane exchanges of dense and compacted (variable-length) planes on an open chain and on a ring,
skipped zero-length messages, the split communicator, the scalar and the large all-reduces -- runs with 2, 3 and 4 ranks behind
oi_tort_create_slab / oi_effdiff_create_slab, and the answers equal the single-domain ones -- once with a double that completes every
call before it returns and once with one whose send / receive are stream-ordered by events only, like the real library's.  What this
cannot show: the collectives' asynchrony, RCCL's own kernels and resource use, xGMI."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import ctypes, os, sys, threading
fake = ctypes.CDLL(sys.argv[1], mode=ctypes.RTLD_GLOBAL)      # BEFORE the product library: its nccl* calls resolve here
sys.path.insert(0, sys.argv[2])
import numpy as np
import openimpala_amd as oia
from oracle import oracle as o

assert fake.fake_rccl_present() == 1
assert oia.Comm.unique_id().startswith(b"fake-rccl-"), "librccl answered, not the double"

def ranks(n, work):
    """n host threads = n RCCL ranks on device 0; returns what work(rank, comm) returned"""
    uid = oia.Comm.unique_id()
    out, err = [None] * n, []
    def run(r):
        try:
            c = oia.Comm.create(uid, n, r, 0)        # blocks until every rank has joined (ncclCommInitRank + ncclCommSplit)
            try:
                info = c.info()
                assert info["kind"] == "rccl" and info["ranks_collective"] == n and info["ranks_p2p"] == n and info["exchange_overlaps"]
                out[r] = work(r, c)
            finally:
                c.destroy()
        except Exception as e:
            err.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(n)]
    [t.start() for t in th]
    [t.join(timeout=240) for t in th]
    assert not any(t.is_alive() for t in th), "rank threads hung"
    assert not err, err
    return out

phase = o.gen_spheres((36, 40, 44), 21, 5, 0.45)
nz = phase.shape[0]
checked = 0
# ---- flow-through path: open chain of z-slabs ----
for direction in (0, 2):
    for name, kw in (("multigrid-PCG (default)", dict()), ("single-kernel CG, compacted planes, split launches", dict(kernel=oia.KERNEL_FUSED1)),
                     ("Jacobi two-kernel, dense planes", dict(kernel=oia.KERNEL_FUSED, precond=oia.PRECOND_JACOBI)),
                     ("unfused", dict(kernel=oia.KERNEL_UNFUSED, precond=oia.PRECOND_NONE))):
        single = oia.TortuosityMI355(phase, 1, direction, -1.0, 1.0, eps=1e-11, **kw)
        tau1, it1, mask1, x1 = single.value(), single.getSolverIterations(), single.mask(), single.solution()
        assert single.status == oia.OI_OK, (name, direction, single.status, single.stats().final_relres)
        single.destroy()
        for n in (2, 3, 4):
            def work(r, c):
                z0, z1 = oia.slab_range(nz, n, r)
                t = oia.TortuosityMI355(np.ascontiguousarray(phase[z0:z1]), 1, direction, -1.0, 1.0, eps=1e-11, comm=c, z0=z0, nz_global=nz, **kw)
                ok = t.checkMatrixProperties()
                res = (t.value(), t.getSolverIterations(), t.mask(), t.solution(), ok, t.status, t.stats().final_relres)
                t.destroy()
                return res
            res = ranks(n, work)
            assert all(r[4] and r[5] == oia.OI_OK for r in res), (name, direction, n, [(r[4], r[5], r[1], r[6]) for r in res], it1)
            assert all(r[0] == res[0][0] for r in res), (name, n)                      # every rank the same global answer
            assert abs(res[0][0] - tau1) <= 1e-9 * abs(tau1), (name, direction, n, res[0][0], tau1)
            assert all(abs(r[1] - it1) <= 2 for r in res), (name, n, [r[1] for r in res], it1)
            assert np.array_equal(np.concatenate([r[2] for r in res]), mask1), (name, n)
            np.testing.assert_allclose(np.concatenate([r[3] for r in res]), x1, rtol=0, atol=2e-9)
            checked += 1
# ---- homogenization: a RING of z-slabs (periodic z), dense exchange (the default over RCCL) and the compacted one ----
sub = np.ascontiguousarray(phase[:32, :36, :40])
single = oia.EffectiveDiffusivityMI355(sub, 1, eps=1e-10, precond=oia.PRECOND_MG)
ref = [single.solve(k) for k in range(3)]
D1 = single.tensor().copy()
single.destroy()
for lines in (0, 1):
    oia.set_option("ring_lines_rccl", lines)
    for n in (2, 3, 4):
        def work(r, c):
            z0, z1 = oia.slab_range(sub.shape[0], n, r)
            h = oia.EffectiveDiffusivityMI355(np.ascontiguousarray(sub[z0:z1]), 1, eps=1e-10, precond=oia.PRECOND_MG, comm=c, z0=z0, nz_global=sub.shape[0])
            conv = [h.solve(k) for k in range(3)]
            D = h.tensor().copy()
            lay = h.layout()
            h.destroy()
            return conv, D, lay
        res = ranks(n, work)
        for conv, D, lay in res:
            assert all(cv[0] for cv in conv), (lines, n)
            np.testing.assert_allclose(D, D1, rtol=0, atol=5e-9)
        # the compacted ring exchange runs where every slab starts on an even plane; otherwise (and with the option off) dense vectors
        even = all(oia.slab_range(sub.shape[0], n, r)[0] % 2 == 0 for r in range(n))
        assert all(bool(lay[0]) == (bool(lines) and even) for _, _, lay in res), (lines, n, [l for _, _, l in res])
        checked += 1
oia.set_option("ring_lines_rccl", 0)
print("fake-rccl ok:", checked, "multi-rank configurations")
'''


@pytest.fixture(scope="module")
def fake_lib(tmp_path_factory):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc to build the RCCL test double")
    lib = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "csrc", "fake_rccl.cpp"), "-o", lib])
    return lib


@pytest.mark.parametrize("asynchronous", [0, 1])
def test_rccl_transport_with_two_three_and_four_ranks_through_the_test_double(fake_lib, asynchronous):
    """asynchronous = 1: the double's send / receive are enqueued on the caller's streams and ordered by events only (as in the real
    library), so the solver's own event waits between the exchange stream and the kernels' streams are what keeps the planes current."""
    p = subprocess.run([sys.executable, "-c", CHILD, fake_lib, ROOT], cwd=ROOT, capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, FAKE_RCCL_ASYNC=str(asynchronous)))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "fake-rccl ok: 30 multi-rank configurations" in p.stdout, p.stdout[-1000:]
