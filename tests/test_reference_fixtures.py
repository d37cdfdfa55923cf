"""The reference's OWN tests for this path, with the reference's own input values and bundled data files, run through
this repository's readers and -- where a GPU is there -- through the HIP path.

These are the only checks the reference itself holds for the path: its tests pin NO tortuosity or
volume-fraction number (tests/inputs/tTortuosity.inputs:28-32 are commented out, tVolumeFraction.inputs:53-54 are -1), so
what they assert is structural -- dimensions, {0, 1} after thresholding, counts equal to a direct recount, VF0 + VF1 = 1,
"tau is finite", checkMatrixProperties, convergence, tensor symmetry -- and exactly that is asserted here.
Data fixtures are byte-identical copies of /root/reference/data/* (data, not source)."""
import math
import os

import numpy as np
import pytest

import openimpala_amd as oia
from conftest import GOLDEN

RAW = os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_uint8.raw")
H5 = os.path.join(GOLDEN, "SampleData_2Phase_3d.hdf5")
TIF = os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif")


# ---- readers: host code, no GPU needed ------------------------------------------------------------------------------
def test_tRawReader_inputs():
    """tests/inputs/tRawReader.inputs: 100 x 100 x 100, UINT8, threshold 0.5; tRawReader.cpp:176-179 (dimensions as given),
    :242-243 (thresholded field is {0, 1} with both values present)."""
    assert os.path.getsize(RAW) == 100 * 100 * 100
    r = oia.RawReader(RAW, 100, 100, 100, "UINT8")
    ph = r.threshold(0.5)
    assert ph.shape == (100, 100, 100) and ph.dtype == np.uint8
    assert ph.min() == 0 and ph.max() == 1
    raw = np.fromfile(RAW, np.uint8).reshape(100, 100, 100)  # x fastest = the reader's (nz, ny, nx)
    assert np.array_equal(ph, (raw > 0.5).astype(np.uint8))
    with pytest.raises(oia.OpenImpalaError):  # RawReader.cpp: file size must match the dimensions
        oia.RawReader(RAW, 100, 100, 99, "UINT8").threshold(0.5)
    with pytest.raises(oia.OpenImpalaError):
        oia.RawReader(RAW, 100, 100, 100, "INT16_LE").threshold(0.5)


def test_tHDF5Reader_inputs():
    """tests/inputs/tHDF5Reader.inputs: dataset "image", threshold 0.5; tHDF5Reader.cpp:88-90,119 (100^3), :167 ({0, 1})."""
    ph = oia.HDF5Reader(H5, "image").threshold(0.5)
    assert ph.shape == (100, 100, 100)
    assert ph.min() == 0 and ph.max() == 1
    with pytest.raises(oia.OpenImpalaError):
        oia.HDF5Reader(H5, "no_such_dataset").threshold(0.5)


def test_tTiffReader_expectations():
    """tTiffReader.cpp:98-103: the bundled stack is 100 x 100 x 100 at 1 bit per sample; thresholding gives {0, 1}."""
    r = oia.TiffReader(TIF)
    ph = r.threshold(0.5)
    assert (r.width(), r.height(), r.depth(), r.bitsPerSample()) == (100, 100, 100, 1)
    assert ph.min() == 0 and ph.max() == 1


# ---- operators: through the HIP path ----------------------------------------------------------------------------------
def _phases():
    return {"tif": oia.TiffReader(TIF).threshold(0.5), "raw": oia.RawReader(RAW, 100, 100, 100, "UINT8").threshold(0.5),
            "hdf5": oia.HDF5Reader(H5, "image").threshold(0.5)}


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["tif", "raw", "hdf5"])
def test_tVolumeFraction_inputs(source):
    """tests/inputs/tVolumeFraction.inputs (phase ids 0 / 1, threshold 0.5, tolerance 1e-9): tVolumeFraction.cpp:441,449 -- the
    class's counts equal a direct recount; :413 -- VF0 + VF1 = 1 within the tolerance.  (expected_vf0/1 = -1: no number.)"""
    ph = _phases()[source]
    c0, tot0 = oia.VolumeFraction(ph, 0).value()
    c1, tot1 = oia.VolumeFraction(ph, 1).value()
    assert tot0 == tot1 == ph.size
    assert c0 == int(np.count_nonzero(ph == 0)) and c1 == int(np.count_nonzero(ph == 1))
    assert abs(1.0 - (c0 / tot0 + c1 / tot1)) <= 1e-9


@pytest.mark.gpu
def test_tTortuosity_inputs():
    """tests/inputs/tTortuosity.inputs: phase 0, X, hypre.eps 1e-10, hypre.maxiter 1000, remspot_passes 0, vlo 0 / vhi 1
    (tTortuosity.cpp:271-301).  The test's own pass criteria: checkMatrixProperties() (:285-292) and a finite
    tortuosity (:304); value() itself applies the 1e-6 flux-conservation gate (TortuosityHypre.cpp:794-823)."""
    ph = oia.TiffReader(TIF).threshold(0.5)
    t = oia.TortuosityMI355(ph, 0, oia.Direction.X, 0.0, 1.0, eps=1e-10, maxiter=1000)
    assert t.checkMatrixProperties()
    tau = t.value()
    st = t.stats()
    assert t.status == oia.OI_OK and math.isfinite(tau) and tau > 0
    assert st.converged and st.iterations <= 1000 and st.final_relres <= 1e-10
    assert abs(abs(st.flux_in) - abs(st.flux_out)) / (0.5 * (abs(st.flux_in) + abs(st.flux_out))) <= 1e-6
    t.destroy()


@pytest.mark.gpu
def test_tEffectiveDiffusivity_inputs():
    """tests/inputs/tEffectiveDiffusivity.inputs: phase 1, threshold 0.5, hypre.eps 1e-9, hypre.maxiter 500.  The reference
    test checks that the three chi_k solves converge; tEffectiveDiffusivity.cpp:423-445 -- tensor symmetric to 1e-7,
    diagonal non-negative."""
    ph = oia.TiffReader(TIF).threshold(0.5)
    h = oia.EffectiveDiffusivityMI355(ph, 1, eps=1e-9, maxiter=500)
    res = [h.solve(k) for k in range(3)]
    assert all(r[0] for r in res) and all(r[1] <= 500 for r in res)
    D = h.tensor()
    assert np.all(np.isfinite(D))
    for a in range(3):
        assert D[a, a] >= 0.0
        for b in range(a + 1, 3):
            assert abs(D[a, b] - D[b, a]) <= 1e-7
    h.destroy()
