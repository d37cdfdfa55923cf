"""The superseded kernels kept as bit-identity references of the shipped ones -- k_cg1 (the branching first form of the single-kernel
CG iteration: dense vectors, option cg1_kernel = 1) and k_fused_a on compacted vectors (option fused_kernel = 1) -- are not in the
product library.  They are built into libopenimpala_amd_xcheck.so (`make xcheck`, -DOI_XCHECK=1: the same sources plus
oi_k_xcheck.hip), and this file re-runs the tests that compare the shipped kernels with them in a child process that loads THAT
library (OI_LIB_PATH).  The parent process -- and with it every other test, the benchmark and the application -- stays on the
product library."""
import os
import subprocess
import sys

import pytest

import openimpala_amd as oia

from conftest import ROOT

XLIB = os.path.join(ROOT, "openimpala_amd", "libopenimpala_amd_xcheck.so")
# tests whose cross-check halves need the test library (they skip or shorten themselves on the product library)
XCHECK_K = ("test_single_kernel_layouts_compacted_and_dense or test_single_kernel_iteration_two_kernels_are_bit_identical or "
            "test_multigrid_rotated_iteration_and_its_cross_checks or test_cache_policy_does_not_touch_the_arithmetic")


@pytest.mark.gpu
@pytest.mark.skipif(oia.has_xcheck(), reason="this IS the test library")
def test_shipped_kernels_equal_the_superseded_ones_bit_for_bit():
    assert os.path.exists(XLIB), "make -C openimpala_amd/csrc xcheck (part of __graft_entry__.build())"
    env = dict(os.environ, OI_LIB_PATH=XLIB)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", XCHECK_K], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = p.stdout[-3000:] + p.stderr[-1000:]
    assert p.returncode == 0, tail
    # every selected test RAN there (none skipped): 1 + 5 + 2 + 1
    assert " passed" in p.stdout and "skipped" not in p.stdout.splitlines()[-1], tail


@pytest.mark.gpu
@pytest.mark.skipif(oia.has_xcheck(), reason="product library only")
def test_the_product_library_refuses_what_needs_a_cross_check_kernel(oi_option):
    import numpy as np
    ph = np.ones((8, 8, 8), np.uint8)
    for key in ("cg1_kernel", "fused_kernel"):
        with pytest.raises(oia.OpenImpalaError) as e:
            oia.set_option(key, 1)
        assert e.value.status == oia.OI_BAD_ARG and "cross-check" in str(e.value)
        oia.set_option(key, 0)
    try:
        oia.set_cg1_layout(0)   # dense vectors: fine for the multigrid-preconditioned and the two-kernel iteration ...
        t = oia.TortuosityMI355(ph, 1, 0)
        assert 0.8 < t.value() <= 1.0   # (a full cube: tau = (N - 1) / N)
        t.destroy()
        with pytest.raises(oia.OpenImpalaError) as e:   # ... the single-kernel iteration would need k_cg1
            oia.TortuosityMI355(ph, 1, 0, kernel=oia.KERNEL_FUSED1)
        assert e.value.status == oia.OI_BAD_ARG and "k_cg1" in str(e.value)
    finally:
        oia.set_cg1_layout(1)
