import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sample_phase():
    from oracle import oracle as o
    return o.read_tiff_phase(os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif"))


@pytest.fixture(scope="session")
def goldens():
    import json
    with open(os.path.join(GOLDEN, "sample_goldens.json")) as f:
        return json.load(f)


@pytest.fixture
def oi_option():
    """Set a named process default of the library (oi_tort_set_option) for one test; `reset` is restored afterwards."""
    import openimpala_amd as oia
    changed = []

    def setter(key, value, reset=0):
        oia.set_option(key, value)
        changed.append((key, reset))

    yield setter
    for key, reset in changed:
        oia.set_option(key, reset)


def pytest_collection_modifyitems(config, items):
    """Robustness tests (tests/test_zz_*: misuse, threads, out-of-memory, release) run after every parity test whatever the
    order of the command line's arguments: under `-x` a failure there must not hide a parity row."""
    last = [it for it in items if os.path.basename(str(it.fspath)).startswith("test_zz_")]
    if last:
        first = [it for it in items if it not in last]
        items[:] = first + last
