"""apps/CathodeWrite.H (host text exporters, reference src/io/CathodeWrite.cpp): byte-identical files to the oracle's restatement
of the reference's two formats, for the reference's default parameters and for awkward ones.  CPU only."""
import os
import subprocess

import pytest

from conftest import ROOT
from oracle import oracle as o


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cathode") / "cathode_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "csrc", "cathode_driver.cpp"), "-o", exe])
    return exe


CASES = [(0.5, 5e-6, 100.0, 51000.0),            # CathodeParams' defaults (CathodeWrite.H:21-35)
         (0.398309, 2.5e-6, 0.18, 49340.0),      # the bundled sample's volume fraction
         (1.0, 1.0, 1e-12, 1e12), (0.0, 3.3e-9, 123456.789, 0.5),
         (0.25, 0.0, 1.0, 1.0), (0.25, 1e-17, 1.0, 1.0)]   # vanishing radius: area 0 and a warning (:52-57)


@pytest.mark.parametrize("vf,radius,sigma,cmax", CASES)
def test_files_equal_the_reference_formats(driver, tmp_path, vf, radius, sigma, cmax):
    dl, pb = tmp_path / "cathode.txt", tmp_path / "cathode.csv"
    r = subprocess.run([driver, repr(vf), repr(radius), repr(sigma), repr(cmax), str(dl), str(pb)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert dl.read_text() == o.cathode_dandeliion_text(vf, radius, sigma, cmax)
    assert pb.read_text() == o.cathode_pybamm_text(vf, radius, sigma, cmax)
    assert f"DandeLiion parameters written to: {dl}" in r.stdout and f"PyBamm parameters written to: {pb}" in r.stdout
    assert ("near zero" in r.stderr) == (radius <= 2.3e-16)


def test_a_file_that_cannot_be_opened_is_reported_not_fatal(driver, tmp_path):
    r = subprocess.run([driver, "0.5", "5e-6", "100", "51000", str(tmp_path / "no" / "such" / "dir.txt"), str(tmp_path / "ok.csv")],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "Could not open file" in r.stdout and (tmp_path / "ok.csv").exists()
