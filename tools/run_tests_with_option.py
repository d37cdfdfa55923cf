"""Runs pytest in THIS process after setting a process default of the library (tests that set the same option themselves override it):
python tools/run_tests_with_option.py <option> <value> [pytest args...]   e.g.  mg_coarse_tiled 2 tests/test_gpu_slabs.py -m gpu -q"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pytest
import openimpala_amd as oia
oia.set_option(sys.argv[1], float(sys.argv[2]))
sys.exit(pytest.main(sys.argv[3:]))
