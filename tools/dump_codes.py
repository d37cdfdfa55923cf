"""Dump the code bytes (matrix) of the bench structure for tools/stream_bench_masked.hip.
usage: python tools/dump_codes.py <size> <out file>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia
n = int(sys.argv[1])
buf = oia.gen_spheres_device((n, n, n), 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, 0.0, 1.0, shape=(n, n, n))
t.codes().tofile(sys.argv[2])
