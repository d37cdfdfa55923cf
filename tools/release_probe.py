"""Free device memory around every handle kind, cycle by cycle (the record behind tests/test_zz_robustness_gpu.py):
    python tools/release_probe.py [cycles] [threads-first]
Prints, per cycle, the change of hipMemGetInfo's free figure over each variant's create-solve-destroy, in MiB."""
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import openimpala_amd as oia  # noqa: E402
from test_zz_robustness_gpu import RELEASE_VARIANTS, release_cycle  # noqa: E402

ncycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
threads_first = len(sys.argv) > 2 and sys.argv[2] == "threads"
tif = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "SampleData_2Phase_stack_3d_1bit.tif")
sample = oia.TiffReader(tif).threshold(0.5)
sub = np.ascontiguousarray(sample[:96, :90, :88])
print("free at start (MiB):", oia.device_memory()[0] >> 20)
if threads_first:  # what preceded the failing test in GPUTEST_r02: six handles from six host threads
    def work(d, precond):
        t = oia.TortuosityMI355(sample, 1, d, -1.0, 1.0, eps=1e-8, precond=precond)
        t.value(); t.destroy()
    th = [threading.Thread(target=work, args=(d, p)) for d in range(3) for p in (oia.PRECOND_MG, oia.PRECOND_JACOBI)]
    [t.start() for t in th]; [t.join() for t in th]
    print("free after six threaded handles:", oia.device_memory()[0] >> 20)
names = [n for n, _ in RELEASE_VARIANTS] + ["effdiff"]
print("cycle  free  " + "  ".join(f"{n:>11s}" for n in names))
for c in range(ncycles):
    d = {}
    release_cycle(sub, d)
    print(f"{c:5d} {oia.device_memory()[0] >> 20:6d} " + "  ".join(f"{d[n][0] / 2**20:11.1f}" for n in names), flush=True)
