"""k_cg1b against k_cg1 on the bundled sample: same iterates, bit for bit (tile shapes, fp32-stored p, Jacobi, z chunks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openimpala_amd as oia
from oracle import oracle as o
sample = o.read_tiff_phase(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/SampleData_2Phase_stack_3d_1bit.tif"))
debug = len(sys.argv) > 1
xs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cases = [(16, 1, False, 0), (16, 0, False, 0)] if debug else [(r, f, j, z) for r in (8, 16, 32) for f in (0, 1) for j in (False, True) for z in (0, 5)]
for rows, pf32, jac, zchunk in cases:
    for n in ((2, 3, 4, 5) if debug else (41,)):
        res = {}
        for kern in (1, 0):
            oia.set_option("cg1_kernel", kern)
            oia.set_tuning(16, zchunk, 32, rows, xs)
            t = oia.TortuosityMI355(sample, 1, 2, 0.0, 1.0, kernel=oia.KERNEL_FUSED1, p_storage=pf32, eps=1e-300, maxiter=500,
                                    precond=oia.PRECOND_JACOBI if jac else oia.PRECOND_NONE)
            t.cg_begin(); t.cg_iterate(n); t.sync()
            its, rel, conv = t.cg_state()
            t.cg_finish()
            res[kern] = (its, rel, t.solution().copy())
            t.destroy()
        same = res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
        print(rows, pf32, jac, zchunk, n, res[1][:2], res[0][:2], "IDENTICAL" if same else "DIFFERENT max|dx| %.3e" % np.abs(res[0][2] - res[1][2]).max())
        if debug and not same:
            d = np.argwhere(res[0][2] != res[1][2])
            print("  differing voxels:", len(d), "z", np.unique(d[:, 0])[:12], "y%16", np.unique(d[:, 1] % 16), "x%16", np.unique(d[:, 2] % 16))
            z, y, x = d[0]
            print("  first", d[0], res[1][2][z, y, x], res[0][2][z, y, x])
oia.set_tuning(); oia.set_option("cg1_kernel", 0)
