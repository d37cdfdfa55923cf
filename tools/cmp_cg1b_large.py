"""k_cg1b against k_cg1 at 512^3 (spheres) and on a ragged 300 x 260 x 200 blob structure: K iterations, same residual history
and the same x bit for bit; fp64 and fp32-stored p, Jacobi on and off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openimpala_amd as oia
oia.set_option("place_trials", 0)
bad = 0
for shape, gen in (((512, 512, 512), "spheres"), ((200, 260, 300), "blobs")):
    buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4) if gen == "spheres" else oia.gen_blobs_device(shape, 7, 16, 0.3)
    for pf32 in (0, 1):
        for jac in (False, True):
            for rows in (16, 32):
                res = {}
                for kern in (1, 0):
                    oia.set_option("cg1_kernel", kern)
                    oia.set_tuning(16, 0, 32, rows)
                    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, kernel=oia.KERNEL_FUSED1, p_storage=pf32, eps=1e-300, maxiter=500,
                                            precond=oia.PRECOND_JACOBI if jac else oia.PRECOND_NONE)
                    t.cg_begin(); t.cg_iterate(61); t.sync()
                    its, rel, conv = t.cg_state()
                    t.cg_finish()
                    res[kern] = (its, rel, t.solution().copy())
                    t.destroy()
                same = res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
                bad += not same
                print(shape, gen, "pf32", pf32, "jacobi", jac, "rows", rows, res[0][:2], "IDENTICAL" if same else "DIFFERENT", flush=True)
    buf.free()
oia.set_tuning(); oia.set_option("cg1_kernel", 0)
print("differing cases:", bad)
sys.exit(1 if bad else 0)
