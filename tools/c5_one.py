"""k_cg1b on config 5's per-GPU share (4096 x 4096 x 512 blobs at porosity 0.2, Jacobi), ms per launch (the library comes from OI_LIB_PATH)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openimpala_amd as oia
shape = (512, 4096, 4096)
oia.set_option("place_trials", 0)
buf = oia.gen_blobs_device(shape, 20251031, 24, 0.2)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=100000, precond=(oia.PRECOND_NONE if os.environ.get("OI_C5_NONE") else oia.PRECOND_JACOBI), kernel=oia.KERNEL_FUSED1, profile=1)
t.cg_begin(); t.cg_iterate(6); t.sync(); t.reset_kernel_times()
t.cg_iterate(16); t.sync()
k = t.kernel_times()["cg1"]
print(os.path.basename(os.environ.get("OI_LIB_PATH", "default")), f"{k[0] / k[1]:.3f} ms", flush=True)
