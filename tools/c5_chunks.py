"""Chunk lengths and launch orders of k_cg1b on config 5's per-GPU share (4096 x 4096 x 512 blobs at porosity 0.2, Jacobi), one handle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openimpala_amd as oia
shape = (512, 4096, 4096)
oia.set_option("place_trials", 0)
buf = oia.gen_blobs_device(shape, 20251031, 24, 0.2)
t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, maxiter=100000, precond=oia.PRECOND_JACOBI, kernel=oia.KERNEL_FUSED1, profile=1)
t.cg_begin(); t.cg_iterate(4); t.sync()
def run(label):
    t.retune()
    t.cg_iterate(2); t.sync()
    k0 = t.kernel_times()["cg1"]
    t.cg_iterate(8); t.sync()
    k1 = t.kernel_times()["cg1"]
    print(f"{label:44s} {(k1[0] - k0[0]) / (k1[1] - k0[1]):.3f} ms", flush=True)
for rep in range(2):
    for tm in (2, 0, 1):
        for zc in (64, 128, 256, 512):
            oia.set_option("tile_map", tm)
            oia.set_tuning(16, zc, 32, 16)
            run(f"tile_map {tm} zchunk {zc}")
oia.set_tuning(); oia.set_option("tile_map", 2)
