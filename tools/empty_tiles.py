"""Fraction of 128 x 16 tile planes without a free voxel (and with none in the 1-voxel ring around them) in the bench structures."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openimpala_amd as oia
for name, shape, mk in (("spheres 0.4", (256, 1024, 1024), lambda s: oia.gen_spheres_device(s, 20251031, 24, 0.4)),
                        ("blobs 0.2", (256, 1024, 1024), lambda s: oia.gen_blobs_device(s, 20251031, 24, 0.2))):
    buf = mk(shape)
    ph = buf.download(np.uint8, shape)
    buf.free()
    nz, ny, nx = shape
    pore = ph == 1
    for by in (16, 32):
        t = pore.reshape(nz, ny // by, by, nx // 128, 128).any(axis=(2, 4))          # tile planes with a pore voxel
        # with the halo ring: dilate by one tile in y/x is too coarse; use the voxel ring
        d = pore.copy()
        d[:, 1:, :] |= pore[:, :-1, :]; d[:, :-1, :] |= pore[:, 1:, :]
        d[:, :, 1:] |= pore[:, :, :-1]; d[:, :, :-1] |= pore[:, :, 1:]
        tr = d.reshape(nz, ny // by, by, nx // 128, 128).any(axis=(2, 4))
        e1 = 1.0 - tr.mean()
        e3 = (~tr[:-2] & ~tr[1:-1] & ~tr[2:]).mean()
        lines = pore.reshape(nz, ny, nx // 8, 8).any(axis=3)
        seg = lines.reshape(nz, ny, nx // 128, 16).any(axis=3)
        print(f"{name}: porosity {pore.mean():.3f}, live lines {lines.mean():.3f}, live 128-voxel row segments {seg.mean():.3f}; "
              f"{by}-row tiles: empty tile planes (with ring) {e1:.3f}, three in a row {e3:.3f}")
