"""Does the slow plateau of vector placements (a set of allocations in which every placement is ~8 % slow) follow a process that used
the whole card?  mode `hog`: allocate and write FRAC of the free memory, exit.  mode `one`: build one 1024^3 single-kernel-CG handle
(verbose: the search's own lines go to stderr), print what was kept and how the kept kernel runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode == "hog":
    import torch
    free, total = torch.cuda.mem_get_info()
    frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.95
    piece = 4 << 30
    keep = []
    for _ in range(int(free * frac) // piece):
        keep.append(torch.empty(piece, dtype=torch.uint8, device="cuda").fill_(7))
    torch.cuda.synchronize()
    print("hog: wrote %.0f GB" % (len(keep) * piece / 1e9), flush=True)
    sys.exit(0)
import openimpala_amd as oia
if len(sys.argv) > 2:
    oia.set_option("place_retry_gbps", float(sys.argv[2]))
n = 1024
shape = (n, n, n)
buf = oia.gen_spheres_device(shape, 20251031, 24, 0.4)
t0 = time.time()
t = oia.TortuosityMI355(buf, 1, oia.Direction.X, 0.0, 1.0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, precond=oia.PRECOND_NONE,
                        maxiter=100000, verbose=2, profile=1)
dt = time.time() - t0
pl = t.placement()
t.cg_begin(); t.cg_iterate(10); t.sync(); t.reset_kernel_times()
t.cg_iterate(40); t.sync()
kt = t.kernel_times()
ms = kt["cg1"][0] / kt["cg1"][1]
print("one: create %.2f s tried %d in-order %.3f kept %.3f | kernel %.3f ms | stream %.0f GB/s" %
      (dt, pl["tried"], pl["ms_in_order"], pl["ms_kept"], ms, 0.0), flush=True)
t.cg_finish(); t.destroy()
