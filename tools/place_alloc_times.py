"""Per-allocation times of the vector set of a 1024^3 single-kernel handle (verbose = 3), optionally after this process's
predecessor has dirtied the card:  python tools/place_alloc_times.py dirty   allocates and writes 240 GB, then exits."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openimpala_amd as oia
if len(sys.argv) > 1 and sys.argv[1] == "dirty":
    bufs = [oia.gen_bernoulli_device((1024, 1024, 1024 * 8), 1 + k, 0.5) for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 28)]   # N x 8.6 GB, every byte written
    print("dirtied", sum(1 for _ in bufs) * 8.6, "GB; free now", oia.device_memory()[0] / 1e9)
    sys.exit(0)
n = 1024
buf = oia.gen_spheres_device((n, n, n), 20251031, 24, 0.4)
t = oia.TortuosityMI355(buf, 1, 0, 0.0, 1.0, shape=(n, n, n), eps=1e-300, kernel=oia.KERNEL_FUSED1, precond=oia.PRECOND_NONE, maxiter=100, verbose=3)
print(t.placement())
