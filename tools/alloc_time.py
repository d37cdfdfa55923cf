import time, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import openimpala_amd as oia, ctypes
for gb in (1, 8, 8, 8):
    t0 = time.perf_counter()
    b = oia.DeviceBuffer(gb << 30)
    t1 = time.perf_counter()
    b.free()
    t2 = time.perf_counter()
    print(f"hipMalloc {gb} GiB: {1e3*(t1-t0):.1f} ms, free {1e3*(t2-t1):.1f} ms", flush=True)
