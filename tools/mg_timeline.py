"""Where the wall time of a multigrid-PCG solve at 1024^3 goes on the host's clock: cg_begin, every batch of iterations, cg_finish
(profile = 0 and 1, batches of 4 / 8 / 32)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openimpala_amd as oia
n = 1024
buf = oia.gen_spheres_device((n, n, n), 20251031, 24, 0.4)
for profile, batch in ((1, 4), (0, 4), (0, 8), (0, 32), (1, 4), (0, 4)):
    tm = oia.TortuosityMI355(buf, 1, oia.Direction.X, 0.0, 1.0, shape=(n, n, n), eps=1e-6, precond=oia.PRECOND_MG, profile=profile, maxiter=1000)
    tm.sync()
    marks = [("start", time.perf_counter())]
    tm.cg_begin(); tm.sync(); marks.append(("begin", time.perf_counter()))
    done, its = False, 0
    while not done and its < 1000:
        tm.cg_iterate(batch)
        its, rel, done = tm.cg_state()
        marks.append((f"its {its}", time.perf_counter()))
    tm.cg_finish(); tm.sync(); marks.append(("finish", time.perf_counter()))
    t0 = marks[0][1]
    print(f"profile {profile} batch {batch}: total {marks[-1][1] - t0:.4f} s |", " ".join(f"{k} +{(b - a) * 1e3:.1f}" for (k, b), (_, a) in zip(marks[1:], marks[:-1])), flush=True)
    tm.destroy()
