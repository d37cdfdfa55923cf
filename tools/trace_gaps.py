"""Where the wall time of the multigrid-PCG solve goes on the DEVICE's clock: from a rocprofv3 --kernel-trace CSV, the window from the
first FORM 2 launch (k_cg1b<..., 2, ...>: launch A of iteration 0) to the last kernel of the solve -- busy time per kernel family,
idle time between kernels, the largest gaps and what surrounds them.
usage: python tools/trace_gaps.py <dir or kernel_trace.csv> [per-iteration]"""
import csv, glob, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short, bench_name


def main(path):
    if os.path.isdir(path):
        c = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
        path = c[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
    rows.sort()
    f2 = [i for i, r in enumerate(rows) if bench_name(r[2]) == "mg_form2"]
    if not f2:
        print("no FORM 2 launch in the trace"); return
    # the solves of the run: FORM 2 launches closer than 100 ms belong to one solve
    solves, cur = [], [f2[0]]
    for i in f2[1:]:
        if rows[i][0] - rows[cur[-1]][0] > 100e6:
            solves.append(cur); cur = []
        cur.append(i)
    solves.append(cur)
    for sidx, sv in enumerate(solves):
        a = sv[0]
        # the end: the last kernel that starts within 20 ms of the last FORM 2's end
        b = sv[-1]
        while b + 1 < len(rows) and rows[b + 1][0] - rows[b][1] < 20e6:
            b += 1
        # walk back to the start of cg_begin: kernels (memsets do not show) right before the first FORM 2, within 30 ms
        a0 = a
        while a0 > 0 and rows[a0][0] - rows[a0 - 1][1] < 5e6 and rows[a][0] - rows[a0 - 1][0] < 30e6:
            a0 -= 1
        win = rows[a0:b + 1]
        span = (win[-1][1] - win[0][0]) / 1e6
        busy = defaultdict(lambda: [0.0, 0])
        for s, e, k in win:
            key = bench_name(k) or k.split("<")[0]
            busy[key][0] += (e - s) / 1e6
            busy[key][1] += 1
        tot_busy = sum(v[0] for v in busy.values())
        gaps = [((win[i + 1][0] - win[i][1]) / 1e3, win[i][2], win[i + 1][2]) for i in range(len(win) - 1)]
        idle = sum(max(0.0, g[0]) for g in gaps) / 1e3
        print(f"== solve {sidx}: {len(sv)} FORM 2 launches, window {span:.2f} ms, kernels busy {tot_busy:.2f} ms, idle between kernels {idle:.2f} ms "
              f"({len(win)} launches) ==")
        for k, (ms, n) in sorted(busy.items(), key=lambda kv: -kv[1][0])[:16]:
            print(f"   {k[:60]:60s} {n:6d} launches {ms:9.3f} ms  ({ms / max(1, len(sv)):.3f} per iteration)")
        # idle by what FOLLOWS the gap
        after = defaultdict(lambda: [0.0, 0])
        for g, k0, k1 in gaps:
            if g > 0:
                key = bench_name(k1) or k1.split("<")[0]
                after[key][0] += g / 1e3
                after[key][1] += 1
        print("   idle time by the kernel that follows the gap:")
        for k, (ms, n) in sorted(after.items(), key=lambda kv: -kv[1][0])[:10]:
            print(f"      before {k[:50]:50s} {n:6d} gaps {ms:8.3f} ms (avg {ms / n * 1e3:7.1f} us)")
        print("   largest gaps:")
        for g, k0, k1 in sorted(gaps, key=lambda t: -t[0])[:8]:
            print(f"      {g:9.1f} us  after {k0[:48]:48s} before {k1[:48]}")


if __name__ == "__main__":
    main(sys.argv[1])
