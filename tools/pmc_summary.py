"""Summarise a tools_rocprof.sh output directory: per-kernel time stats and HBM bytes per
launch from the FETCH_SIZE / WRITE_SIZE passes (gfx950: FETCH_SIZE counts 64 B per 128-B
request for wide coalesced reads -> doubled, as MI355X_MICROARCH.md prescribes; both
counters are in KiB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0] if not name.startswith("void oi::k_cg1b<") and not name.startswith("oi::k_cg1b<") else name[:name.index(">(") + 1] if ">(" in name else name
    for tok in ("void ", "oi::"):
        name = name.replace(tok, "")
    return name.strip()


def bench_name(kernel):
    """rocprof kernel name -> the key bench.py uses (k_cg1b by its FORM template argument: 0 the single-kernel CG iteration, 1 / 2 / 3
    the fine kernels of the multigrid-preconditioned iteration -- FORM 3 with (mg_form3x) and without (mg_form3) the pairwise x update)"""
    if kernel.startswith("k_cg1b<"):
        args = [a.strip() for a in kernel[len("k_cg1b<"):].rstrip(">").split(",")]
        form = int(args[5]) if len(args) > 5 and args[5].lstrip("-").isdigit() else 0
        nox = len(args) > 9 and args[9] == "true"
        if form == 0:
            return "cg1"
        if form == 3:
            return "mg_form3" if nox else "mg_form3x"
        return "mg_form%d" % form
    if kernel.startswith("k_cg1"):
        return "cg1"
    if kernel.startswith("k_mg_") and not kernel.startswith(("k_mg_fine", "k_mg_dot", "k_mg_build", "k_mg_compact")):
        return "mg_coarse"   # reported PER ITERATION (all coarse-level launches of one cycle), see main()
    if kernel.startswith("k_apply7"):
        return "apply"
    if kernel.startswith("k_fused_a"):
        if ", true" not in kernel:
            return "fused_a"
        return "apply_lds" if ", 0, " in kernel else "apply_cgvec"  # LEAN = 3: the apply on a CG vector
    if kernel.startswith("k_update_xr"):
        return "update_r" if kernel.startswith("k_update_xr<false") else "update_xr"
    if kernel.startswith("k_apply<") or kernel == "k_apply":
        return "apply_direct"
    if kernel.startswith("k_update_p"):
        return "update_p"
    return None


def main(out):
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if stats:
        print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
        rows = list(csv.DictReader(open(stats)))
        for r in rows[:14]:
            print("%-64s calls %6s  avg %10.1f us  total %8.2f ms  %5.1f%%" % (
                short(r["Name"])[:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6,
                float(r["Percentage"])))
    # the CG kernel's launches in order: the first ones belong to the placement search (two per placement tried, at
    # placements of every speed), the LAST `steps` ones are bench.py's timed region -- the figure its line reports
    trace = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    line = os.path.join(out, "bench_trace.json")
    if trace and os.path.exists(line):
        import json
        try:
            bl = json.loads(open(line).read().strip().splitlines()[-1])
            steps = int(bl["steps"])
            d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(trace))
                 if bench_name(short(r["Kernel_Name"])) == "cg1"]
            d.sort()
            last = [x[1] for x in d[-steps:]]
            if last:
                print("== CG kernel, the last %d launches (the timed region): rocprofv3 avg %.1f us; bench.py's HIP events of the same run "
                      "%.1f us; all %d launches avg %.1f us ==" % (len(last), sum(last) / len(last) / 1e3, bl["roofline"]["ms_per_launch"] * 1e3,
                                                                  len(d), sum(x[1] for x in d) / len(d) / 1e3))
        except (KeyError, ValueError, IndexError):
            pass
    traffic = defaultdict(float)
    for tag, counter, scale in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
        f = find(os.path.join(out, tag), "*counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
        print(f"== {counter} per launch (KiB -> bytes, x{scale:g} gfx950 correction) ==")
        for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:14]:
            print("%-72s launches %6d  %14.0f bytes/launch" % (k[:72], n, v / n * 1024.0 * scale))
        # per bench key: bytes per launch, averaged over ALL launches of the kernels behind the key; the coarse levels of the multigrid
        # cycle: bytes per ITERATION = the sum over every coarse-level launch / the number of FORM 2 launches (one per iteration)
        per_key = defaultdict(lambda: [0.0, 0])
        for k, (v, n) in acc.items():
            b = bench_name(k)
            if b:
                per_key[b][0] += v
                per_key[b][1] += n
        iters = per_key["mg_form2"][1] if "mg_form2" in per_key else 0
        for b, (v, n) in per_key.items():
            if b == "mg_coarse":
                if iters:
                    traffic[b] += v / iters * 1024.0 * scale
            elif n:
                traffic[b] += v / n * 1024.0 * scale
    if traffic:
        import hashlib
        import json
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        import bench
        sha = bench.kernels_sha()   # (one definition: every device source of the product library)
        tagname = os.environ.get("OI_PMC_KEY", "1024:1")
        js = {f"{k}:{tagname}": {"hbm_bytes_per_launch": v, "kernels_sha256": sha, "method": "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, "
                                 "separate passes, KiB -> bytes, mean over launches"} for k, v in traffic.items()}
        with open(os.path.join(out, "pmc_summary.json"), "w") as f:
            json.dump(js, f, indent=1)
        print("== HBM bytes per launch (FETCH x2 + WRITE) ==")
        for k, v in sorted(traffic.items()):
            print("%-12s %14.0f" % (k, v))


if __name__ == "__main__":
    main(sys.argv[1])
