"""Per kernel: HBM read requests by size (TCC_EA0_RDREQ_32B / _64B / _128B, rocprofv3 --pmc) and the bytes they stand for.
usage: python tools/rdreq_summary.py <output dir of tools/rocprof_rdreq.sh>"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import short, bench_name


def main(out):
    f = glob.glob(os.path.join(out, "pmc_rdreq", "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
    print("%-62s %7s %12s %12s %12s %12s %14s %14s" % ("kernel", "calls", "all", "32 B", "64 B", "128 B", "bytes/launch", "FETCHx2 rule"))
    res = {}
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("TCC_EA0_RDREQ_sum", 0))[:24]:
        n = max(1, len(launches[k]))
        a, n32, n64, n128 = (c.get(x, 0.0) / n for x in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
        other = a - n32 - n64 - n128
        b = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(0.0, other)
        rule = 2.0 * (32 * n32 + 64 * (a - n32))   # what FETCH_SIZE x 2 reports (TCC_BUBBLE = 0 assumed)
        print("%-62s %7d %12.0f %12.0f %12.0f %12.0f %14.0f %14.0f" % (k[:62], n, a, n32, n64, n128, b, rule))
        bn = bench_name(k)
        if bn:
            res.setdefault(bn, {"read_bytes_per_launch": 0.0, "n32": 0.0, "n64": 0.0, "n128": 0.0, "calls": 0})
            e = res[bn]
            # (several instantiations behind one key: launch-weighted mean)
            w0, w1 = e["calls"], n
            for key, v in (("read_bytes_per_launch", b), ("n32", n32), ("n64", n64), ("n128", n128)):
                e[key] = (e[key] * w0 + v * w1) / (w0 + w1)
            e["calls"] = w0 + w1
    json.dump(res, open(os.path.join(out, "rdreq_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
