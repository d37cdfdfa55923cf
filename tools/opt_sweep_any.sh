#!/bin/bash
# cache policy and chunk length on the OTHER BASELINE structures (one handle each, alternating): config 2 (512^3 Bernoulli 0.5) and a
# 768^3 blob pack at porosity 0.2 -- do round 4's defaults (cg1_nt = 1, 24-plane chunks) hold away from the 1024^3 sphere pack?
cd "$(dirname "$0")/.."
python - <<'PY'
import sys
sys.path.insert(0, ".")
import openimpala_amd as oia
def sweep(buf, shape, label, key, vals):
    t = oia.TortuosityMI355(buf, 1, 0, shape=shape, eps=1e-300, kernel=oia.KERNEL_FUSED1, profile=1, maxiter=100000)
    t.cg_begin(); t.cg_iterate(10); t.sync()
    res = {v: [] for v in vals}
    for rep in range(3):
        for v in vals:
            oia.set_option(key, v); t.retune()
            t.cg_iterate(4); t.sync(); t.reset_kernel_times()
            t.cg_iterate(30); t.sync()
            kt = t.kernel_times()
            res[v].append(kt["cg1"][0] / kt["cg1"][1])
    for v in vals:
        print(f"{label}  {key} = {v:g}: " + " ".join(f"{x:.3f}" for x in res[v]) + f"  | median {sorted(res[v])[1]:.3f} ms", flush=True)
    oia.set_option(key, vals[0]); t.destroy()
n = 512
b = oia.gen_bernoulli_device((n, n, n), 20251031, 0.5)
sweep(b, (n, n, n), "512^3 Bernoulli 0.5", "cg1_nt", [1, 0])
sweep(b, (n, n, n), "512^3 Bernoulli 0.5", "zchunk", [0, 16, 24, 32, 64])
b.free()
n = 768
b = oia.gen_blobs_device((n, n, n), 20251031, 12, 0.2)
sweep(b, (n, n, n), "768^3 blobs 0.2", "cg1_nt", [1, 0])
sweep(b, (n, n, n), "768^3 blobs 0.2", "zchunk", [0, 24, 64, 128])
b.free()
PY
