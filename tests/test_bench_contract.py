"""bench.py's output contract and the smoke entry point, on a small volume (GPU box only)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--size", "128", "--steps", "6", "--warmup", "2", "--cpu-size", "64", "--cpu-seconds", "1"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline", "cpu_baseline"}


def one_line(cmd, env=None):
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines   # ONE JSON line on stdout
    return json.loads(lines[0])


def check(d, n_gpus):
    assert KEYS <= set(d)
    assert d["metric"] == "Mvoxel-iters/sec" and d["unit"] == "Mvoxel-iters/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == n_gpus and d["steps"] == 6 and d["warmup"] == 2 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["scaling"] == "strong" and "workload" in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(r) and r["bound"] == "hbm" and r["unit"] == "GB/s"
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0 < r["frac"] < 1


def test_bench_line_single_gpu():
    d = one_line([sys.executable, "bench.py"] + SMALL)
    check(d, 1)
    c = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(c) and c["kind"] == "port" and c["value"] > 0
    assert d["time_to_1e-6"]["converged"] and d["time_to_1e-6_multigrid_pcg"]["converged"]
    assert all(d["homogenization"]["converged"])
    yz = d["directions_y_z"]   # BASELINE config 3: "all three directions"
    assert set(yz) == {"Y", "Z"} and all(v["value"] > 0 and v["multigrid_pcg_to_1e-6"]["converged"] for v in yz.values())
    # the default solver path carries its own roofline record (round 5): the three fine kernels of an iteration + the coarse levels
    m = d["time_to_1e-6_multigrid_pcg"]
    assert m["launches"] in (m["iterations"], m["iterations"] + 1, m["iterations"] + 2) and m["create_total_s"] > 0
    rf = m["roofline"]
    for k in ("form1_first_kernel", "form2_residual_presmoother_restriction", "form3_post_smoother_incl_pairwise_x"):
        assert {"ms_per_launch", "algorithmic_bytes_per_launch", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf[k]) and 0 < rf[k]["frac"] < 1
    assert rf["coarse_levels"]["ms_per_iteration"] > 0 and rf["iteration"]["ms"] > 0
    assert d["transport"] is None   # (one GPU, no launcher: no transport to describe)


def test_bench_line_through_the_rccl_path():
    """The launch line of the multi-GPU runs with ONE rank: process group, unique-id broadcast, the library's two RCCL
    communicators, split launches and the scalar all-reduce all run (there is no neighbour to exchange planes with)."""
    d = one_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                  "127.0.0.1", "--master-port", "29541", "bench.py", "--gpus", "1", "--force-dist", "--no-cpu-baseline"] + SMALL[:6])
    check(d, 1)
    assert d["time_to_1e-6_multigrid_pcg"]["converged"]


def test_watchdog_prints_the_line_without_the_unfinished_extras():
    d = one_line([sys.executable, "bench.py", "--size", "256", "--steps", "6", "--warmup", "2", "--extras-budget-s", "0.05"])
    assert d["value"] > 0 and "note" in d and d["homogenization"] is None


def test_run_watchdog_reports_a_start_that_never_finishes():
    """The whole-run watchdog (start-up + warm-up + timed region; a collective that never returns at N > 1): ONE line with a
    null value and a note, exit code 3, no re-exec of a process that touched the GPU."""
    p = subprocess.run([sys.executable, "bench.py", "--size", "256", "--steps", "6", "--warmup", "2", "--run-budget-s", "0.02"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert KEYS <= set(d) and d["value"] is None and "watchdog" in d["note"]


def test_smoke_entry_point():
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "smoke ok" in p.stdout, p.stderr[-2000:]
