"""`python bench.py --gpus N` with no launcher around it starts its own N ranks (CPU checks; the reference's analogue is
`mpirun -np N Diffusion inputs`, GNUmakefile:132-133).  No GPU is needed: --dry-launch ranks rendezvous over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    return p, lines


def test_self_launch_starts_n_fresh_ranks_and_relays_one_line():
    p, lines = run(["--gpus", "2", "--dry-launch", "--steps", "7", "--warmup", "3"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, lines      # ONE JSON line on the launcher's stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 3 and d["dry_launch"] is True
    ranks = d["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and [r["local_rank"] for r in ranks] == [0, 1]
    assert all(r["world_size"] == 2 and r["self_launched"] and r["launcher_was_gpu_clean"] for r in ranks)
    # the ranks are separate fresh processes, children of ONE launcher, started with the launcher's own command line
    assert len({r["pid"] for r in ranks}) == 2 and len({r["ppid"] for r in ranks}) == 1
    assert all(r["argv"] == ["--gpus", "2", "--dry-launch", "--steps", "7", "--warmup", "3"] for r in ranks)
    assert len({r["master"] for r in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    assert all(r["ipc_mode_legacy"] == "0" for r in ranks)   # (dmabuf IPC: RCCL across processes needs it on this pool)
    assert d["launcher"]["ranks"] == 2 and d["launcher"]["exit_codes"] == [0, 0]


def test_self_launch_honours_a_given_master_port_and_three_ranks():
    p, lines = run(["--gpus", "3", "--dry-launch"], {"MASTER_PORT": "29617"})
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads(lines[-1])
    assert len(d["ranks"]) == 3 and all(r["master"] == "127.0.0.1:29617" for r in d["ranks"])


def test_under_a_launcher_nothing_is_started_and_a_size_mismatch_is_refused():
    p, lines = run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and not lines and "disagree" in p.stderr


def test_a_rank_that_fails_gives_one_null_line_and_its_exit_code():
    """Without a GPU the real ranks cannot start (no HIP device): the launcher still prints ONE line (value null, the ranks' exit
    codes) and leaves with a non-zero code -- the driver never sees silence or a hang."""
    import openimpala_amd as oia
    if oia.lib().oi_device_count() > 0:
        import pytest
        pytest.skip("a GPU is present: the ranks would run the benchmark")
    p, lines = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--launch-grace-s", "5", "--run-budget-s", "60"])
    assert p.returncode != 0
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and any(c != 0 for c in d["launcher"]["exit_codes"])
