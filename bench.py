#!/usr/bin/env python3
"""bench.py -- Mvoxel-iters/s of the flow-through tortuosity CG on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
N > 1 is launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
(one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Workload (BASELINE.json metric: "1024^3 two-phase, 1/2/4/8 GPUs"): a seeded 1024^3
overlapping-sphere pack (config 3: radius 24 voxels, porosity ~0.4) generated in HBM,
phase 1 (pore) conducting, X direction, fp64 CG.  A "step" is ONE CG iteration over the
whole volume; the volume is z-slab decomposed over the N ranks (strong scaling: the total
work is fixed).  Inputs, mask and matrix codes are resident in HBM before the timed
region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is achievable
# compulsory HBM bytes per voxel of each kernel; cg1: 57 B on odd and
# 65 + 8 B on even iterations (x is touched every second iteration) = 61 B on average
BYTES_PER_VOXEL = {"cg1": 61.0, "fused_a": 49.0, "update_r": 24.0, "apply": 17.0, "update_p": 24.0, "update_xr": 48.0}


class _StdoutToStderr:
    """RCCL prints a version banner on the C stdout when its first communicator comes up; the
    driver wants exactly one JSON line there, so file descriptor 1 points at stderr meanwhile."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)  # C stdio buffers of every library in the process
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=1024, help="voxels per edge")
    ap.add_argument("--nz", type=int, default=0, help="planes along z if not cubic (slab-size rehearsal)")
    ap.add_argument("--structure", default="spheres", choices=["spheres", "bernoulli", "blobs"])
    ap.add_argument("--direction", default="X", choices=["X", "Y", "Z"], help="flow axis (headline: X)")
    ap.add_argument("--kernel", default="fused1", choices=["fused1", "fused", "unfused", "lean"],
                    help="lean = capacity mode (x, r, p only; 25 B/voxel of HBM)")
    ap.add_argument("--precond", default="none", choices=["none", "jacobi"])
    ap.add_argument("--no-solve", action="store_true", help="skip the wall-time-to-1e-6 solves")
    ap.add_argument("--no-cg-solve", action="store_true", help="skip the plain-CG solve to 1e-6 only (profiling runs of the multigrid path)")
    ap.add_argument("--solve-budget-s", type=float, default=150.0)
    ap.add_argument("--run-budget-s", type=float, default=600.0,
                    help="wall-clock cap on start-up + warm-up + timed region (a collective that never returns at N > 1): past it "
                         "rank 0 prints a line with value null and a note, and every rank leaves with exit code 3")
    ap.add_argument("--extras-budget-s", type=float, default=480.0,
                    help="wall-clock cap on everything after the timed region (solves to 1e-6, variants, homogenization, CPU "
                         "baseline): past it the JSON line is printed without the unfinished parts and the process exits")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-size", type=int, default=384, help="edge of the small second CPU sample")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="CPU seconds spent on the small second sample")
    ap.add_argument("--cpu-iters", type=int, default=10, help="timed CPU iterations on the bench configuration itself")
    ap.add_argument("--no-apply-bench", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the fp32-stored-p variant")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the torch.distributed + RCCL code path even with one rank (rehearsal)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="named process default of the library (oi_tort_set_option) set before any handle is built: measurement runs")
    ap.add_argument("--dry-launch", action="store_true",
                    help="N > 1 without a launcher: start the N child ranks, let them rendezvous over gloo on the CPU and report "
                         "their environment and command line instead of benchmarking (tests/test_bench_launch_cpu.py)")
    ap.add_argument("--launch-grace-s", type=float, default=30.0,
                    help="self-launched ranks: once one rank has left with an error the others get this long before they are stopped")
    return ap.parse_args()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (the reference's analogue: `mpirun -np N Diffusion inputs`,
    GNUmakefile:132-133).  This process has NOT imported torch or the solver library and never touches HIP: it starts N fresh
    interpreters of this file, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torchrun would
    set them, relays rank 0's single JSON line, sends everything else the ranks print to stderr, and leaves with the worst exit
    code.  A rank that dies takes the others with it after --launch-grace-s (they would wait in a collective until their own
    watchdog); the whole launch is bounded by the ranks' own budgets plus a margin."""
    import signal
    import subprocess
    n = args.gpus
    port = int(os.environ.get("MASTER_PORT") or free_port())
    argv = [a for a in sys.argv[1:]]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(port), OI_BENCH_SELF_LAUNCHED="1",
                   # (what the ranks may report: this process has loaded neither torch nor the solver library, so no HIP call was made)
                   OI_BENCH_LAUNCHER_CLEAN="1" if not ({"torch", "openimpala_amd"} & set(sys.modules)) else "0")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode(errors="replace").splitlines()), daemon=True)
    reader.start()
    deadline = time.time() + args.run_budget_s + args.extras_budget_s + args.solve_budget_s + 120.0
    first_bad = None
    note = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        now = time.time()
        if first_bad is None and any(p.poll() not in (None, 0) for p in procs):
            first_bad = now
        if (first_bad is not None and now - first_bad > args.launch_grace_s) or now > deadline:
            note = "launcher: a rank left with an error and the others were stopped" if first_bad is not None else \
                   "launcher: the ranks overran every budget and were stopped"
            for p in procs:
                if p.poll() is None:
                    p.send_signal(signal.SIGTERM)
            t_kill = time.time() + 10.0
            while any(p.poll() is None for p in procs) and time.time() < t_kill:
                time.sleep(0.2)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
    for p in procs:
        p.wait()
    reader.join(timeout=10.0)
    codes = [p.returncode for p in procs]
    lines = [l for l in out0 if l.strip()]
    json_lines = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if l not in json_lines[-1:]:
            print(l, file=sys.stderr)
    rc = 0
    for c in codes:
        if c != 0:
            rc = max(rc, c if c > 0 else 128 - c)   # (a rank ended by signal s counts as 128 + s)
    if json_lines:
        line = json_lines[-1]
        try:
            d = json.loads(line)
            d["launcher"] = {"kind": "self (bench.py started the ranks)", "ranks": n, "exit_codes": codes, "master_port": port}
            if note:
                d["launcher"]["note"] = note
            line = json.dumps(d)
        except ValueError:
            pass
        print(line, flush=True)
    else:
        print(json.dumps({"metric": "Mvoxel-iters/sec", "value": None, "unit": "Mvoxel-iters/s", "n_gpus": n, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                          "config": {"workload": f"{args.size}^3 two-phase {args.structure}, z-slabs over {n} GPU(s)"},
                          "roofline": None, "cpu_baseline": None,
                          "launcher": {"kind": "self (bench.py started the ranks)", "ranks": n, "exit_codes": codes, "master_port": port},
                          "note": (note or "launcher: rank 0 printed no line") + f"; exit codes {codes}; nothing was measured"}), flush=True)
        rc = rc or 4
    sys.exit(rc)


def dry_rank(args, world, rank, local_rank):
    """--dry-launch child: rendezvous over gloo at the address the launcher gave, gather every rank's launch facts, rank 0 prints
    them as the one JSON line.  Touches neither HIP nor the solver library."""
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = {"rank": rank, "local_rank": local_rank, "world_size": world, "argv": sys.argv[1:], "pid": os.getpid(), "ppid": os.getppid(),
            "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
            "self_launched": os.environ.get("OI_BENCH_SELF_LAUNCHED") == "1",
            "launcher_was_gpu_clean": os.environ.get("OI_BENCH_LAUNCHER_CLEAN") == "1", "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "Mvoxel-iters/sec", "value": None, "unit": "Mvoxel-iters/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_launch": True, "ranks": allr}), flush=True)
    dist.destroy_process_group()


def make_structure(oia, args, shape, z0, nzl):
    if args.structure == "bernoulli":
        return oia.gen_bernoulli_device(shape, 20251031, 0.5, z0, nzl), "Bernoulli(0.5), seed 20251031"
    if args.structure == "blobs":  # config 5 geometry: correlated, porosity 0.2
        spacing = 24 if shape[2] >= 512 else max(6, shape[2] // 20)
        return (oia.gen_blobs_device(shape, 20251031, spacing, 0.2, z0, nzl),
                f"interpolated lattice-noise blobs, spacing {spacing}, target porosity 0.2, seed 20251031")
    radius = 24 if shape[2] >= 512 else max(4, shape[2] // 24)
    return (oia.gen_spheres_device(shape, 20251031, radius, 0.4, z0, nzl),
            f"overlapping spheres r={radius}, target porosity 0.4, seed 20251031")


def kernels_sha():
    """hash of the kernel sources a PMC companion run must match (profiles/pmc_summary.json): every device source of the product library"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "openimpala_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(csrc, "oi_k_*.hip")) + [os.path.join(csrc, n) for n in ("oi_kern.h", "oi_device.h", "oi_mg.hip", "oi_effdiff.hip")]):
        if os.path.basename(f) == "oi_k_xcheck.hip":
            continue
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read())
    return h.hexdigest()


def pmc_traffic(key):
    """HBM bytes per launch from the committed PMC companion run (tools/rocprof.sh) -- only if it was taken with THIS
    kernel source; a stale file gives None, never an old number."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        ent = pmc.get(key)
        if ent and ent.get("kernels_sha256") == kernels_sha():
            return ent["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_cg_rate(o, code, x0, iters):
    t0 = time.time()
    o.cg_fixed(code, x0, iters)
    return time.time() - t0


def cpu_baseline(oia, args, full=None):
    """The oracle's matrix-free CG (same algorithm, OpenMP), timed on this box's host cores: `--cpu-iters` iterations of
    the bench configuration itself (`full` = (code, mask, description) downloaded from the GPU handle -- the oracle's own
    mask pass would take minutes at 1024^3 and is not what is timed), plus a small second sample with a one-core figure.
    A reported baseline, not the target."""
    import numpy as np
    from oracle import oracle as o

    threads = o.lib().oi_oracle_num_threads()
    out = {"unit": "Mvoxel-iters/s", "cores": threads, "kind": "port"}
    label = "CPU restatement (matrix-free CG, OpenMP), not Hypre FlexGMRES+SMG"
    if full is not None:
        code, mask, desc = full
        nvox = code.size
        x0 = o.initial_field(mask, 0, 0.0, 1.0)
        del mask
        cpu_cg_rate(o, code, x0, 2)  # page faults, thread start-up
        dt = cpu_cg_rate(o, code, x0, args.cpu_iters)
        out["value"] = nvox * args.cpu_iters / dt / 1e6
        out["sample"] = f"the bench configuration itself: {desc}, X direction, {args.cpu_iters} CG iterations in {dt:.1f} s; {label}"
        del x0, code
    # small second sample (and the one-core rate of the same restatement)
    n = args.cpu_size
    shape = (n, n, n)
    buf, desc = make_structure(oia, args, shape, 0, n)
    phase = buf.download(np.uint8, shape)
    buf.free()
    mask, nact = o.active_mask(phase, 1, 0)
    if nact == 0:
        return out if "value" in out else None
    code = o.codes(mask, 0)
    x0 = o.initial_field(mask, 0, 0.0, 1.0)
    cpu_cg_rate(o, code, x0, 5)
    per_iter = cpu_cg_rate(o, code, x0, 10) / 10
    iters = int(max(10, min(20000, args.cpu_seconds / max(per_iter, 1e-6))))
    dt = cpu_cg_rate(o, code, x0, iters)
    small = {"value": n ** 3 * iters / dt / 1e6, "sample": f"{n}^3 {desc}, X direction, {iters} CG iterations in {dt:.1f} s"}
    try:  # the reference's README advises OMP_NUM_THREADS=1 under MPI: the same kernel on ONE core
        o.lib().oi_oracle_set_threads(1)
        n1 = max(2, min(20, int(2.0 / max(per_iter * threads, 1e-6))))
        small["one_core_value"] = n ** 3 * n1 / cpu_cg_rate(o, code, x0, n1) / 1e6
    finally:
        o.lib().oi_oracle_set_threads(threads)
    out["small_sample"] = small
    if "value" not in out:
        out["value"] = small["value"]
        out["sample"] = small["sample"] + "; " + label
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)   # (does not return; nothing GPU-related has been imported or called in this process)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree",
                  file=sys.stderr)
        sys.exit(2)
    if args.dry_launch:
        dry_rank(args, world, rank, local_rank)
        return

    # Whole-run watchdog: start-up, warm-up and the timed region of an N > 1 run contain collectives that this code has
    # never executed on real multi-GPU hardware.  If one never returns, rank 0 still prints ONE line (value null + note)
    # and every rank leaves through os._exit -- never by re-executing a process that has touched the GPU.
    phase_box = {"phase": "start-up"}

    def run_overrun():
        if rank == 0:
            print(json.dumps({"metric": "Mvoxel-iters/sec", "value": None, "unit": "Mvoxel-iters/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                              "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": f"{args.size}^3 two-phase {args.structure}, z-slabs over {world} GPU(s)"},
                              "roofline": None, "cpu_baseline": None,
                              "note": f"watchdog: no progress past '{phase_box['phase']}' within {args.run_budget_s:.0f} s "
                                      "(a collective or launch did not return); nothing was measured"}), flush=True)
        sys.stdout.flush()
        os._exit(3)

    run_watchdog = threading.Timer(args.run_budget_s, run_overrun)
    run_watchdog.daemon = True
    run_watchdog.start()

    use_dist = world > 1 or args.force_dist
    dist = None
    torch = None
    comm = None
    if use_dist:
        # torch FIRST: it brings its own libamdhip64 / librccl; loaded first, the solver library
        # binds to the same HIP runtime and RCCL instead of pulling a second copy into the process
        import torch
        import torch.distributed as dist
        if args.force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")

    import openimpala_amd as oia
    if args.force_dist:
        oia.set_option("force_comm", 1)
    for kv in args.option:
        key, _, val = kv.partition("=")
        oia.set_option(key.strip(), float(val))

    if use_dist:
        with _StdoutToStderr():
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            uid = torch.zeros(oia.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(oia.Comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(uid, 0)
            comm = oia.Comm.create(bytes(uid.cpu().numpy().tobytes()), world, rank, local_rank)
            dist.barrier()
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    phase_box["phase"] = "structure + handle creation (mask, codes, vectors)"
    n = args.size
    nz = args.nz if args.nz > 0 else n
    nvox = n * n * nz
    shape = (nz, n, n)
    z0, z1 = oia.slab_range(nz, world, rank)
    nzl = z1 - z0
    t_gen = time.time()
    buf, desc = make_structure(oia, args, shape, z0, nzl)
    t_gen = time.time() - t_gen
    direction = {"X": oia.Direction.X, "Y": oia.Direction.Y, "Z": oia.Direction.Z}[args.direction]
    if args.kernel == "lean":  # q-free passes: p.Ap reads p + code; the update reads p, code, r, x and writes r, x
        BYTES_PER_VOXEL.update({"apply": 9.0, "update_xr": 41.0, "update_p": 24.0})
    kernel = {"fused1": oia.KERNEL_FUSED1, "fused": oia.KERNEL_FUSED, "unfused": oia.KERNEL_UNFUSED,
              "lean": oia.KERNEL_LEAN}[args.kernel]
    precond = oia.PRECOND_JACOBI if args.precond == "jacobi" else oia.PRECOND_NONE
    t_create = time.time()
    # eps = 1e-300 during the timed region: every one of the K steps does full work
    kw = dict(eps=1e-300, kernel=kernel, precond=precond, profile=1, maxiter=200000, device=local_rank)
    if comm is not None:
        t = oia.TortuosityMI355(buf, 1, direction, 0.0, 1.0, shape=(nzl, n, n), comm=comm, z0=z0, nz_global=nz,
                                **kw)
    else:
        t = oia.TortuosityMI355(buf, 1, direction, 0.0, 1.0, shape=shape, **kw)
    t_create = time.time() - t_create
    buf.free()
    placement = t.placement()
    per_rank = None
    if dist is not None:  # N > 1: what every rank's handle cost and where its vectors went (rank order)
        mine = {"rank": rank, "create_total_s": t_create, "planes": nzl,
                "vector_placement": {k: placement[k] for k in ("nslots", "slots", "tried", "ms_in_order", "ms_kept", "alloc_s", "search_s", "cache_hit")}}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    st = t.stats()
    if st.active_vf <= 0:
        raise RuntimeError("synthetic structure does not percolate")

    # ---- stand-alone stencil apply (the kernel the 70 %-of-roofline target names) ----
    apply_info = None
    phase_box["phase"] = "stand-alone apply"
    if not args.no_apply_bench and args.kernel != "lean":
        ms = t.bench_apply(20, oia.KERNEL_FUSED)
        vox_local = n * n * nzl
        gbs = BYTES_PER_VOXEL["apply"] * vox_local / ms / 1e6
        apply_info = {"kernel": "k_apply7 (q = A v for ANY v, 17 B/voxel, every voxel streamed; wave-autonomous, DPP x-neighbours, "
                                "z-sweep launch order)", "ms_per_launch": ms,
                      "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                      "gvoxel_per_s": vox_local / ms / 1e6, "traffic": pmc_traffic(f"apply:{n}:{world}")}
        # what a plain 1 read : 1 write copy stream of the same size reaches on this box right now (16 B/voxel)
        try:
            cms = oia.stream_copy_bench(8 * vox_local, 10)
            apply_info["copy_stream_same_box"] = {"ms": cms, "GBps": 16.0 * vox_local / cms / 1e6,
                                                  "frac": 16.0 * vox_local / cms / 1e6 / HBM_PEAK_GBS,
                                                  "what": "linear float4-class copy (8 x 16 B nontemporal loads in flight per lane), "
                                                          "8 bytes read + 8 written per voxel of the same volume"}
        except oia.OpenImpalaError as e:
            apply_info["copy_stream_same_box"] = {"error": str(e)}
        # the LDS-tiled form (round 1's kernel) for comparison
        oia.set_option("apply_kernel", 1); oia.set_option("tile_map", 0); t.retune()
        ms_lds = t.bench_apply(10, oia.KERNEL_FUSED)
        oia.set_option("apply_kernel", 0); oia.set_option("tile_map", 2); t.retune()
        apply_info["lds_tiled_form"] = {"kernel": "k_fused_a<PURE>, 64-plane chunks, XCD-contiguous order", "ms_per_launch": ms_lds,
                                        "frac": BYTES_PER_VOXEL["apply"] * vox_local / ms_lds / 1e6 / HBM_PEAK_GBS}
        # the same stencil as the CG applies it: v vanishes off the free set, lines without a free voxel are skipped
        ms_cg = t.bench_apply(20, oia.KERNEL_LEAN)
        live = t.live_fraction(16)
        moved = (1.0 + 16.0 * live) * vox_local
        apply_info["cg_vector"] = {"ms_per_launch": ms_cg, "live_line_fraction": live, "gvoxel_per_s": vox_local / ms_cg / 1e6,
                                   "achieved": moved / ms_cg / 1e6, "frac": moved / ms_cg / 1e6 / HBM_PEAK_GBS,
                                   "dense_equivalent_GBps": BYTES_PER_VOXEL["apply"] * vox_local / ms_cg / 1e6}

    # ---- timed region: K CG iterations ----
    phase_box["phase"] = "warm-up iterations"
    t.cg_begin()
    t.cg_iterate(args.warmup)
    t.sync()
    t.reset_kernel_times()
    barrier()
    phase_box["phase"] = "timed region"
    t0 = time.perf_counter()
    t.cg_iterate(args.steps)
    t.sync()
    barrier()
    dt = time.perf_counter() - t0
    phase_box["phase"] = "after the timed region"
    run_watchdog.cancel()
    ktimes = t.kernel_times()
    it_done, relres, conv = t.cg_state()
    transport = None
    if dist is not None:
        # what every rank measured and what its transport really is: the N-rank line describes itself
        info = comm.info()
        mine_t = {"rank": rank, "ms_per_step": dt / max(1, args.steps) * 1e3, "cg1_ms_per_launch": (ktimes["cg1"][0] / max(1, ktimes["cg1"][1])) if "cg1" in ktimes else None,
                  "rccl_ranks_collective": info["ranks_collective"], "rccl_ranks_p2p": info["ranks_p2p"], "exchange_overlaps": info["exchange_overlaps"],
                  "transport": info["kind"], "device": local_rank}
        timed = [None] * world
        dist.all_gather_object(timed, mine_t)
        for a, b in zip(per_rank, timed):
            a.update(b)
        transport = {"kind": info["kind"], "rccl_ranks": [min(t["rccl_ranks_collective"] for t in timed), min(t["rccl_ranks_p2p"] for t in timed)],
                     "exchange_overlaps": all(t["exchange_overlaps"] for t in timed),
                     "exchange_stream": ("own stream" + (f", priority {os.environ['OI_COMM_PRIO']}" if os.environ.get("OI_COMM_PRIO") in ("greatest", "least") else ""))
                                        if all(t["exchange_overlaps"] for t in timed) else "solver stream",
                     "ms_per_step_per_rank": [t["ms_per_step"] for t in timed]}
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # eps = 1e-300 keeps every step busy; should CG nevertheless finish inside the timed region (tiny volumes, huge
    # --steps), the later launches return at once and only the iterations really done count
    steps_done = min(args.steps, max(0, it_done - args.warmup)) if conv else args.steps
    value = nvox * steps_done / dt / 1e6
    alg_bytes_iter = sum(BYTES_PER_VOXEL[k] for k in ktimes if k in BYTES_PER_VOXEL)

    # dominant kernel of the timed region, from the HIP events on the solver's stream
    main_k = max((k for k in ktimes if k in BYTES_PER_VOXEL), key=lambda k: ktimes[k][0])
    ms_launch = ktimes[main_k][0] / max(1, ktimes[main_k][1])
    traffic = pmc_traffic(f"{main_k}:{n}:{world}")  # None unless a PMC companion run of THIS kernel source is committed
    # The CG-form kernels never move a line of r/q/p/x that holds no free voxel, and move
    # live lines whole.  ALGORITHMIC bytes of a launch = the code byte of every voxel + the kernel's vector
    # bytes (SURVEY 8(d) accounting, 60 of the 61 B) for every voxel of a LIVE line; `dense_equivalent` prices the
    # same launch as if every voxel streamed its 61 B (what earlier bench lines of this round reported, and
    # what the metric Mvoxel-iters/s counts) -- it may exceed the HBM peak and is not a bandwidth.
    # (the compacted vectors of the single-kernel iteration store lines of oia.line_voxels() = 8 voxels, one 64-byte
    # HBM burst; the dense vectors of the other variants are predicated on whole 128-byte cache lines)
    if steps_done < args.steps:
        extras_note_early = f"CG converged after {steps_done} of the {args.steps} timed steps; value counts the steps done"
    else:
        extras_note_early = None
    line_vox = oia.line_voxels() if main_k == "cg1" else 16
    nfree_frac = t.live_fraction(1)
    live16 = t.live_fraction(line_vox)
    dense_bytes = BYTES_PER_VOXEL[main_k] * n * n * nzl
    alg_bytes = (1.0 + (BYTES_PER_VOXEL[main_k] - 1.0) * live16) * n * n * nzl
    achieved = alg_bytes / ms_launch / 1e6
    roofline = {"bound": "hbm", "kernel": main_k, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes, "live_line_fraction": live16, "line_voxels": line_vox,
                "free_voxel_minimum": {"what": "code byte of every voxel + the vector bytes of FREE voxels only (no layout can move less)",
                                       "bytes_per_launch": (1.0 + (BYTES_PER_VOXEL[main_k] - 1.0) * nfree_frac) * n * n * nzl,
                                       "GBps": (1.0 + (BYTES_PER_VOXEL[main_k] - 1.0) * nfree_frac) * n * n * nzl / ms_launch / 1e6},
                "dense_equivalent": {"bytes_per_voxel_iter": alg_bytes_iter, "bytes_per_launch": dense_bytes,
                                     "GBps": dense_bytes / ms_launch / 1e6,
                                     "frac_of_peak": dense_bytes / ms_launch / 1e6 / HBM_PEAK_GBS,
                                     "survey_89B_model_frac": value * 1e6 * 89.0 / world / (HBM_PEAK_GBS * 1e9)},
                "ms_per_launch": ms_launch,
                "bytes_per_voxel_iter_in_live_lines": alg_bytes_iter}
    t.cg_finish()
    if main_k == "cg1" and args.kernel == "fused1":
        # in-situ ceiling of the CG kernel's traffic mix: a kernel that only moves the six double-buffered vectors where
        # they lie (3 read, 3 written, 16-byte nontemporal accesses), HIP events on the handle's stream
        try:
            sms, sbytes = t.bench_stream(10)
            roofline["stream_3r3w_same_vectors"] = {
                "ms": sms, "GBps": sbytes / sms / 1e6, "frac_of_peak": sbytes / sms / 1e6 / HBM_PEAK_GBS,
                "cg_kernel_vs_this_stream": achieved / (sbytes / sms / 1e6),
                # which class of allocations this process got (profiles/r02/placement/final_12_runs.txt: eleven of twelve
                # processes stream their vectors at 6.09-6.41 TB/s and run the kernel at 6.39-6.58 ms; one at 5.80 / 7.24)
                "placement_class": "fast" if sbytes / sms / 1e6 >= 6000.0 else "slow",
                "what": "d_j = s_j over r2,q2,p[1] -> r,q,p[0] of this handle (whole vectors, same placement); the CG kernel also "
                        "streams x, the code bytes and the line table"}
        except oia.OpenImpalaError as e:
            roofline["stream_3r3w_same_vectors"] = {"error": str(e)}
    cpu_full = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_iters > 0 and args.kernel != "lean" and nvox <= 1100 ** 3:
        cpu_full = (t.codes(), t.mask(), f"{n}x{n}x{nz} {desc}")  # (bit-identical to the oracle's own: tests/test_gpu_parity.py)

    # Everything below is reported beside the metric, never part of it.  A watchdog prints the line without the
    # unfinished parts if they overrun (a collective that never returns on some node must not cost the measurement).
    extras = {"solve": None, "solve_mg": None, "variants": None, "other_dirs": None, "homog": None, "cpu": None, "note": extras_note_early}
    emitted = threading.Lock()

    def emit():
        if not emitted.acquire(blocking=False):
            return
        if rank != 0:
            return
        out = {
            "metric": "Mvoxel-iters/sec", "value": value, "unit": "Mvoxel-iters/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(1, steps_done) * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{n}x{n}x{nz} two-phase {desc}; phase 1, {args.direction} direction; matrix-free CG "
                                   f"({args.kernel}, precond {args.precond}); z-slabs over {world} GPU(s)",
                       "voxels": nvox, "active_vf": st.active_vf, "volume_fraction": st.vf,
                       "unknowns": st.n_free, "parallelism": f"z-slab x{world}"},
            "roofline": roofline, "cpu_baseline": extras["cpu"],
            "apply_roofline": apply_info,
            "kernels_ms_per_launch": {k: v[0] / max(1, v[1]) for k, v in ktimes.items()},
            "steps_effective": steps_done, "relres_after_timed_region": relres,
            "time_to_1e-6": extras["solve"], "time_to_1e-6_multigrid_pcg": extras["solve_mg"],
            "variants": extras["variants"], "directions_y_z": extras.get("other_dirs"), "homogenization": extras["homog"],
            "setup_s": {"generate": t_gen, "create_total": t_create, "mask": st.t_mask_s,
                        "flood_sweeps": st.flood_sweeps, "upload_pack": st.t_upload_s, "codes_vectors_placement": st.t_setup_s,
                        "placement_alloc_s": placement["alloc_s"], "placement_s": placement["search_s"]},
            "device_bytes": st.bytes_device,
            # where the CG vectors were put (slots of one arena) and what the placement search measured, ms per iteration
            "vector_placement": placement,
            "per_rank": per_rank,
            "transport": transport,
        }
        if extras["note"]:
            out["note"] = extras["note"]
        print(json.dumps(out), flush=True)

    def overrun():
        extras["note"] = f"extras stopped by the watchdog after {args.extras_budget_s:.0f} s; null entries did not finish"
        emit()
        sys.stdout.flush()
        os._exit(0)

    watchdog = threading.Timer(args.extras_budget_s, overrun)
    watchdog.daemon = True
    watchdog.start()

    # ---- A/B on THIS handle (same vectors, same placement, one CG run): the default cache policy against the
    # nontemporal accesses the kernel ships with (option cg1_nt, DESIGN.md section 4 "Round 4: cache policy") ----
    if args.kernel == "fused1" and not args.no_variants and main_k == "cg1":
        try:
            ab = {}
            t.cg_begin()
            t.cg_iterate(args.warmup)
            for pol in (0, 1):
                oia.set_option("cg1_nt", pol); t.retune()
                t.cg_iterate(4); t.sync(); t.reset_kernel_times()
                t.cg_iterate(args.steps); t.sync()
                kt = t.kernel_times()
                ab[pol] = kt["cg1"][0] / max(1, kt["cg1"][1])
            roofline["cache_policy_same_handle"] = {
                "what": "ms per launch of the CG kernel on this handle, alternating: every access with the default cache policy / "
                        "r, p, x written and x, p(k-2) read nontemporally (shipped)",
                "default_policy_ms": ab[0], "nontemporal_ms": ab[1], "gain": 1.0 - ab[1] / ab[0]}
            t.cg_finish()
        finally:
            oia.set_option("cg1_nt", 1); t.retune()

    # ---- wall time to relative residual 1e-6 (second half of the BASELINE metric) ----
    solve = None
    if not args.no_solve and not args.no_cg_solve:
        t.reset_kernel_times()
        barrier()
        t.set_eps(1e-6)
        s0 = time.perf_counter()
        t.cg_begin()
        done, its, rel = False, 0, float("nan")
        stop = False
        while not stop:
            t.cg_iterate(250)
            its, rel, done = t.cg_state()  # identical on every rank (all-reduced r.r)
            stop = done or its >= 199000 or time.perf_counter() - s0 >= args.solve_budget_s
            if dist is not None:  # the wall-clock test differs between ranks: agree on it
                flag = torch.tensor([1 if stop else 0], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                stop = bool(flag.item())
        t.cg_finish()
        barrier()
        s1 = time.perf_counter() - s0
        solve = {"eps": 1e-6, "converged": bool(done), "iterations": its, "relres": rel, "seconds": s1,
                 "mvoxel_iters_per_s": nvox * its / s1 / 1e6 if s1 > 0 else None}
        extras["solve"] = solve

    # ---- same solve with the multigrid cycle as preconditioner (the role of SMG in the reference) ----
    solve_mg = None
    if not args.no_solve and args.kernel != "lean":  # (the capacity mode is for volumes that leave no room for the cycle)
        t.destroy()
        bufm, _ = make_structure(oia, args, shape, z0, nzl)
        kwm = dict(eps=1e-6, precond=oia.PRECOND_MG, profile=1, maxiter=1000, device=local_rank)
        t_mg_create = time.perf_counter()
        if comm is not None:
            tm = oia.TortuosityMI355(bufm, 1, direction, 0.0, 1.0, shape=(nzl, n, n), comm=comm, z0=z0, nz_global=nz, **kwm)
        else:
            tm = oia.TortuosityMI355(bufm, 1, direction, 0.0, 1.0, shape=shape, **kwm)
        t_mg_create = time.perf_counter() - t_mg_create
        bufm.free()
        barrier()
        m0 = time.perf_counter()
        tm.cg_begin()
        mlaunches = tm.cg_run()   # the library's own convergence loop (what value() runs); the stop test is the device's flag
        mits, mrel, mdone = tm.cg_state()
        tm.cg_finish()
        barrier()
        m1 = time.perf_counter() - m0
        mk = tm.kernel_times()
        mg_create_s = t_mg_create
        solve_mg = {"eps": 1e-6, "preconditioner": "2x2x2 aggregation multigrid cycle per PCG iteration", "converged": bool(mdone),
                    "iterations": mits, "launches": mlaunches, "relres": mrel, "seconds": m1, "setup_seconds": tm.stats().t_setup_s,
                    # the DEFAULT handle of the ABI (precond = mg): no placement search on this path
                    "create_total_s": mg_create_s, "mask_s": tm.stats().t_mask_s,
                    "device_GB": tm.stats().bytes_device / 1e9,
                    "kernels_ms_per_launch": {k: v[0] / max(1, v[1]) for k, v in mk.items()
                                              if k in ("mg_vcycle", "mg_post", "fused_a", "update_r", "apply", "update_p", "update_xr")}}
        # ---- where the default path stands against the roofline: the three fine kernels of an iteration are forms of the descriptor-
        # addressed kernel (k_cg1b FORM 1 / 2 / 3) on compacted vectors; algorithmic bytes = the code byte of every voxel + the form's
        # vector bytes for every voxel of a STORED line (FORM 1 z 8 + p 8 read, p 8 + q 8 written; FORM 2 r 8 +
        # q 8 read, r 8 written, 0.5 coarse right-hand side; FORM 3 b 8 + 0.5 coarse correction read, z 8 written, and on every
        # second launch x 8 + two p 8 + 8 read, x 8 written = 32, i.e. 16 per launch on average); HIP events on the solver's stream;
        # traffic from the committed PMC passes of the same kernel sources (tools/rocprof_mg.sh), null otherwise
        if all(k in mk for k in ("fused_a", "update_r", "mg_vcycle", "mg_post")):   # ("mg_post" is timed only in the rotated iteration on compacted vectors)
            livem = tm.live_fraction(oia.line_voxels())
            voxl = n * n * nzl
            per = {"form1_first_kernel": ("fused_a", 32.0, "mg_form1"), "form2_residual_presmoother_restriction": ("update_r", 24.5, "mg_form2"),
                   "form3_post_smoother_incl_pairwise_x": ("mg_post", 32.5, "mg_form3")}
            roof = {}
            for name, (kk, bv, pkey) in per.items():
                ms_k = mk[kk][0] / max(1, mk[kk][1])
                ab = (1.0 + bv * livem) * voxl
                tr = pmc_traffic(f"{pkey}:{n}:{world}")
                if pkey == "mg_form3":   # (two instantiations alternate: with and without the x update)
                    tx = pmc_traffic(f"mg_form3x:{n}:{world}")
                    tr = 0.5 * (tr + tx) if tr is not None and tx is not None else None
                roof[name] = {"ms_per_launch": ms_k, "algorithmic_bytes_per_launch": ab, "achieved": ab / ms_k / 1e6, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": ab / ms_k / 1e6 / HBM_PEAK_GBS, "traffic": tr}
            ms_c = (mk["mg_vcycle"][0] - mk["mg_post"][0]) / max(1, mk["mg_vcycle"][1])
            roof["coarse_levels"] = {"ms_per_iteration": ms_c, "traffic": pmc_traffic(f"mg_coarse:{n}:{world}"),
                                     "what": "every level below the fine one (first coarse level: an eighth of the cells, fp32): ~100 launches, latency- and L2-bound; no byte model"}
            it_ms = sum(mk[k][0] / max(1, mk[k][1]) for k in ("fused_a", "update_r", "mg_vcycle"))
            roof["iteration"] = {"ms": it_ms, "live_line_fraction": livem, "bound": "hbm",
                                 "fine_kernels_algorithmic_bytes": sum(v["algorithmic_bytes_per_launch"] for k_, v in roof.items() if k_.startswith("form")),
                                 "fine_kernels_frac_of_peak": sum(v["algorithmic_bytes_per_launch"] for k_, v in roof.items() if k_.startswith("form"))
                                 / (sum(v["ms_per_launch"] for k_, v in roof.items() if k_.startswith("form")) * 1e6) / HBM_PEAK_GBS}
            solve_mg["roofline"] = roof
        tm.destroy()
        t = None
        extras["solve_mg"] = solve_mg
        # the same solve with the search direction STORED as fp32 (opts.p_storage; x, r, q, z and every operation fp64): a variant,
        # not the default -- the reference keeps every vector in fp64
        if not args.no_variants:
            bufm, _ = make_structure(oia, args, shape, z0, nzl)
            kwm["p_storage"] = oia.STORE_F32
            if comm is not None:
                tm = oia.TortuosityMI355(bufm, 1, direction, 0.0, 1.0, shape=(nzl, n, n), comm=comm, z0=z0, nz_global=nz, **kwm)
            else:
                tm = oia.TortuosityMI355(bufm, 1, direction, 0.0, 1.0, shape=shape, **kwm)
            bufm.free()
            barrier()
            m0 = time.perf_counter()
            tm.cg_begin()
            tm.cg_run()
            fits, frel, fdone = tm.cg_state()
            tm.cg_finish()
            barrier()
            solve_mg["p_stored_fp32"] = {"converged": bool(fdone), "iterations": fits, "relres": frel, "seconds": time.perf_counter() - m0,
                                         "device_GB": tm.stats().bytes_device / 1e9}
            tm.destroy()

    # ---- variant: search direction p STORED as fp32 (x, r, q and all arithmetic fp64) ----
    variants = None
    if not args.no_variants and args.kernel not in ("unfused", "lean"):
        if t is not None:
            t.destroy()
        buf2, _ = make_structure(oia, args, shape, z0, nzl)
        kw2 = dict(kw, p_storage=oia.STORE_F32)
        if comm is not None:
            t = oia.TortuosityMI355(buf2, 1, direction, 0.0, 1.0, shape=(nzl, n, n), comm=comm, z0=z0,
                                    nz_global=nz, **kw2)
        else:
            t = oia.TortuosityMI355(buf2, 1, direction, 0.0, 1.0, shape=shape, **kw2)
        buf2.free()
        t.cg_begin()
        t.cg_iterate(args.warmup)
        t.sync()
        t.reset_kernel_times()
        barrier()
        v0 = time.perf_counter()
        t.cg_iterate(args.steps)
        t.sync()
        barrier()
        vdt = time.perf_counter() - v0
        vk = t.kernel_times()
        if dist is not None:
            tt = torch.tensor([vdt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            vdt = float(tt.item())
        t.cg_finish()
        variants = {"p_stored_fp32": {
            "note": "search direction kept as fp32 in HBM; x, r, q and every operation fp64; r stays the exact "
                    "recurrence residual of x (tests/test_gpu_parity.py::test_fp32_stored_search_direction)",
            "value": nvox * args.steps / vdt / 1e6, "unit": "Mvoxel-iters/s", "ms_per_step": vdt / args.steps * 1e3,
            "bytes_per_voxel_iter": alg_bytes_iter - (10.0 if args.kernel == "fused1" else 8.0),
            "kernels_ms_per_launch": {k: v[0] / max(1, v[1]) for k, v in vk.items()}}}

    # ---- BASELINE config 3 names "all three directions": the same metric and the default solver's time to 1e-6 along Y and Z ----
    other_dirs = None
    if not args.no_variants and not args.no_solve and world == 1 and comm is None and args.kernel == "fused1" and args.direction == "X":
        if t is not None:
            t.destroy()
            t = None
        other_dirs = {}
        try:   # (an extra: whatever happens here must not cost the line its headline figures)
            bufd, _ = make_structure(oia, args, shape, z0, nzl)
            for dname, dval in (("Y", oia.Direction.Y), ("Z", oia.Direction.Z)):
                td = oia.TortuosityMI355(bufd, 1, dval, 0.0, 1.0, shape=shape, **kw)
                td.cg_begin(); td.cg_iterate(args.warmup); td.sync(); td.reset_kernel_times()
                d0 = time.perf_counter()
                td.cg_iterate(args.steps); td.sync()
                ddt = time.perf_counter() - d0
                dk = td.kernel_times()
                td.cg_finish()
                live_d = td.live_fraction(oia.line_voxels())
                td.destroy()
                ms_d = dk["cg1"][0] / max(1, dk["cg1"][1])
                ab_d = (1.0 + (BYTES_PER_VOXEL["cg1"] - 1.0) * live_d) * nvox
                tmd = oia.TortuosityMI355(bufd, 1, dval, 0.0, 1.0, shape=shape, eps=1e-6, precond=oia.PRECOND_MG, maxiter=1000, device=local_rank)
                m0 = time.perf_counter()
                tmd.cg_begin(); ld = tmd.cg_run(1000); tmd.cg_finish()
                mdt = time.perf_counter() - m0
                mits, mrel, mdone = tmd.cg_state()
                tmd.destroy()
                other_dirs[dname] = {"value": nvox * args.steps / ddt / 1e6, "unit": "Mvoxel-iters/s", "ms_per_step": ddt / args.steps * 1e3,
                                     "cg1_ms_per_launch": ms_d, "live_line_fraction": live_d, "roofline_frac": ab_d / ms_d / 1e6 / HBM_PEAK_GBS,
                                     "multigrid_pcg_to_1e-6": {"converged": bool(mdone), "iterations": mits, "launches": ld, "relres": mrel, "seconds": mdt}}
            bufd.free()
        except oia.OpenImpalaError as e:
            other_dirs["error"] = str(e)

    # ---- the reference's default calculation_method on the same structure: three periodic cell problems + D_eff ----
    homog = None
    if not args.no_variants and world == 1 and comm is None and args.kernel not in ("unfused", "lean"):
        if t is not None:
            t.destroy()
            t = None
        import numpy as np
        bufh, _ = make_structure(oia, args, shape, z0, nzl)
        ph = bufh.download(np.uint8, shape)
        bufh.free()
        free_before = oia.device_memory()[0]
        h0 = time.perf_counter()
        eh = oia.EffectiveDiffusivityMI355(ph, 1, eps=1e-9, precond=oia.PRECOND_MG, device=local_rank)
        hc = time.perf_counter()
        res, hs = [], []
        for k in range(3):
            res.append(eh.solve(k))
            hs.append(time.perf_counter())
        D = eh.tensor()
        h1 = time.perf_counter()
        lay = eh.layout()
        eh.destroy()
        del ph
        # (GPUTEST_r02 showed 3.7 s here against 1.25 s in every run of the builder: the split says where such a run spends it
        # -- `create` holds the 1 GB pageable upload of the phase bytes, the mask and the multigrid hierarchy)
        homog = {"what": "EffectiveDiffusivityHypre path: chi_x, chi_y, chi_z to relres 1e-9 (periodic multigrid PCG) + tensor",
                 "seconds_incl_upload": h1 - h0,
                 "seconds_split": {"create_incl_pageable_upload": hc - h0, "solve_x": hs[0] - hc, "solve_y": hs[1] - hs[0],
                                   "solve_z": hs[2] - hs[1], "tensor": h1 - hs[2]},
                 "device_free_GB_before": free_before / 1e9,
                 # (round 4: even nx on one GPU -> the flow-through path's compacted vectors and rotated multigrid iteration)
                 "vectors": {"compacted_lines": bool(lay[0]), "GB_per_vector": lay[1] * 8 / 1e9},
                 "iterations": [r[1] for r in res], "converged": [bool(r[0]) for r in res],
                 "D_diag": [float(D[i, i]) for i in range(3)]}

    extras["variants"] = variants
    extras["other_dirs"] = other_dirs
    extras["homog"] = homog
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        extras["cpu"] = cpu_baseline(oia, args, cpu_full)
    watchdog.cancel()
    emit()
    if t is not None:
        t.destroy()
    if comm is not None:
        comm.destroy()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
