"""numpy/ctypes front end of the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It wraps oracle/liboi_oracle.so (the C restatement in oi_oracle.c, which
cites the reference file:line of every function) and adds a dependency-free TIFF
reader that restates the reference's raw-bit path
(src/io/TiffReader.cpp:395-436: uncompressed strips, 1-bit MSB-first unless
FILLORDER says otherwise, PHOTOMETRIC ignored, voxel > threshold -> 1 else 0).

Array convention: numpy arrays are shaped (nz, ny, nx), C-contiguous, so that
x is the fastest axis exactly as in the reference's Fortran-ordered boxes.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboi_oracle.so")
_lib = None

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "oi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboi_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def cpu_budget(cap=16):
    """Threads the oracle may use: the smaller of the affinity mask, the cgroup CPU quota
    and `cap` (a GPU box exposes all host cores but grants ~16; oversubscribed OpenMP
    barriers would crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        n = min(n, int(os.environ["OMP_NUM_THREADS"]))
    return max(1, min(n, cap))


def lib():
    global _lib
    if _lib is None:
        build()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # before libgomp initialises
        L = C.CDLL(_LIB_PATH)
        i, ll, d, u64 = C.c_int, C.c_longlong, C.c_double, C.c_uint64
        L.oi_oracle_num_threads.restype = i
        L.oi_oracle_set_threads.argtypes = [i]
        L.oi_oracle_gen_bernoulli.argtypes = [_u8p, i, i, i, u64, u64]
        L.oi_oracle_gen_spheres.argtypes = [_u8p, i, i, i, u64, i, i]
        L.oi_oracle_gen_blobs.argtypes = [_u8p, i, i, i, u64, i, u64]
        L.oi_oracle_volume_fraction.argtypes = [_u8p, ll, i, C.POINTER(ll), C.POINTER(ll)]
        L.oi_oracle_remspot.argtypes = [_u8p, i, i, i]
        L.oi_oracle_remspot_boxes.argtypes = [_u8p, i, i, i, i]
        L.oi_oracle_box_cuts.argtypes = [i, i, C.POINTER(i)]
        L.oi_oracle_box_cuts.restype = i
        L.oi_oracle_active_mask.argtypes = [_u8p, i, i, i, i, i, _u8p]
        L.oi_oracle_active_mask.restype = ll
        L.oi_oracle_fillmtx.argtypes = [_u8p, _u8p, i, i, i, i, i, _f64p, d, d, _f64p, _f64p, _f64p]
        L.oi_oracle_check_matrix.argtypes = [_f64p, _f64p, _u8p, i, i, i, i, d, d]
        L.oi_oracle_check_matrix.restype = ll
        L.oi_oracle_solve_assembled.argtypes = [_f64p, _f64p, _f64p, i, i, i, d, i, i, _f64p,
                                                C.POINTER(i), C.POINTER(d)]
        L.oi_oracle_solve_assembled.restype = i
        L.oi_oracle_codes.argtypes = [_u8p, i, i, i, i, _u8p]
        L.oi_oracle_apply.argtypes = [_u8p, _f64p, i, i, i, _f64p]
        L.oi_oracle_initial_field.argtypes = [_u8p, i, i, i, i, d, d, _f64p]
        L.oi_oracle_bnorm.argtypes = [_u8p, i, i, i, i, d, d]
        L.oi_oracle_bnorm.restype = d
        L.oi_oracle_cg.argtypes = [_u8p, i, i, i, d, d, i, i, _f64p, C.POINTER(i), C.POINTER(d)]
        L.oi_oracle_cg.restype = i
        L.oi_oracle_cg_fixed.argtypes = [_u8p, i, i, i, i, _f64p]
        L.oi_oracle_fluxes.argtypes = [_f64p, _u8p, i, i, i, i, C.POINTER(d), C.POINTER(d),
                                       C.POINTER(ll), C.POINTER(ll)]
        L.oi_oracle_tau.argtypes = [d, i, d, d, i, i, i, i, d, d, C.POINTER(d), C.POINTER(i)]
        L.oi_oracle_tau.restype = d
        L.oi_oracle_effdiff_fillmtx.argtypes = [_u8p, i, i, i, i, i, _f64p, _f64p]
        L.oi_oracle_effdiff_solve.argtypes = [_u8p, i, i, i, i, i, d, i, _f64p, C.POINTER(i), C.POINTER(d)]
        L.oi_oracle_effdiff_solve.restype = i
        L.oi_oracle_effdiff_tensor.argtypes = [_u8p, i, i, i, i, _f64p, _f64p, _f64p, _f64p]
        L.oi_oracle_set_threads(cpu_budget())
        _lib = L
    return _lib


def _dims(a):
    nz, ny, nx = a.shape
    return nx, ny, nz


def thr53(p):
    """Integer threshold of the Bernoulli generator: P(voxel = 1) = thr53 / 2**53."""
    return int(round(float(p) * (1 << 53)))


def gen_bernoulli(shape, seed, porosity):
    out = np.empty(shape, np.uint8)
    nx, ny, nz = _dims(out)
    lib().oi_oracle_gen_bernoulli(out, nx, ny, nz, seed, thr53(porosity))
    return out


def spheres_count(shape, radius, porosity):
    """Number of overlapping spheres giving the target porosity (Boolean model)."""
    nz, ny, nx = shape
    vol = float(nx) * ny * nz
    return max(1, int(round(-np.log(porosity) * vol / (4.0 / 3.0 * np.pi * radius ** 3))))


def gen_spheres(shape, seed, radius, porosity):
    out = np.empty(shape, np.uint8)
    nx, ny, nz = _dims(out)
    lib().oi_oracle_gen_spheres(out, nx, ny, nz, seed, spheres_count(shape, radius, porosity),
                                int(2 * radius))
    return out


def gen_blobs(shape, seed, spacing, level):
    out = np.empty(shape, np.uint8)
    nx, ny, nz = _dims(out)
    lib().oi_oracle_gen_blobs(out, nx, ny, nz, seed, spacing, level)
    return out


def volume_fraction(phase, phase_id):
    pc, tc = C.c_longlong(), C.c_longlong()
    p = np.ascontiguousarray(phase, np.uint8)
    lib().oi_oracle_volume_fraction(p.reshape(-1), p.size, phase_id, C.byref(pc), C.byref(tc))
    return pc.value, tc.value


def remspot(phase, passes=1, box_size=0):
    """box_size <= 0: the whole domain as one box; > 0: per box of ba.maxSize(box_size), as the reference runs it."""
    q = np.ascontiguousarray(phase, np.uint8).copy()
    nx, ny, nz = _dims(q)
    for _ in range(passes):
        if box_size > 0:
            lib().oi_oracle_remspot_boxes(q, nx, ny, nz, int(box_size))
        else:
            lib().oi_oracle_remspot(q, nx, ny, nz)
    return q


def box_cuts(n, chunk):
    first = (C.c_int * (n + 2))()
    nb = lib().oi_oracle_box_cuts(int(n), int(chunk), first)
    return [first[b] for b in range(nb + 1)]


def active_mask(phase, phase_id, direction):
    p = np.ascontiguousarray(phase, np.uint8)
    nx, ny, nz = _dims(p)
    mask = np.empty_like(p)
    n = lib().oi_oracle_active_mask(p, nx, ny, nz, phase_id, direction, mask)
    return mask, int(n)


def fillmtx(phase, mask, phase_id, direction, vlo, vhi):
    p = np.ascontiguousarray(phase, np.uint8)
    nx, ny, nz = _dims(p)
    n = p.size
    a = np.empty(7 * n)
    rhs = np.empty(n)
    xinit = np.empty(n)
    lib().oi_oracle_fillmtx(p, mask, nx, ny, nz, phase_id, direction, np.ones(3), vlo, vhi, a, rhs, xinit)
    return a.reshape(n, 7), rhs, xinit


def check_matrix(a, rhs, mask, direction, vlo, vhi):
    nx, ny, nz = _dims(mask)
    return int(lib().oi_oracle_check_matrix(np.ascontiguousarray(a).reshape(-1), rhs, mask, nx, ny, nz,
                                            direction, vlo, vhi))


def solve_assembled(a, rhs, xinit, shape, eps=1e-9, maxiter=100000, jacobi=False):
    nz, ny, nx = shape
    x = np.empty(nx * ny * nz)
    it, rel = C.c_int(), C.c_double()
    ok = lib().oi_oracle_solve_assembled(np.ascontiguousarray(a).reshape(-1), rhs, xinit, nx, ny, nz, eps,
                                         maxiter, int(jacobi), x, C.byref(it), C.byref(rel))
    return x.reshape(shape), bool(ok), it.value, rel.value


def codes(mask, direction):
    nx, ny, nz = _dims(mask)
    code = np.empty_like(mask)
    lib().oi_oracle_codes(mask, nx, ny, nz, direction, code)
    return code


def apply(code, v):
    nx, ny, nz = _dims(code)
    v = np.ascontiguousarray(v, np.float64)
    q = np.empty_like(v)
    lib().oi_oracle_apply(code, v, nx, ny, nz, q)
    return q


def initial_field(mask, direction, vlo, vhi):
    nx, ny, nz = _dims(mask)
    x = np.empty(mask.shape)
    lib().oi_oracle_initial_field(mask, nx, ny, nz, direction, vlo, vhi, x)
    return x


def bnorm(mask, direction, vlo, vhi):
    nx, ny, nz = _dims(mask)
    return float(lib().oi_oracle_bnorm(mask, nx, ny, nz, direction, vlo, vhi))


def cg(code, x0, bn, eps=1e-9, maxiter=100000, jacobi=False):
    nx, ny, nz = _dims(code)
    x = np.ascontiguousarray(x0, np.float64).copy()
    it, rel = C.c_int(), C.c_double()
    ok = lib().oi_oracle_cg(code, nx, ny, nz, bn, eps, maxiter, int(jacobi), x, C.byref(it), C.byref(rel))
    return x, bool(ok), it.value, rel.value


def cg_fixed(code, x0, niter):
    nx, ny, nz = _dims(code)
    x = np.ascontiguousarray(x0, np.float64).copy()
    lib().oi_oracle_cg_fixed(code, nx, ny, nz, niter, x)
    return x


def fluxes(x, mask, direction):
    nx, ny, nz = _dims(mask)
    fi, fo = C.c_double(), C.c_double()
    ni, no = C.c_longlong(), C.c_longlong()
    lib().oi_oracle_fluxes(np.ascontiguousarray(x, np.float64), mask, nx, ny, nz, direction, C.byref(fi),
                           C.byref(fo), C.byref(ni), C.byref(no))
    return fi.value, fo.value, ni.value, no.value


def tau(active_vf, converged, flux_in, flux_out, shape, direction, vlo, vhi):
    nz, ny, nx = shape
    deff, st = C.c_double(), C.c_int()
    t = lib().oi_oracle_tau(active_vf, int(converged), flux_in, flux_out, nx, ny, nz, direction, vlo, vhi,
                            C.byref(deff), C.byref(st))
    return float(t), deff.value, st.value


def tortuosity(phase, phase_id, direction, vlo=0.0, vhi=1.0, eps=1e-9, maxiter=100000, jacobi=False,
               assembled=False):
    """Whole path: mask -> system -> solve -> flux -> tau (TortuosityHypre ctor + value())."""
    phase = np.ascontiguousarray(phase, np.uint8)
    mask, nact = active_mask(phase, phase_id, direction)
    active_vf = nact / phase.size
    out = dict(active_vf=active_vf, n_active=nact, converged=False, iterations=-1, relres=float("nan"),
               flux_in=0.0, flux_out=0.0, deff=0.0, mask=mask, x=None)
    if active_vf <= np.finfo(float).eps:
        out.update(tau=float("nan"), status=2)
        return out
    if assembled:
        a, rhs, xinit = fillmtx(phase, mask, phase_id, direction, vlo, vhi)
        x, ok, it, rel = solve_assembled(a, rhs, xinit, phase.shape, eps, maxiter, jacobi)
    else:
        code = codes(mask, direction)
        x0 = initial_field(mask, direction, vlo, vhi)
        x, ok, it, rel = cg(code, x0, bnorm(mask, direction, vlo, vhi), eps, maxiter, jacobi)
        out["n_free"] = int(np.count_nonzero(code))
    fi, fo, _, _ = fluxes(x, mask, direction)
    t, deff, st = tau(active_vf, ok, fi, fo, phase.shape, direction, vlo, vhi)
    out.update(converged=ok, iterations=it, relres=rel, flux_in=fi, flux_out=fo, deff=deff, tau=t,
               status=st, x=x)
    return out


# ---------------------------------------------------------------------------
# homogenization path (EffectiveDiffusivityHypre + Diffusion.cpp tensor integrator)
# ---------------------------------------------------------------------------
def effdiff_fillmtx(phase, phase_id, dir_k):
    p = np.ascontiguousarray(phase, np.uint8)
    nx, ny, nz = _dims(p)
    a = np.empty(7 * p.size)
    rhs = np.empty(p.size)
    lib().oi_oracle_effdiff_fillmtx(p, nx, ny, nz, phase_id, dir_k, a, rhs)
    return a.reshape(-1, 7), rhs


def effdiff_solve(phase, phase_id, dir_k, eps=1e-9, maxiter=100000):
    p = np.ascontiguousarray(phase, np.uint8)
    nx, ny, nz = _dims(p)
    chi = np.empty(p.shape)
    it, rel = C.c_int(), C.c_double()
    ok = lib().oi_oracle_effdiff_solve(p, nx, ny, nz, phase_id, dir_k, eps, maxiter, chi, C.byref(it), C.byref(rel))
    return chi, bool(ok), it.value, rel.value


def effdiff_tensor(phase, phase_id, chis):
    p = np.ascontiguousarray(phase, np.uint8)
    nx, ny, nz = _dims(p)
    D = np.empty(9)
    lib().oi_oracle_effdiff_tensor(p, nx, ny, nz, phase_id, *[np.ascontiguousarray(c, np.float64) for c in chis], D)
    return D.reshape(3, 3)


def effective_diffusivity(phase, phase_id, eps=1e-9, maxiter=100000):
    """Diffusion.cpp homogenization branch (:509-589): three chi_k solves, then the tensor."""
    chis, its = [], []
    for k in range(3):
        chi, ok, it, rel = effdiff_solve(phase, phase_id, k, eps, maxiter)
        if not ok:
            return None, its
        chis.append(chi)
        its.append(it)
    return effdiff_tensor(phase, phase_id, chis), its


# ---------------------------------------------------------------------------
# TIFF raw-bit reader (restates src/io/TiffReader.cpp:139-202 metadata walk and
# :395-436 strip decoding for uncompressed data; page index = z, row = y).
# ---------------------------------------------------------------------------
def read_tiff_stack(path):
    """Returns (raw values as float64 array (nz, ny, nx), bits_per_sample)."""
    f = open(path, "rb").read()
    bo = "<" if f[:2] == b"II" else ">"
    if struct.unpack(bo + "H", f[2:4])[0] != 42:
        raise ValueError("not a classic TIFF")
    off = struct.unpack(bo + "I", f[4:8])[0]
    tsz = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 16: 8}
    tfm = {1: "B", 3: "H", 4: "I", 16: "Q"}
    pages = []
    while off:
        cnt = struct.unpack(bo + "H", f[off:off + 2])[0]
        tags = {}
        for e in range(cnt):
            ent = f[off + 2 + 12 * e: off + 14 + 12 * e]
            tag, typ, c = struct.unpack(bo + "HHI", ent[:8])
            if typ not in tfm:
                continue
            nbytes = tsz[typ] * c
            data = ent[8:8 + nbytes] if nbytes <= 4 else f[struct.unpack(bo + "I", ent[8:12])[0]:][:nbytes]
            tags[tag] = list(struct.unpack(bo + tfm[typ] * c, data))
        w, h = tags[256][0], tags[257][0]
        bps = tags.get(258, [1])[0]
        if tags.get(259, [1])[0] != 1:
            raise ValueError("compressed TIFF not supported by the oracle reader")
        fill = tags.get(266, [1])[0]
        fmt = tags.get(339, [1])[0]
        raw = b"".join(f[o:o + n] for o, n in zip(tags[273], tags[279]))
        if bps == 1:
            bpl = (w + 7) // 8
            rows = np.frombuffer(raw, np.uint8)[: bpl * h].reshape(h, bpl)
            bits = np.unpackbits(rows, axis=1, bitorder="big" if fill == 1 else "little")[:, :w]
            page = bits.astype(np.float64)
        else:
            kind = {1: "u", 2: "i", 3: "f"}[fmt]
            dt = np.dtype(f"{bo}{kind}{bps // 8}")
            page = np.frombuffer(raw, dt)[: w * h].reshape(h, w).astype(np.float64)
        pages.append(page)
        off = struct.unpack(bo + "I", f[off + 2 + 12 * cnt: off + 6 + 12 * cnt])[0]
    return np.stack(pages), bps


def read_tiff_phase(path, threshold=0.5):
    raw, _ = read_tiff_stack(path)
    return (raw > threshold).astype(np.uint8)


# ---- CathodeWrite text exporters (reference src/io/CathodeWrite.cpp) ----------------------------------------------------------
# Test infrastructure like everything in this file: tests/test_cathode_cpu.py compares apps/CathodeWrite.H's files with these.

def cathode_dandeliion_text(vf_solid, radius, conductivity, cmax):
    """CathodeWrite::writeDandeLiionParameters (CathodeWrite.cpp:35-103): scientific notation, 8 digits (:62);
    el = 1 - vf_solid (:48), bet = 3 vf_solid / R or 0 for a vanishing radius (:52-57), B = el / 1.94 (:59)."""
    e = lambda v: "%.8e" % v
    el = 1.0 - vf_solid
    bet = 3.0 * vf_solid / radius if radius > np.finfo(np.float64).eps else 0.0
    bar = "#" * 59
    return "\n".join([
        bar, "# Cathode parameter container (Generated by CathodeWrite) #", "# Units should correspond to DandeLiion expectations.     #", bar, "",
        "N = 55                # Number of mesh nodes in liquid (Example value)",
        "M = 50                # Number of mesh nodes in solid (Example value)", "",
        "R = " + e(radius) + "          # Particle radius, m", "",
        "el      = " + e(el) + "        # Volume fraction of electrolyte, Calculated (1 - vf_solid)",
        "bet     = " + e(bet) + "  # BET surface area, m^-1 (Verify formula: 3*vf_solid/R ?)",
        "B       = " + e(el / 1.94) + "   # Permeability factor of electrolyte (Verify formula: el/1.94 ?)",
        "cmax    = " + e(cmax) + "   # Maximum concentration of Li ions in solid, mol m^-3",
        "sigma_s = " + e(conductivity) + "    # Solid conductivity, S m^-1", bar, ""])


def cathode_pybamm_text(vf_solid, radius, conductivity, cmax):
    """CathodeWrite::writePyBammParameters (CathodeWrite.cpp:106-175): fixed notation, 8 decimals (:135)."""
    f = lambda v: "%.8f" % v
    el = 1.0 - vf_solid
    area = 3.0 * vf_solid / radius if radius > np.finfo(np.float64).eps else 0.0
    c = ",OpenImpala Calculated Parameter,"
    return "\n".join([
        "Name [units],Value,Reference,Notes", "# Parameter file generated by CathodeWrite,,,", "# Units should correspond to PyBamm expectations,,,",
        ",,,", "# Electrode properties,,,",
        "Positive electrode conductivity [S.m-1]," + f(conductivity) + c,
        "Maximum concentration in positive electrode [mol.m-3]," + f(cmax) + c,
        "Positive electrode diffusivity [m2.s-1],[function]lico2_diffusivity_Dualfoil1998,,Requires Function Definition in PyBamm",
        "Positive electrode OCP [V],[function]lico2_ocp_Dualfoil1998,,", ",,,", "# Microstructure,,,",
        "Positive electrode porosity," + f(el) + c + "electrolyte volume fraction (1 - vf_solid)",
        "Positive electrode active material volume fraction," + f(vf_solid) + c,
        "Positive particle radius [m]," + f(radius) + c,
        "Positive electrode surface area density [m-1]," + f(area) + c + "Calculated as 3*vf_solid/R (Verify formula)", ",,,",
        "# Interfacial reactions,,,", "Positive electrode cation signed stoichiometry,-1,,", "Positive electrode electrons in reaction,1,,",
        ",,,", ",,,", ""])
