"""openimpala_amd -- MI355X-native flow-through tortuosity solver (host-side mirror).

A thin ctypes binding of the C ABI in include/openimpala_amd.h plus Python mirrors of
the reference's operator interface for this path:

    reference (C++)                                   here
    OpenImpala::Direction {X,Y,Z}                     Direction        (Tortuosity.H:9-13)
    OpenImpala::Tortuosity::value(refresh)            TortuosityMI355.value(refresh)
    TortuosityHypre(geom, ba, dm, phase, vf, ...)     TortuosityMI355(phase, phase_id, dir, ...)
    TortuosityHypre::checkMatrixProperties()          .checkMatrixProperties()
    getSolverConverged() ... getActiveVolumeFraction() same names          (TortuosityHypre.H:114-120)
    VolumeFraction(mf, phase).value(pc, tc)           VolumeFraction(phase, id).value()
    TiffReader(file).threshold(thr, 1, 0, mf)         TiffReader(file).threshold(thr)

All compute happens in the HIP library (openimpala_amd/libopenimpala_amd.so).  There is
no CPU fallback: without the built library or without a GPU every call raises.
Arrays are numpy uint8 / float64 shaped (nz, ny, nx), x fastest -- the reference's
Fortran box order and the TIFF/RAW payload order.
"""
import ctypes as C
import enum
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OI_LIB_PATH", os.path.join(_HERE, "libopenimpala_amd.so"))  # (override: A/B builds)
ABI_VERSION = 2

OI_OK, OI_BAD_ARG, OI_NO_PERCOLATION, OI_NOT_CONVERGED, OI_FLUX_NOT_CONSERVED = 0, 1, 2, 3, 4
OI_HIP_ERROR, OI_RCCL_ERROR, OI_NO_DEVICE, OI_IO_ERROR = 5, 6, 7, 8
PRECOND_NONE, PRECOND_JACOBI, PRECOND_MG, PRECOND_DEFAULT = 0, 1, 2, 3
KERNEL_AUTO, KERNEL_UNFUSED, KERNEL_FUSED, KERNEL_FUSED1, KERNEL_LEAN = 0, 1, 2, 3, 4
STORE_F64, STORE_F32 = 0, 1
COMM_ID_BYTES = 128

# every symbol include/openimpala_amd.h declares (checked by the CPU test-suite)
ABI_SYMBOLS = (
    "oi_tort_default_opts oi_tort_create oi_tort_create_device oi_tort_create_slab oi_tort_destroy "
    "oi_tort_value oi_tort_check_matrix_properties oi_tort_get_stats oi_tort_get_solution oi_tort_get_mask "
    "oi_volume_fraction oi_remspot oi_effdiff_create oi_effdiff_create_slab oi_effdiff_solve oi_effdiff_get_chi oi_effdiff_tensor "
    "oi_effdiff_info oi_effdiff_layout oi_effdiff_destroy oi_tort_set_eps oi_tort_cg_begin oi_tort_cg_iterate oi_tort_cg_run oi_tort_cg_finish oi_tort_sync "
    "oi_tort_cg_state oi_tort_debug_apply oi_tort_debug_get_codes oi_tort_true_residual oi_debug_cg1_grid oi_tort_placement oi_tort_placement_cost oi_tort_bench_stream oi_tort_bench_apply oi_tort_live_fraction oi_tort_kernel_times "
    "oi_kernel_name oi_tort_reset_kernel_times oi_tort_set_tuning oi_tort_set_apply_tuning oi_tort_set_cg1_tuning oi_tort_set_cg1_layout oi_tort_line_voxels oi_tort_set_option oi_tort_retune oi_comm_unique_id oi_comm_create "
    "oi_comm_create_local oi_comm_destroy oi_comm_rank oi_comm_size oi_comm_info oi_gen_bernoulli_device "
    "oi_gen_spheres_device oi_gen_blobs_device oi_blobs_level oi_dev_malloc oi_dev_mem_info oi_dev_stream_copy_bench oi_dev_free oi_dev_pool oi_dev_pool_trim oi_dev_pool_stats oi_dev_live_bytes oi_dev_live_blocks oi_dev_peak_bytes oi_debug_fail_alloc_after oi_dev_upload oi_dev_download oi_device_count "
    "oi_last_error oi_status_string oi_abi_version oi_build_has_xcheck oi_read_tiff oi_read_tiff_planes oi_read_tiff_sequence oi_read_raw oi_read_raw_planes oi_read_hdf5 oi_read_hdf5_planes oi_read_dat oi_free_host oi_read_set_pinned"
).split()


class Direction(enum.IntEnum):
    """OpenImpala::Direction (src/props/Tortuosity.H:9-13)."""
    X = 0
    Y = 1
    Z = 2


class OpenImpalaError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"openimpala_amd: status {status}: {message}")
        self.status = status


class Opts(C.Structure):
    _fields_ = [("eps", C.c_double), ("maxiter", C.c_int), ("precond", C.c_int), ("remspot_passes", C.c_int),
                ("remspot_box_size", C.c_int), ("verbose", C.c_int), ("check_every", C.c_int), ("kernel", C.c_int), ("profile", C.c_int),
                ("device", C.c_int), ("p_storage", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("converged", C.c_int), ("iterations", C.c_int), ("final_relres", C.c_double),
                ("flux_in", C.c_double), ("flux_out", C.c_double), ("active_vf", C.c_double), ("vf", C.c_double),
                ("deff", C.c_double), ("tau", C.c_double), ("n_active", C.c_int64), ("n_free", C.c_int64),
                ("n_phase", C.c_int64), ("flood_sweeps", C.c_int), ("t_upload_s", C.c_double),
                ("t_mask_s", C.c_double), ("t_setup_s", C.c_double), ("t_solve_s", C.c_double),
                ("t_flux_s", C.c_double), ("bytes_device", C.c_double)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OpenImpalaError(OI_NO_DEVICE, f"{LIB_PATH} is missing: run __graft_entry__.build() "
                              "(make -C openimpala_amd/csrc); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i, d, u64, i64p = C.c_void_p, C.c_int, C.c_double, C.c_uint64, C.POINTER(C.c_int64)
    pp = C.POINTER(C.c_void_p)
    L.oi_abi_version.restype = i
    L.oi_build_has_xcheck.restype = i
    L.oi_last_error.restype = C.c_char_p
    L.oi_status_string.restype = C.c_char_p
    L.oi_status_string.argtypes = [i]
    L.oi_kernel_name.restype = C.c_char_p
    L.oi_kernel_name.argtypes = [i]
    L.oi_tort_default_opts.argtypes = [C.POINTER(Opts)]
    L.oi_tort_default_opts.restype = None
    L.oi_tort_create.argtypes = [pp, vp, i, i, i, i, i, d, d, C.POINTER(Opts)]
    L.oi_tort_create_device.argtypes = [pp, vp, i, i, i, i, i, d, d, C.POINTER(Opts)]
    L.oi_tort_create_slab.argtypes = [pp, vp, vp, i, i, i, i, i, i, i, i, d, d, C.POINTER(Opts)]
    L.oi_tort_destroy.argtypes = [vp]
    L.oi_tort_destroy.restype = None
    L.oi_tort_value.argtypes = [vp, i, C.POINTER(d)]
    L.oi_tort_check_matrix_properties.argtypes = [vp, i64p]
    L.oi_tort_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.oi_tort_get_solution.argtypes = [vp, vp]
    L.oi_tort_get_mask.argtypes = [vp, vp]
    L.oi_volume_fraction.argtypes = [vp, C.c_size_t, i, i64p, i64p]
    L.oi_remspot.argtypes = [vp, i, i, i, i, i]
    L.oi_effdiff_create.argtypes = [pp, vp, i, i, i, i, C.POINTER(Opts)]
    L.oi_effdiff_create_slab.argtypes = [pp, vp, vp, i, i, i, i, i, i, i, C.POINTER(Opts)]
    L.oi_effdiff_solve.argtypes = [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(d)]
    L.oi_effdiff_get_chi.argtypes = [vp, i, vp]
    L.oi_effdiff_tensor.argtypes = [vp, C.POINTER(d)]
    L.oi_effdiff_info.argtypes = [vp, i64p, C.POINTER(i), C.POINTER(d)]
    L.oi_effdiff_layout.argtypes = [vp, C.POINTER(i), i64p]
    L.oi_effdiff_destroy.argtypes = [vp]
    L.oi_effdiff_destroy.restype = None
    L.oi_tort_set_eps.argtypes = [vp, d]
    L.oi_tort_cg_begin.argtypes = [vp]
    L.oi_tort_cg_iterate.argtypes = [vp, i]
    L.oi_tort_cg_finish.argtypes = [vp]
    L.oi_tort_cg_run.argtypes = [vp, i, C.POINTER(i)]
    L.oi_tort_sync.argtypes = [vp]
    L.oi_tort_cg_state.argtypes = [vp, C.POINTER(i), C.POINTER(d), C.POINTER(i)]
    L.oi_tort_debug_apply.argtypes = [vp, vp, vp, i]
    L.oi_tort_debug_get_codes.argtypes = [vp, vp]
    L.oi_tort_bench_apply.argtypes = [vp, i, i, C.POINTER(d)]
    L.oi_tort_live_fraction.argtypes = [vp, i, C.POINTER(d)]
    L.oi_tort_kernel_times.argtypes = [vp, i, C.POINTER(i), C.POINTER(d), i64p]
    L.oi_tort_kernel_times.restype = i
    L.oi_tort_reset_kernel_times.argtypes = [vp]
    L.oi_tort_reset_kernel_times.restype = None
    L.oi_tort_set_tuning.argtypes = [i, i]
    L.oi_tort_set_tuning.restype = None
    L.oi_tort_set_apply_tuning.argtypes = [i]
    L.oi_tort_set_apply_tuning.restype = None
    L.oi_tort_set_cg1_tuning.argtypes = [i, i]
    L.oi_tort_set_cg1_tuning.restype = None
    L.oi_tort_set_cg1_layout.argtypes = [i]
    L.oi_tort_set_cg1_layout.restype = None
    L.oi_tort_set_option.argtypes = [C.c_char_p, C.c_double]
    L.oi_tort_retune.argtypes = [vp]
    L.oi_tort_true_residual.argtypes = [vp, C.POINTER(C.c_double)]
    L.oi_read_set_pinned.argtypes = [i]
    L.oi_read_set_pinned.restype = None
    L.oi_debug_cg1_grid.argtypes = [i, i, i, i, i, i] + [C.POINTER(C.c_int)] * 4
    L.oi_tort_bench_stream.argtypes = [vp, i, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.oi_tort_placement.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.oi_tort_placement_cost.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.oi_tort_line_voxels.argtypes = []
    L.oi_tort_line_voxels.restype = i
    L.oi_comm_unique_id.argtypes = [vp]
    L.oi_comm_create.argtypes = [pp, vp, i, i, i]
    L.oi_comm_create_local.argtypes = [pp, i, i]
    L.oi_comm_destroy.argtypes = [vp]
    L.oi_comm_destroy.restype = None
    L.oi_comm_rank.argtypes = [vp]
    L.oi_comm_size.argtypes = [vp]
    L.oi_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.oi_gen_bernoulli_device.argtypes = [vp, i, i, i, i, i, u64, u64]
    L.oi_gen_spheres_device.argtypes = [vp, i, i, i, i, i, u64, i, i]
    L.oi_gen_blobs_device.argtypes = [vp, i, i, i, i, i, u64, i, u64]
    L.oi_blobs_level.argtypes = [u64, i, d]
    L.oi_blobs_level.restype = u64
    L.oi_dev_malloc.argtypes = [pp, C.c_size_t]
    L.oi_dev_mem_info.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.oi_dev_stream_copy_bench.argtypes = [C.c_size_t, i, C.POINTER(C.c_double)]
    L.oi_dev_free.argtypes = [vp]
    L.oi_dev_pool.argtypes = [i, d]
    L.oi_dev_pool_trim.restype = d
    L.oi_dev_pool_stats.argtypes = [C.POINTER(d), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(i)]
    L.oi_dev_live_bytes.restype = C.c_longlong
    L.oi_dev_live_blocks.restype = C.c_longlong
    L.oi_dev_peak_bytes.argtypes = [i]
    L.oi_dev_peak_bytes.restype = C.c_longlong
    L.oi_debug_fail_alloc_after.argtypes = [C.c_longlong]
    L.oi_debug_fail_alloc_after.restype = None
    L.oi_dev_upload.argtypes = [vp, vp, C.c_size_t]
    L.oi_dev_download.argtypes = [vp, vp, C.c_size_t]
    L.oi_device_count.restype = i
    L.oi_read_tiff.argtypes = [C.c_char_p, d, pp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.oi_read_tiff_planes.argtypes = [C.c_char_p, d, i, i, pp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.oi_read_tiff_sequence.argtypes = [C.c_char_p, i, i, i, C.c_char_p, d, pp, C.POINTER(i), C.POINTER(i), C.POINTER(i),
                                        C.POINTER(i)]
    L.oi_read_raw.argtypes = [C.c_char_p, i, i, i, i, d, pp]
    L.oi_read_raw_planes.argtypes = [C.c_char_p, i, i, i, i, d, i, i, pp]
    L.oi_read_hdf5.argtypes = [C.c_char_p, C.c_char_p, d, pp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.oi_read_hdf5_planes.argtypes = [C.c_char_p, C.c_char_p, d, i, i, pp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.oi_read_dat.argtypes = [C.c_char_p, d, pp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.oi_free_host.argtypes = [vp]
    L.oi_free_host.restype = None
    if L.oi_abi_version() != ABI_VERSION:
        raise OpenImpalaError(OI_BAD_ARG, "ABI version mismatch between the python binding and the library")
    _lib = L
    return L


def _check(status, allow=()):
    if status != OI_OK and status not in allow:
        L = lib()
        raise OpenImpalaError(status, f"{L.oi_status_string(status).decode()}: {L.oi_last_error().decode()}")
    return status


def device_count():
    return lib().oi_device_count()


def default_opts(**kw):
    o = Opts()
    lib().oi_tort_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k}")
        setattr(o, k, v)
    return o


def set_tuning(fused_tile_rows=16, fused_zchunk=0, apply_tile_rows=32, cg1_tile_rows=0, cg1_x_every_second=1):
    """Rows per LDS tile (8, 16 or 32) of the fused CG kernel / the stand-alone apply / the
    single-kernel CG iteration (0 = automatic), planes per block of the z-march (0 = auto),
    and whether the single-kernel iteration touches x only every second iteration."""
    lib().oi_tort_set_tuning(fused_tile_rows, fused_zchunk)
    lib().oi_tort_set_apply_tuning(apply_tile_rows)
    lib().oi_tort_set_cg1_tuning(cg1_tile_rows, cg1_x_every_second)


# ------------------------------------------------------------------------------------------
# device buffers and seeded synthetic structures resident in HBM
# ------------------------------------------------------------------------------------------
class DeviceBuffer:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _check(lib().oi_dev_malloc(C.byref(self.ptr), self.nbytes))

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        _check(lib().oi_dev_upload(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        _check(lib().oi_dev_download(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().oi_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def thr53(porosity):
    return int(round(float(porosity) * (1 << 53)))


def spheres_count(shape, radius, porosity):
    nz, ny, nx = shape
    return max(1, int(round(-math.log(porosity) * float(nx) * ny * nz / (4.0 / 3.0 * math.pi * radius ** 3))))


def gen_bernoulli_device(shape, seed, porosity, z0=0, nz_local=None):
    """Bernoulli two-phase cube (BASELINE.json config 2) generated in HBM; shape = global (nz, ny, nx)."""
    nz, ny, nx = shape
    nzl = nz if nz_local is None else nz_local
    buf = DeviceBuffer(nx * ny * nzl)
    _check(lib().oi_gen_bernoulli_device(buf.ptr, nx, ny, nz, z0, nzl, seed, thr53(porosity)))
    return buf


def gen_spheres_device(shape, seed, radius, porosity, z0=0, nz_local=None):
    """Overlapping-sphere pack (BASELINE.json config 3) generated in HBM; pore = 1."""
    nz, ny, nx = shape
    nzl = nz if nz_local is None else nz_local
    buf = DeviceBuffer(nx * ny * nzl)
    _check(lib().oi_gen_spheres_device(buf.ptr, nx, ny, nz, z0, nzl, seed, spheres_count(shape, radius, porosity),
                                       int(2 * radius)))
    return buf


def blobs_level(seed, spacing, porosity):
    """Threshold level (0..65535) of the blob generator for a target porosity (host arithmetic)."""
    return int(lib().oi_blobs_level(seed, spacing, porosity))


def gen_blobs_device(shape, seed, spacing, porosity, z0=0, nz_local=None):
    """Correlated low-porosity structure (BASELINE.json config 5) generated in HBM; pore = 1."""
    nz, ny, nx = shape
    nzl = nz if nz_local is None else nz_local
    buf = DeviceBuffer(nx * ny * nzl)
    _check(lib().oi_gen_blobs_device(buf.ptr, nx, ny, nz, z0, nzl, seed, spacing, blobs_level(seed, spacing, porosity)))
    return buf


# ------------------------------------------------------------------------------------------
# communicator
# ------------------------------------------------------------------------------------------
class Comm:
    """One rank of the slab chain (RCCL over xGMI, one process per GPU)."""

    def __init__(self, handle, owned=True):
        self._h = handle
        self._owned = owned

    @staticmethod
    def unique_id():
        buf = (C.c_char * COMM_ID_BYTES)()
        _check(lib().oi_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, uid, nranks, rank, device=-1):
        h = C.c_void_p()
        _check(lib().oi_comm_create(C.byref(h), uid, nranks, rank, device))
        return cls(h)

    @classmethod
    def create_local(cls, nranks, device=-1):
        """`nranks` in-process communicators for `nranks` host threads on one GPU (tests)."""
        arr = (C.c_void_p * nranks)()
        _check(lib().oi_comm_create_local(arr, nranks, device))
        return [cls(C.c_void_p(arr[r])) for r in range(nranks)]

    @property
    def rank(self):
        return lib().oi_comm_rank(self._h)

    @property
    def size(self):
        return lib().oi_comm_size(self._h)

    def info(self):
        """What the transport is: kind ('rccl' | 'local'), ncclCommCount of the all-reduce and of the plane-exchange
        communicator, and whether the exchange overlaps the all-reduce on a stream of its own."""
        v = [C.c_int(0) for _ in range(4)]
        _check(lib().oi_comm_info(self._h, *[C.byref(x) for x in v]))
        return {"kind": {1: "rccl", 2: "local"}.get(v[0].value, "?"), "ranks_collective": v[1].value, "ranks_p2p": v[2].value,
                "exchange_overlaps": bool(v[3].value)}

    def destroy(self):
        if self._h and self._owned:
            lib().oi_comm_destroy(self._h)
        self._h = None


def set_cg1_layout(compact_lines=1):
    """Layout of r, q, p in the single-kernel CG iteration for handles created afterwards: 1 = compacted
    lines (default; also r, q, z, p of the multigrid-preconditioned iteration), 0 = dense arrays with predicated
    accesses.  +2: pair-granular accesses (A/B switch)."""
    lib().oi_tort_set_cg1_layout(int(compact_lines))


def stream_copy_bench(nbytes, reps=10):
    """ms per launch of the library's reference copy stream between two scratch arrays of nbytes each."""
    ms = C.c_double()
    _check(lib().oi_dev_stream_copy_bench(int(nbytes), int(reps), C.byref(ms)))
    return ms.value


def set_option(key, value):
    """Named switch of the process defaults (see include/openimpala_amd.h); new handles only."""
    _check(lib().oi_tort_set_option(key.encode(), float(value)))


def cg1_grid(nx, ny, nz_local, descr=1, split=0, part=0):
    """(tile rows, planes per block, blocks, capacity of the partial-sum array) of the single-kernel iteration: host arithmetic."""
    v = [C.c_int(0) for _ in range(4)]
    _check(lib().oi_debug_cg1_grid(nx, ny, nz_local, descr, split, part, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def read_set_pinned(on=True):
    """Readers return page-locked buffers from now on (oi_read_set_pinned): one DMA transfer per upload."""
    lib().oi_read_set_pinned(1 if on else 0)


def device_memory():
    """(free, total) bytes of the current device."""
    f, t = C.c_size_t(), C.c_size_t()
    _check(lib().oi_dev_mem_info(C.byref(f), C.byref(t)))
    return f.value, t.value


def dev_pool(enable, cap_gb=0.0):
    """The process's pool of released handle arrays (include/openimpala_amd.h: oi_dev_pool): on for a loop over handles of one
    shape (directions, a REV study), off (the default) to hand a destroyed handle's memory back to the driver at once."""
    _check(lib().oi_dev_pool(1 if enable else 0, float(cap_gb)))


def dev_pool_trim():
    """Pooled blocks back to the driver; returns the bytes released."""
    return int(lib().oi_dev_pool_trim())


def dev_pool_stats():
    held, blocks, hits, misses, on = C.c_double(), C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_int()
    _check(lib().oi_dev_pool_stats(C.byref(held), C.byref(blocks), C.byref(hits), C.byref(misses), C.byref(on)))
    return {"held_bytes": int(held.value), "held_blocks": blocks.value, "hits": hits.value, "misses": misses.value, "enabled": bool(on.value)}


def has_xcheck():
    """True when the loaded library is the TEST build (libopenimpala_amd_xcheck.so: OI_LIB_PATH points at it) with the superseded
    cross-check kernels; the product library answers OI_BAD_ARG to configurations that need them."""
    return bool(lib().oi_build_has_xcheck())


def dev_live():
    """(bytes, blocks) of the library's device allocations alive now -- the exact ledger of oi_pool.h, not hipMemGetInfo."""
    return int(lib().oi_dev_live_bytes()), int(lib().oi_dev_live_blocks())


def dev_peak_bytes(reset=False):
    """High-water mark of the ledger's live bytes since the last reset."""
    return int(lib().oi_dev_peak_bytes(1 if reset else 0))


def debug_fail_alloc_after(n):
    """Test hook: the n-th device allocation from now fails once (0 disarms)."""
    lib().oi_debug_fail_alloc_after(int(n))


def line_voxels():
    """Voxels per stored line of the compacted CG vectors (8 = one 64-byte HBM burst)."""
    return int(lib().oi_tort_line_voxels())


def _take_bytes(p, n):
    """Copy of n bytes at the malloc'ed address p (np.ctypeslib.as_array is 5x slower on large buffers)."""
    return np.frombuffer((C.c_uint8 * int(n)).from_address(p.value), dtype=np.uint8).copy()


def slab_range(nz, nranks, rank):
    """Planes [z0, z1) of `rank` in the z-slab decomposition (as even as possible, every
    rank gets at least one plane when nz >= nranks)."""
    base, rem = divmod(nz, nranks)
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


# ------------------------------------------------------------------------------------------
# operators
# ------------------------------------------------------------------------------------------
class VolumeFraction:
    """OpenImpala::VolumeFraction (src/props/VolumeFraction.{H,cpp})."""

    def __init__(self, phase, phase_id):
        self._phase = np.ascontiguousarray(phase, np.uint8)
        self._id = int(phase_id)

    def value(self):
        """-> (phase_count, total_count), VolumeFraction.cpp:22-66."""
        pc, tc = C.c_int64(), C.c_int64()
        _check(lib().oi_volume_fraction(self._phase.ctypes.data_as(C.c_void_p), self._phase.size, self._id,
                                        C.byref(pc), C.byref(tc)))
        return pc.value, tc.value

    def value_vf(self):
        pc, tc = self.value()
        return pc / tc if tc else 0.0


def remspot(phase, passes=1, box_size=32):
    """preconditionPhaseFab: `passes` remspot passes per box of ba.maxSize(box_size) (<= 0: one box = the whole domain)."""
    q = np.ascontiguousarray(phase, np.uint8).copy()
    nz, ny, nx = q.shape
    _check(lib().oi_remspot(q.ctypes.data_as(C.c_void_p), nx, ny, nz, passes, int(box_size)))
    return q


class TortuosityMI355:
    """Mirror of OpenImpala::TortuosityHypre (src/props/TortuosityHypre.H:36-186) behind the
    abstract Tortuosity::value(refresh) interface (src/props/Tortuosity.H:47-69).

    phase    : uint8 (nz, ny, nx) host array, or a DeviceBuffer with `shape=` given
    phase_id : which phase conducts
    direction: Direction.X/Y/Z
    vlo, vhi : Dirichlet potentials (defaults of the reference class: 0, 1)
    eps, maxiter: hypre.eps / hypre.maxiter of the reference (TortuosityHypre.cpp:142-149)
    comm + z0 + nz_global: this object is one z-slab of a decomposed volume
    """

    def __init__(self, phase, phase_id, direction, vlo=0.0, vhi=1.0, verbose=0, shape=None, comm=None, z0=0,
                 nz_global=None, **opts):
        self._h = C.c_void_p()
        o = default_opts(verbose=verbose, **opts)
        self.opts = o
        L = lib()
        if isinstance(phase, DeviceBuffer):
            if shape is None:
                raise TypeError("shape=(nz, ny, nx) is required with a device buffer")
            nz, ny, nx = shape
            ptr, on_device = phase.ptr, 1
        else:
            self._host = np.ascontiguousarray(phase, np.uint8)
            nz, ny, nx = self._host.shape
            ptr, on_device = self._host.ctypes.data_as(C.c_void_p), 0
        self.shape = (nz, ny, nx)
        self.direction = Direction(int(direction))
        if comm is not None:
            nzg = nz if nz_global is None else nz_global
            st = L.oi_tort_create_slab(C.byref(self._h), comm._h, ptr, on_device, nx, ny, nzg, z0, nz, phase_id,
                                       int(direction), vlo, vhi, C.byref(o))
        elif on_device:
            st = L.oi_tort_create_device(C.byref(self._h), ptr, nx, ny, nz, phase_id, int(direction), vlo, vhi,
                                         C.byref(o))
        else:
            st = L.oi_tort_create(C.byref(self._h), ptr, nx, ny, nz, phase_id, int(direction), vlo, vhi, C.byref(o))
        _check(st)
        self.status = OI_OK

    # --- Tortuosity interface -------------------------------------------------------------
    def value(self, refresh=False):
        """Tortuosity::value: tau, NaN / +Inf exactly where the reference returns them."""
        tau = C.c_double()
        self.status = _check(lib().oi_tort_value(self._h, int(bool(refresh)), C.byref(tau)),
                             allow=(OI_NO_PERCOLATION, OI_NOT_CONVERGED, OI_FLUX_NOT_CONSERVED))
        return tau.value

    def checkMatrixProperties(self):
        n = C.c_int64()
        _check(lib().oi_tort_check_matrix_properties(self._h, C.byref(n)), allow=(OI_NO_PERCOLATION,))
        return n.value == 0

    # --- getters (TortuosityHypre.H:114-120) ----------------------------------------------
    def stats(self):
        s = Stats()
        _check(lib().oi_tort_get_stats(self._h, C.byref(s)))
        return s

    def getSolverConverged(self):
        return bool(self.stats().converged)

    def getFinalRelativeResidualNorm(self):
        return self.stats().final_relres

    def getSolverIterations(self):
        return self.stats().iterations

    def getFluxIn(self):
        return self.stats().flux_in

    def getFluxOut(self):
        return self.stats().flux_out

    def getActiveVolumeFraction(self):
        return self.stats().active_vf

    # --- fields ---------------------------------------------------------------------------
    def solution(self):
        out = np.empty(self.shape, np.float64)
        _check(lib().oi_tort_get_solution(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def mask(self):
        out = np.empty(self.shape, np.uint8)
        _check(lib().oi_tort_get_mask(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def codes(self):
        out = np.empty(self.shape, np.uint8)
        _check(lib().oi_tort_debug_get_codes(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def bench_stream(self, reps=10):
        """(ms per pass, bytes per pass) of a 3-read / 3-write stream over the handle's own CG vectors (oi_tort_bench_stream)."""
        ms, nb = C.c_double(0.0), C.c_double(0.0)
        _check(lib().oi_tort_bench_stream(self._h, reps, C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def placement(self):
        """Slots of the single-kernel iteration's vectors and what the placement search measured (oi_tort_placement)."""
        n, tried = C.c_int(0), C.c_int(0)
        slots = (C.c_int * 7)()
        first, best = C.c_double(0.0), C.c_double(0.0)
        _check(lib().oi_tort_placement(self._h, C.byref(n), slots, C.byref(tried), C.byref(first), C.byref(best)))
        a, b, pk, kp, hit = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        _check(lib().oi_tort_placement_cost(self._h, C.byref(a), C.byref(b), C.byref(pk), C.byref(kp), C.byref(hit)))
        return {"nslots": n.value, "slots": list(slots)[:6], "x_after_slots": slots[6], "tried": tried.value, "ms_in_order": first.value, "ms_kept": best.value,
                # what it cost: seconds allocating the set (with a spacer, the driver clears what it hands out) and searching,
                # device bytes of the whole handle at the peak of the search, bytes of x + the six vectors kept
                "alloc_s": a.value, "search_s": b.value, "peak_device_bytes": pk.value, "kept_vector_bytes": kp.value, "cache_hit": bool(hit.value)}

    def true_residual(self):
        """||b - A x|| / ||b|| of the current solution, recomputed on the device with the stand-alone apply kernel."""
        v = C.c_double()
        _check(lib().oi_tort_true_residual(self._h, C.byref(v)))
        return v.value

    def retune(self):
        """Take the current process defaults of the launch shapes (set_tuning, set_option) into this handle."""
        _check(lib().oi_tort_retune(self._h))

    def apply(self, v, variant=KERNEL_FUSED):
        v = np.ascontiguousarray(v, np.float64)
        assert v.shape == self.shape
        q = np.empty_like(v)
        _check(lib().oi_tort_debug_apply(self._h, v.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p),
                                         variant))
        return q

    # --- stepping (bench) -----------------------------------------------------------------
    def set_eps(self, eps):
        _check(lib().oi_tort_set_eps(self._h, eps))

    def cg_begin(self):
        _check(lib().oi_tort_cg_begin(self._h))

    def cg_iterate(self, n):
        _check(lib().oi_tort_cg_iterate(self._h, n))

    def cg_run(self, max_launches=0):
        """The library's own convergence loop (what value() runs between cg_begin and cg_finish); returns the launches it issued."""
        n = C.c_int()
        _check(lib().oi_tort_cg_run(self._h, int(max_launches), C.byref(n)))
        return n.value

    def cg_finish(self):
        _check(lib().oi_tort_cg_finish(self._h))

    def sync(self):
        _check(lib().oi_tort_sync(self._h))

    def cg_state(self):
        it, rel, conv = C.c_int(), C.c_double(), C.c_int()
        _check(lib().oi_tort_cg_state(self._h, C.byref(it), C.byref(rel), C.byref(conv)))
        return it.value, rel.value, bool(conv.value)

    def bench_apply(self, reps, variant=KERNEL_FUSED):
        ms = C.c_double()
        _check(lib().oi_tort_bench_apply(self._h, reps, variant, C.byref(ms)))
        return ms.value

    def kernel_times(self):
        ids = (C.c_int * 32)()
        ms = (C.c_double * 32)()
        n = (C.c_int64 * 32)()
        k = lib().oi_tort_kernel_times(self._h, 32, ids, ms, n)
        return {lib().oi_kernel_name(ids[j]).decode(): (ms[j], n[j]) for j in range(k)}

    def reset_kernel_times(self):
        lib().oi_tort_reset_kernel_times(self._h)

    def live_fraction(self, group_voxels=16):
        """Fraction of the aligned x-runs of `group_voxels` voxels holding a free voxel (16 doubles = one
        128-byte line): the lines of r/q/p/x the predicated solver kernels still have to move."""
        f = C.c_double()
        _check(lib().oi_tort_live_fraction(self._h, int(group_voxels), C.byref(f)))
        return f.value

    def destroy(self):
        if self._h:
            lib().oi_tort_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class EffectiveDiffusivityMI355:
    """Mirror of OpenImpala::EffectiveDiffusivityHypre (src/props/EffectiveDiffusivityHypre.{H,cpp})
    plus the tensor integrator of Diffusion.cpp:60-167.  One object serves all three directions."""

    def __init__(self, phase, phase_id, verbose=0, comm=None, z0=0, nz_global=None, **opts):
        """phase: uint8 (nz, ny, nx); with `comm` it is this rank's z-slab [z0, z0 + nz) of a volume of
        nz_global planes spread over the ring of ranks (periodic z across the first and last rank)."""
        self._host = np.ascontiguousarray(phase, np.uint8)
        nz, ny, nx = self._host.shape
        self.shape = (nz, ny, nx)
        self._h = C.c_void_p()
        o = default_opts(verbose=verbose, **opts)
        if comm is None:
            _check(lib().oi_effdiff_create(C.byref(self._h), self._host.ctypes.data_as(C.c_void_p), nx, ny, nz,
                                           phase_id, C.byref(o)))
        else:
            self._comm = comm
            _check(lib().oi_effdiff_create_slab(C.byref(self._h), comm._h, self._host.ctypes.data_as(C.c_void_p), 0, nx,
                                                ny, int(nz_global if nz_global is not None else nz), int(z0), nz,
                                                phase_id, C.byref(o)))

    def solve(self, direction):
        """EffectiveDiffusivityHypre::solve for chi_k -> (converged, iterations, relres)."""
        conv, it, rel = C.c_int(), C.c_int(), C.c_double()
        _check(lib().oi_effdiff_solve(self._h, int(direction), C.byref(conv), C.byref(it), C.byref(rel)))
        return bool(conv.value), it.value, rel.value

    def getChiSolution(self, direction):
        out = np.empty(self.shape, np.float64)
        _check(lib().oi_effdiff_get_chi(self._h, int(direction), out.ctypes.data_as(C.c_void_p)))
        return out

    def tensor(self):
        """D_eff / D_material, 3x3 (rows: derivative axis, columns: chi direction)."""
        D = (C.c_double * 9)()
        _check(lib().oi_effdiff_tensor(self._h, D))
        return np.array(D[:]).reshape(3, 3)

    def layout(self):
        """(compacted, doubles per CG vector): compacted = True when the solves run on line-compacted vectors with the rotated
        multigrid iteration (multigrid preconditioner; one GPU or a ring of slabs that start on even planes), False on dense vectors."""
        c, n = C.c_int(), C.c_int64()
        _check(lib().oi_effdiff_layout(self._h, C.byref(c), C.byref(n)))
        return bool(c.value), n.value

    def destroy(self):
        if self._h:
            lib().oi_effdiff_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# readers (src/io)
# ------------------------------------------------------------------------------------------
class TiffReader:
    """OpenImpala::TiffReader (src/io/TiffReader.{H,cpp}): a stack file, `TiffReader(filename)`, or a
    numbered sequence of single-slice files, `TiffReader(base_pattern, num_files, start_index, digits,
    suffix)` (TiffReader.H:60-80)."""

    def __init__(self, filename, num_files=None, start_index=0, digits=1, suffix=".tif"):
        self.filename = filename
        self.sequence = None if num_files is None else (int(num_files), int(start_index), int(digits), suffix)
        self._meta = None

    def threshold(self, raw_threshold=0.5, z0=0, nz_local=None):
        """-> uint8 (nz, ny, nx): 1 where voxel > threshold else 0 (TiffReader.cpp:449-455).  z0 / nz_local
        select the planes [z0, z0 + nz_local) of a stack file (a rank's z-slab; only those pages are read)."""
        p = C.c_void_p()
        nx, ny, nz, bps = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        planes = None
        if self.sequence is None and (z0 or nz_local is not None):
            planes = -1 if nz_local is None else int(nz_local)
            _check(lib().oi_read_tiff_planes(self.filename.encode(), raw_threshold, int(z0), planes, C.byref(p), C.byref(nx),
                                             C.byref(ny), C.byref(nz), C.byref(bps)))
            planes = nz.value - int(z0) if planes < 0 else planes
        elif self.sequence is None:
            _check(lib().oi_read_tiff(self.filename.encode(), raw_threshold, C.byref(p), C.byref(nx), C.byref(ny),
                                      C.byref(nz), C.byref(bps)))
        else:
            nf, si, dg, sfx = self.sequence
            _check(lib().oi_read_tiff_sequence(self.filename.encode(), nf, si, dg, sfx.encode(), raw_threshold, C.byref(p),
                                               C.byref(nx), C.byref(ny), C.byref(nz), C.byref(bps)))
        depth = nz.value if planes is None else planes
        try:
            n = nx.value * ny.value * depth
            out = _take_bytes(p, n)
        finally:
            lib().oi_free_host(p)
        self._meta = (nx.value, ny.value, nz.value, bps.value)
        return out.reshape(depth, ny.value, nx.value)

    def _need(self):
        if self._meta is None:
            self.threshold()
        return self._meta

    def width(self):
        return self._need()[0]

    def height(self):
        return self._need()[1]

    def depth(self):
        return self._need()[2]

    def bitsPerSample(self):
        return self._need()[3]


RAW_TYPES = {"UINT8": 1, "INT8": 2, "INT16_LE": 3, "INT16_BE": 4, "UINT16_LE": 5, "UINT16_BE": 6, "INT32_LE": 7,
             "INT32_BE": 8, "UINT32_LE": 9, "UINT32_BE": 10, "FLOAT32_LE": 11, "FLOAT32_BE": 12, "FLOAT64_LE": 13,
             "FLOAT64_BE": 14}


class RawReader:
    """OpenImpala::RawReader (src/io/RawReader.{H,cpp}): dimensions and type supplied by the caller."""

    def __init__(self, filename, width, height, depth, data_type="UINT8"):
        self.filename, self.dims, self.dtype = filename, (width, height, depth), RAW_TYPES[data_type.upper()]

    def threshold(self, threshold_value=0.5, z0=0, nz_local=None):
        """z0 / nz_local: only the planes [z0, z0 + nz_local) (a rank's z-slab; only those bytes are read)."""
        p = C.c_void_p()
        nx, ny, nz = self.dims
        nzl = nz - int(z0) if nz_local is None else int(nz_local)
        _check(lib().oi_read_raw_planes(self.filename.encode(), nx, ny, nz, self.dtype, threshold_value, int(z0), nzl,
                                        C.byref(p)))
        try:
            out = _take_bytes(p, nx * ny * nzl)
        finally:
            lib().oi_free_host(p)
        return out.reshape(nzl, ny, nx)


class DatReader:
    """OpenImpala::DatReader (src/io/DatReader.{H,cpp}): int32 W,H,D header + uint16 voxels."""

    def __init__(self, filename):
        self.filename = filename

    def threshold(self, raw_threshold=0.5):
        p = C.c_void_p()
        nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
        _check(lib().oi_read_dat(self.filename.encode(), raw_threshold, C.byref(p), C.byref(nx), C.byref(ny),
                                 C.byref(nz)))
        try:
            n = nx.value * ny.value * nz.value
            out = _take_bytes(p, n)
        finally:
            lib().oi_free_host(p)
        return out.reshape(nz.value, ny.value, nx.value)


class HDF5Reader:
    """OpenImpala::HDF5Reader (src/io/HDF5Reader.{H,cpp}): rank-3 dataset, dims (Z, Y, X)."""

    def __init__(self, filename, hdf5dataset="image"):
        self.filename, self.dataset = filename, hdf5dataset

    def threshold(self, raw_threshold=0.5, z0=0, nz_local=None):
        """z0 / nz_local: only the planes [z0, z0 + nz_local) of the dataset (a rank's z-slab)."""
        p = C.c_void_p()
        nx, ny, nz = C.c_int(), C.c_int(), C.c_int()
        _check(lib().oi_read_hdf5_planes(self.filename.encode(), self.dataset.encode(), raw_threshold, int(z0),
                                         -1 if nz_local is None else int(nz_local), C.byref(p), C.byref(nx), C.byref(ny),
                                         C.byref(nz)))
        depth = nz.value - int(z0) if nz_local is None else int(nz_local)
        try:
            out = _take_bytes(p, nx.value * ny.value * depth)
        finally:
            lib().oi_free_host(p)
        return out.reshape(depth, ny.value, nx.value)
