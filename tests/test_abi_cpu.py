"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/openimpala_amd.h declares, and refuses to compute without a GPU (no fallback)."""
import os
import re

import numpy as np
import pytest

import openimpala_amd as oia

from conftest import ROOT, GOLDEN


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "openimpala_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oi_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert _declared_symbols() == sorted(oia.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = oia.lib()
    for name in _declared_symbols():
        assert hasattr(L, name), name
    assert L.oi_abi_version() == oia.ABI_VERSION


def test_the_product_library_carries_no_cross_check_kernels():
    """`make` builds the product, `make xcheck` the test library: same exports, and only the latter holds the superseded kernels
    (k_cg1 in any instantiation); cross-check options are refused by the product without touching a device."""
    import subprocess
    assert oia.lib().oi_build_has_xcheck() == 0
    with pytest.raises(oia.OpenImpalaError) as e:
        oia.set_option("cg1_kernel", 1)
    assert e.value.status == oia.OI_BAD_ARG
    libs = {n: os.path.join(ROOT, "openimpala_amd", n) for n in ("libopenimpala_amd.so", "libopenimpala_amd_xcheck.so")}
    if not os.path.exists(libs["libopenimpala_amd_xcheck.so"]):
        pytest.skip("test library not built (make -C openimpala_amd/csrc xcheck)")
    syms = {n: subprocess.run(["nm", "-D", "--defined-only", p], capture_output=True, text=True, check=True).stdout for n, p in libs.items()}
    exported = {n: {l.split()[-1] for l in s.splitlines() if " T " in l and l.split()[-1].startswith("oi_")} for n, s in syms.items()}
    assert exported["libopenimpala_amd.so"] == exported["libopenimpala_amd_xcheck.so"]
    has = {n: (b"k_cg1I" in open(p, "rb").read()) for n, p in libs.items()}   # (mangled oi::k_cg1<...>; k_cg1b mangles as 6k_cg1b)
    assert has == {"libopenimpala_amd.so": False, "libopenimpala_amd_xcheck.so": True}, has


def test_struct_layouts_match_header():
    # sizes as the C compiler lays the structs out (natural alignment, no packing)
    assert oia.Opts.__dict__ is not None
    import ctypes as C
    assert C.sizeof(oia.Opts) == 48
    assert C.sizeof(oia.Stats) == 8 + 7 * 8 + 3 * 8 + 8 + 5 * 8 + 8


@pytest.mark.skipif(oia.lib().oi_device_count() > 0, reason="GPU present")
def test_no_gpu_means_loud_failure_not_fallback():
    ph = np.ones((4, 4, 4), np.uint8)
    with pytest.raises(oia.OpenImpalaError) as e:
        oia.TortuosityMI355(ph, 1, oia.Direction.X)
    assert e.value.status == oia.OI_NO_DEVICE
    with pytest.raises(oia.OpenImpalaError):
        oia.VolumeFraction(ph, 1).value()


def test_tiff_reader_matches_oracle_reader():
    """The readers are host code: TiffReader dims/bps as tTiffReader.cpp:98-103 asserts,
    voxels identical to the oracle's independent numpy decoder."""
    from oracle import oracle as o
    tif = os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif")
    r = oia.TiffReader(tif)
    ph = r.threshold(0.5)
    assert (r.width(), r.height(), r.depth(), r.bitsPerSample()) == (100, 100, 100, 1)
    assert np.array_equal(ph, o.read_tiff_phase(tif))
    assert np.array_equal(oia.TiffReader(tif).threshold(1.5), np.zeros_like(ph))


def test_raw_reader_types(tmp_path):
    rng = np.random.default_rng(0)
    vals = rng.integers(0, 60000, size=(3, 4, 5))
    for name, dt in (("UINT16_LE", "<u2"), ("UINT16_BE", ">u2"), ("INT32_BE", ">i4"), ("FLOAT32_LE", "<f4"),
                     ("FLOAT64_BE", ">f8"), ("UINT8", "u1")):
        a = (vals % 200 if name == "UINT8" else vals).astype(dt)
        p = tmp_path / f"{name}.raw"
        a.tofile(p)
        got = oia.RawReader(str(p), 5, 4, 3, name).threshold(100.0)
        assert np.array_equal(got, (a.astype(np.float64) > 100.0).astype(np.uint8)), name
    with pytest.raises(oia.OpenImpalaError):
        oia.RawReader(str(p), 5, 4, 4, "UINT8").threshold(1.0)  # size mismatch


def test_raw_reader_thresholds_equal_the_double_compare_for_every_type(tmp_path):
    """The readers compare integer samples of up to 32 bits in their own type (v > floor(t), vectorised, plane by plane on several
    threads): the result must be what the reference's `(double)value > threshold` gives for EVERY sample type, byte order and
    threshold -- fractional, negative, at and beyond the type's range, infinite, NaN -- including the type's extreme values."""
    rng = np.random.default_rng(5)
    shape = (9, 7, 33)   # (planes, rows, columns): more planes than one thread takes
    types = {"UINT8": "u1", "INT8": "i1", "INT16_LE": "<i2", "INT16_BE": ">i2", "UINT16_LE": "<u2", "UINT16_BE": ">u2", "INT32_LE": "<i4",
             "INT32_BE": ">i4", "UINT32_LE": "<u4", "UINT32_BE": ">u4", "FLOAT32_LE": "<f4", "FLOAT32_BE": ">f4", "FLOAT64_LE": "<f8",
             "FLOAT64_BE": ">f8"}
    for name, dt in types.items():
        d = np.dtype(dt)
        if d.kind in "iu":
            info = np.iinfo(d)
            a = rng.integers(info.min, info.max, size=shape, dtype=np.int64, endpoint=True).astype(d)
            a.flat[:4] = [info.min, info.max, 0 if info.min < 0 else 1, info.max - 1]
            thresholds = [0.0, 0.5, -0.5, -1.0, 100.0, 100.999, float(info.min), float(info.min) - 1.0, float(info.min) + 0.5, float(info.max),
                          float(info.max) - 0.5, float(info.max) - 1.0, float(info.max) + 1.0, 1e30, -1e30, float("inf"), float("-inf"), float("nan")]
        else:
            a = (rng.standard_normal(shape) * 100).astype(d)
            a.flat[:3] = [0.0, -0.0, 100.0]
            thresholds = [0.0, 100.0, -3.25, float("inf"), float("-inf"), float("nan")]
        p = tmp_path / f"{name}.raw"
        a.tofile(p)
        r = oia.RawReader(str(p), shape[2], shape[1], shape[0], name)
        for t in thresholds:
            want = (a.astype(np.float64) > t).astype(np.uint8)
            assert np.array_equal(r.threshold(t), want), (name, t)
            assert np.array_equal(r.threshold(t, z0=2, nz_local=5), want[2:7]), (name, t)


def test_remspot_needs_the_device():
    """remspot runs on the GPU like every other compute step: no CPU path to fall back to."""
    with pytest.raises(oia.OpenImpalaError):
        oia.remspot(np.zeros((4, 4, 4), np.uint8), 1)


def test_slab_ranges_cover_the_volume():
    for nz, n in ((100, 8), (7, 3), (1024, 8), (5, 5)):
        spans = [oia.slab_range(nz, n, r) for r in range(n)]
        assert spans[0][0] == 0 and spans[-1][1] == nz
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert all(z1 > z0 for z0, z1 in spans)


def test_hdf5_reader_sample_and_layouts(tmp_path):
    """HDF5Reader: the bundled sample (dims 100^3, values {0,1}: tHDF5Reader.cpp:88-119) and
    hand-written files covering nested groups, chunked layout, shuffle + deflate, dtypes."""
    from h5write import write_h5
    sample = os.path.join(GOLDEN, "SampleData_2Phase_3d.hdf5")
    ph = oia.HDF5Reader(sample, "image").threshold(0.5)
    raw = np.frombuffer(open(sample, "rb").read()[2048:2048 + 10 ** 6], np.uint8).reshape(100, 100, 100)
    assert ph.shape == (100, 100, 100) and set(np.unique(ph)) == {0, 1}
    assert np.array_equal(ph, (raw > 0.5).astype(np.uint8))
    with pytest.raises(oia.OpenImpalaError):
        oia.HDF5Reader(sample, "nope").threshold(0.5)
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 4000, size=(9, 14, 21))
    cases = [("u1", None, False, False), ("<u2", (4, 5, 8), False, False), ("<u2", (4, 5, 8), True, True),
             (">i4", (3, 14, 21), True, False), ("<f4", (9, 7, 4), True, True), ("<f8", None, False, False),
             ("<i2", (2, 2, 2), False, True)]
    for n, (dt, chunks, gz, sh) in enumerate(cases):
        a = (vals % 200 if dt == "u1" else vals - (1000 if "i" in dt else 0)).astype(dt)
        p = tmp_path / f"c{n}.h5"
        names = ("image",) if n % 2 == 0 else ("vol", "scan", "image")
        write_h5(str(p), a, names=names, chunks=chunks, gzip=gz, shuffle=sh)
        got = oia.HDF5Reader(str(p), "/".join(names)).threshold(100.0)
        assert got.shape == a.shape
        assert np.array_equal(got, (a.astype(np.float64) > 100.0).astype(np.uint8)), (dt, chunks, gz, sh)
    write_h5(str(tmp_path / "r2.h5"), vals.astype("u1"))
    bad = bytearray(open(tmp_path / "r2.h5", "rb").read())
    bad[8] = 2  # pretend "latest format" superblock
    open(tmp_path / "v2.h5", "wb").write(bytes(bad))
    with pytest.raises(oia.OpenImpalaError):
        oia.HDF5Reader(str(tmp_path / "v2.h5")).threshold(1.0)


def test_tiff_reader_other_bundled_files():
    """The reference's other TIFF fixtures: a 64^3 1-bit stack and a PHOTOMETRIC = 0 ("white is
    zero") stack -- the reader works on raw bits and ignores PHOTOMETRIC like the reference
    (TiffReader.cpp:419-426); voxel fractions from SURVEY.md section 8c."""
    from oracle import oracle as o
    sq = oia.TiffReader(os.path.join(GOLDEN, "SampleData_2Phase_squared.tif"))
    ph = sq.threshold(0.5)
    assert ph.shape == (64, 64, 64) and sq.bitsPerSample() == 1
    assert ph.mean() == pytest.approx(0.399326, abs=5e-7)
    assert np.array_equal(ph, o.read_tiff_phase(os.path.join(GOLDEN, "SampleData_2Phase_squared.tif")))
    sp = oia.TiffReader(os.path.join(GOLDEN, "spheres.tif")).threshold(0.5)
    assert sp.shape == (100, 100, 100) and sp.mean() == pytest.approx(0.888024, abs=5e-7)
    assert np.array_equal(sp, o.read_tiff_phase(os.path.join(GOLDEN, "spheres.tif")))


def _pil():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, features
    if not features.check("libtiff"):
        pytest.skip("Pillow without libtiff")
    return Image


@pytest.mark.parametrize("compression", ["raw", "tiff_lzw", "tiff_adobe_deflate", "tiff_deflate", "packbits"])
@pytest.mark.parametrize("mode,dtype", [("L", np.uint8), ("I;16", np.uint16), ("F", np.float32), ("1", np.uint8)])
def test_tiff_reader_compressed_stacks_written_by_libtiff(tmp_path, compression, mode, dtype):
    """Multi-page stacks written by libtiff (through Pillow) in every encoding the reference's libtiff-based
    reader meets in practice; expected voxels = libtiff's own decode of the same file."""
    Image = _pil()
    rng = np.random.default_rng(7)
    nz, ny, nx = 5, 37, 83
    if mode == "1":
        vol = (rng.random((nz, ny, nx)) < 0.4)
        pages = [Image.fromarray(v) for v in vol]
        thr = 0.5
    else:
        smooth = np.cumsum(rng.integers(-3, 4, size=(nz, ny, nx)), axis=2) + 120   # compressible rows
        vol = np.clip(smooth, 0, 250).astype(dtype) * (257 if dtype == np.uint16 else 1)
        pages = [Image.fromarray(v) for v in vol]
        thr = float(np.median(vol))
    p = tmp_path / "stack.tif"
    pages[0].save(p, format="TIFF", compression=compression, save_all=True, append_images=pages[1:])
    expect = np.stack([_page(Image, p, k) for k in range(nz)])
    got = oia.TiffReader(str(p)).threshold(thr)
    assert got.shape == (nz, ny, nx)
    assert np.array_equal(got, (expect.astype(np.float64) > thr).astype(np.uint8))
    assert 0 < got.mean() < 1


def _page(Image, path, k):
    with Image.open(path) as im:
        im.seek(k)
        return np.array(im)


@pytest.mark.parametrize("big", [False, True])
def test_tiff_reader_predictor_bigtiff_and_sequences(tmp_path, big):
    Image = _pil()
    rng = np.random.default_rng(11)
    nz, ny, nx = 4, 29, 50
    vol = (np.cumsum(rng.integers(-40, 41, size=(nz, ny, nx)), axis=2) + 30000).astype(np.uint16)
    thr = float(np.median(vol))
    pages = [Image.fromarray(v) for v in vol]
    kw = dict(format="TIFF", compression="tiff_lzw", tiffinfo={317: 2})   # horizontal differencing predictor
    if big:   # Pillow's libtiff path ignores big_tiff; its own (uncompressed) writer honours it
        kw = dict(format="TIFF", big_tiff=True)
    p = tmp_path / "pred.tif"
    pages[0].save(p, save_all=True, append_images=pages[1:], **kw)
    assert open(p, "rb").read(4)[2] == (43 if big else 42)
    got = oia.TiffReader(str(p)).threshold(thr)
    assert np.array_equal(got, (vol > thr).astype(np.uint8))
    # the same slices as a numbered file sequence (TiffReader.H:117-123)
    for k, im in enumerate(pages):
        im.save(tmp_path / f"slice_{k + 3:04d}.tif", **kw)
    seq = oia.TiffReader(str(tmp_path / "slice_"), num_files=nz, start_index=3, digits=4, suffix=".tif")
    assert np.array_equal(seq.threshold(thr), got)
    assert (seq.width(), seq.height(), seq.depth(), seq.bitsPerSample()) == (nx, ny, nz, 16)
    with pytest.raises(oia.OpenImpalaError):   # a missing member of the sequence is an error, not a short volume
        oia.TiffReader(str(tmp_path / "slice_"), num_files=nz + 1, start_index=3, digits=4).threshold(thr)


@pytest.mark.parametrize("compression", ["tiff_lzw", "tiff_adobe_deflate"])
def test_tiff_reader_floating_point_predictor(tmp_path, compression):
    """Predictor 3 (TIFF Technical Note 3: byte planes of the IEEE samples, differenced as bytes) on float32 stacks written
    by libtiff; expected voxels = libtiff's own decode of the same file."""
    Image = _pil()
    rng = np.random.default_rng(5)
    nz, ny, nx = 3, 31, 67
    vol = (np.cumsum(rng.normal(0.0, 0.3, size=(nz, ny, nx)), axis=2) + 5.0).astype(np.float32)
    pages = [Image.fromarray(v) for v in vol]
    p = tmp_path / "fpred.tif"
    pages[0].save(p, format="TIFF", compression=compression, tiffinfo={317: 3}, save_all=True, append_images=pages[1:])
    with Image.open(p) as im:
        if im.tag_v2.get(317) != 3:
            pytest.skip("this libtiff build did not apply the floating-point predictor")
    expect = np.stack([_page(Image, p, k) for k in range(nz)])
    assert np.array_equal(expect, vol)   # libtiff round-trips the samples
    thr = float(np.median(vol))
    got = oia.TiffReader(str(p)).threshold(thr)
    assert np.array_equal(got, (vol.astype(np.float64) > thr).astype(np.uint8)) and 0 < got.mean() < 1


def _jpeg_volume(seed, shape):
    rng = np.random.default_rng(seed)
    v = np.cumsum(rng.integers(-9, 10, size=shape), axis=2) + 128 + np.cumsum(rng.integers(-5, 6, size=shape), axis=1)
    return np.clip(v, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("quality,rps", [(90, None), (60, 8), (100, 24)])
def test_tiff_reader_jpeg_compressed_stacks(tmp_path, quality, rps):
    """Compression 7 (JPEG-in-TIFF with a JPEGTables tag) on 8-bit stacks written by libtiff + libjpeg: one or several strips per
    page, extents that are no multiples of 8.  Expected voxels = libtiff's own decode of the same file; the decoder here
    (sequential Huffman, libjpeg's integer inverse DCT) reproduces it EXACTLY, so every threshold agrees."""
    Image = _pil()
    vol = _jpeg_volume(5, (3, 83, 131))
    pages = [Image.fromarray(v) for v in vol]
    p = tmp_path / "jpeg.tif"
    kw = dict(format="TIFF", compression="jpeg", quality=quality, save_all=True, append_images=pages[1:])
    if rps:
        kw["tiffinfo"] = {278: rps}
    try:
        pages[0].save(p, **kw)
    except Exception:
        pytest.skip("libtiff without JPEG")
    with Image.open(p) as im:
        assert im.tag_v2.get(259) == 7
    expect = np.stack([_page(Image, p, k) for k in range(vol.shape[0])])
    assert np.abs(expect.astype(int) - vol.astype(int)).max() < 80     # lossy, but the same picture
    for thr in (60.5, 100.5, 127.5, 128.5, 170.5):
        got = oia.TiffReader(str(p)).threshold(thr)
        assert np.array_equal(got, (expect > thr).astype(np.uint8)), thr
    assert np.array_equal(oia.TiffReader(str(p)).threshold(128.5, z0=1, nz_local=2), (expect[1:] > 128.5).astype(np.uint8))


def _wrap_jpeg_tiff(jpeg_bytes, w, h):
    """classic little-endian TIFF, one strip that is a self-contained JPEG stream (no JPEGTables tag)"""
    import struct
    tags = [(256, 3, 1, w), (257, 3, 1, h), (258, 3, 1, 8), (259, 3, 1, 7), (262, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1),
            (278, 3, 1, h), (279, 4, 1, len(jpeg_bytes))]
    ifd_off = 8 + len(jpeg_bytes) + (len(jpeg_bytes) & 1)
    out = b"II" + struct.pack("<HI", 42, ifd_off) + jpeg_bytes + (b"\0" if len(jpeg_bytes) & 1 else b"")
    out += struct.pack("<H", len(tags))
    for tag, typ, cnt, val in tags:
        out += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    return out + struct.pack("<I", 0)


def test_tiff_reader_jpeg_restart_intervals_progressive_and_refusals(tmp_path):
    """JPEG strips written by libjpeg through Pillow's JPEG writer and wrapped into a TIFF by hand (a self-contained stream, no JPEGTables
    tag): with restart markers (DRI), and PROGRESSIVE (SOF2: spectral selection + successive approximation, several scans with their own
    Huffman tables; round 5) -- each decodes to libjpeg's own pixels, bit for bit; libtiff (what the reference reads with) opens the
    same files.  Colour images are refused like every multi-sample TIFF."""
    Image = _pil()
    import io
    cases = [("restart", (64, 96), dict(quality=85, restart_marker_blocks=5)), ("plain", (64, 96), dict(quality=75)),
             ("progressive", (64, 96), dict(quality=85, progressive=True)),
             ("progressive_ragged", (83, 131), dict(quality=60, progressive=True)),            # extents that are no multiples of 8
             ("progressive_q100", (40, 57), dict(quality=100, progressive=True)),              # every coefficient survives: long refinement scans
             ("progressive_flat", (33, 20), dict(quality=30, progressive=True)),               # long end-of-band runs
             ("progressive_restart", (72, 104), dict(quality=90, progressive=True, restart_marker_blocks=7)),
             ("progressive_optimised", (64, 96), dict(quality=80, progressive=True, optimize=True))]
    decoded_progressive = 0
    for name, shape, kw in cases:
        a = _jpeg_volume(len(name), (1,) + shape)[0]
        if name == "progressive_flat":
            a = (a // 64 * 64).astype(np.uint8)
        buf = io.BytesIO()
        try:
            Image.fromarray(a).save(buf, format="JPEG", **kw)
        except Exception:
            pytest.skip("Pillow without these JPEG options")
        data = buf.getvalue()
        if name == "restart" and b"\xff\xdd" not in data:
            pytest.skip("this Pillow does not write restart intervals")
        if name.startswith("progressive"):
            assert b"\xff\xc2" in data and data.count(b"\xff\xda") > 3       # SOF2 and several scans
            if name == "progressive_restart" and b"\xff\xdd" not in data:
                continue
        p = tmp_path / f"{name}.tif"
        p.write_bytes(_wrap_jpeg_tiff(data, a.shape[1], a.shape[0]))
        expect = np.array(Image.open(io.BytesIO(data)))                          # libjpeg's decode of the stream
        assert expect.shape == a.shape and np.abs(expect.astype(int) - a.astype(int)).max() < 120
        try:                                                                     # and libtiff's decode of the wrapped file
            with Image.open(p) as im:
                via_libtiff = np.array(im)
            assert np.array_equal(via_libtiff, expect), name
        except (OSError, ValueError):
            pass                                                                 # (a Pillow whose libtiff lacks JPEG: the stream's decode stands)
        levels = np.unique(expect)
        for thr in list(levels[:: max(1, len(levels) // 12)].astype(float) + 0.5) + [100.5, 128.5, 150.5]:
            got = oia.TiffReader(str(p)).threshold(float(thr))[0]
            assert np.array_equal(got, (expect > thr).astype(np.uint8)), (name, thr)
        decoded_progressive += name.startswith("progressive")
    assert decoded_progressive >= 4
    rgb = tmp_path / "rgb.tif"
    a = _jpeg_volume(9, (1, 64, 96))[0]
    Image.fromarray(np.stack([a] * 3, axis=-1)).save(rgb, format="TIFF", compression="jpeg")
    with pytest.raises(oia.OpenImpalaError):   # three samples per pixel: refused like every multi-sample TIFF (TiffReader.cpp:167-173)
        oia.TiffReader(str(rgb)).threshold(1.0)
    # a progressive stream cut short or with a scan removed is an error, not a picture
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="JPEG", quality=85, progressive=True)
    data = buf.getvalue()
    cut = tmp_path / "cut.tif"
    cut.write_bytes(_wrap_jpeg_tiff(data[: len(data) * 2 // 3], a.shape[1], a.shape[0]))
    with pytest.raises(oia.OpenImpalaError):
        oia.TiffReader(str(cut)).threshold(1.0)


def test_readers_load_a_z_slab_only(tmp_path):
    """oi_read_tiff_planes / oi_read_raw_planes: a rank reads its planes, not the volume."""
    tif = os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif")
    full = oia.TiffReader(tif).threshold(0.5)
    r = oia.TiffReader(tif)
    assert np.array_equal(r.threshold(0.5, z0=37, nz_local=21), full[37:58])
    assert r.depth() == 100                      # the file's depth, not the slab's
    assert np.array_equal(oia.TiffReader(tif).threshold(0.5, z0=90), full[90:])
    with pytest.raises(oia.OpenImpalaError):
        oia.TiffReader(tif).threshold(0.5, z0=95, nz_local=6)
    vals = np.random.default_rng(4).integers(0, 60000, size=(9, 6, 7)).astype("<u2")
    p = tmp_path / "v.raw"
    vals.tofile(p)
    rr = oia.RawReader(str(p), 7, 6, 9, "UINT16_LE")
    assert np.array_equal(rr.threshold(30000.0, z0=2, nz_local=5), (vals[2:7] > 30000).astype(np.uint8))
    assert np.array_equal(rr.threshold(30000.0), (vals > 30000).astype(np.uint8))
    with pytest.raises(oia.OpenImpalaError):
        rr.threshold(1.0, z0=8, nz_local=2)
    # HDF5: contiguous and chunked + filtered layouts (the chunk index is walked, only chunks that meet the planes are decoded)
    from h5write import write_h5
    hv = np.random.default_rng(8).integers(0, 4000, size=(11, 10, 13)).astype("<u2")
    for n_, (chunks, gz, sh) in enumerate(((None, False, False), ((4, 5, 8), True, True), ((3, 10, 13), False, False))):
        hp = tmp_path / f"s{n_}.h5"
        write_h5(str(hp), hv, names=("image",), chunks=chunks, gzip=gz, shuffle=sh)
        hr = oia.HDF5Reader(str(hp), "image")
        assert np.array_equal(hr.threshold(2000.0, z0=3, nz_local=6), (hv[3:9] > 2000).astype(np.uint8)), chunks
        assert np.array_equal(hr.threshold(2000.0, z0=9), (hv[9:] > 2000).astype(np.uint8))
        assert np.array_equal(hr.threshold(2000.0), (hv > 2000).astype(np.uint8))
        with pytest.raises(oia.OpenImpalaError):
            hr.threshold(1.0, z0=10, nz_local=2)


@pytest.mark.parametrize("descr", [1, 0])
def test_launch_geometry_never_exceeds_the_partial_sum_array(descr):
    """Host arithmetic of the single-kernel iteration's launch shape (no device): for wide planes, long thin planes, short slabs
    and the BASELINE shares, every launch -- whole slab, boundary chunks, interior chunks -- stays within the capacity of the
    partial-sum array a handle of that slab allocates (round 1: an 8192 x 4096 cross-section wrote past it)."""
    shapes = [(1024, 1024, 1024), (8192, 4096, 64), (4096, 4096, 512), (2048, 2048, 256), (16384, 100, 300), (100, 16384, 9),
              (100, 100, 100), (130, 7, 3), (1, 1, 1), (5000, 5000, 8), (2, 70000, 40)]
    for nx, ny, nzl in shapes:
        for split, parts in ((0, (0,)), (1, (1, 2))):
            total = 0
            for part in parts:
                rows, zchunk, blocks, cap = oia.cg1_grid(nx, ny, nzl, descr=descr, split=split, part=part)
                assert rows in (8, 16, 32) and 1 <= zchunk <= max(nzl, 1) and cap >= 4096
                assert 0 <= blocks <= cap, (nx, ny, nzl, split, part, blocks, cap)
                total += blocks
            assert total <= cap and total >= 1
    # the default of the descriptor-addressed kernel at the bench size: 16-row tiles, 24-plane chunks (16 until the nontemporal
    # policy of round 4, profiles/r04/k_cg1b_chunk_length_tile_map_blobs_with_nontemporal.txt, slab_chunk_lengths.txt)
    assert oia.cg1_grid(1024, 1024, 1024, descr=1)[:3] == (16, 24, 8 * 64 * 43)
    assert oia.cg1_grid(1024, 1024, 1024, descr=0)[0] == 32


def test_descriptor_kernels_carry_no_scalar_offset_register():
    """profiles/r03/k_cg1b_soffset_hazard_confirmed.txt: on gfx950 a buffer instruction may pick its SCALAR offset up after a
    following SALU instruction has rewritten the register (variant builds with and without wait states, three GPU runs each);
    the descriptor-addressed kernels therefore pass 0 there and carry rows in the lane offset.  This disassembles the code
    object that ships and fails if the compiler has put a scalar-offset register on any buffer instruction of k_cg1b."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    if not os.path.exists(check_isa.OBJDUMP):
        pytest.skip("no llvm-objdump")
    lib = os.path.join(ROOT, "openimpala_amd", "libopenimpala_amd.so")
    found, counts = check_isa.scan(check_isa.device_disassembly(lib))
    assert len(counts) >= 16 and sum(counts.values()) > 1000, counts     # the kernels were found and do use buffer instructions
    assert not found, {k: v[:2] for k, v in found.items()}
