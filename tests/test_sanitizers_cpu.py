"""AddressSanitizer + UBSan builds (CPU only; GPU sanitizers are not available on this pool): the parsers of untrusted input
(oi_io.cpp, oi_hdf5.cpp) and the oracle, driven by small test-only programs (tests/asan/*).  Any sanitizer report makes
the driver exit non-zero."""
import os
import random
import shutil
import subprocess

import numpy as np
import pytest

import h5write
from conftest import GOLDEN
from test_reader_fuzz import _bigtiff, _classic_tiff, corrupt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


@pytest.fixture(scope="module")
def drivers(tmp_path_factory):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("asan")
    csrc = os.path.join(ROOT, "openimpala_amd", "csrc")
    reader = str(d / "reader_driver")
    subprocess.run(["g++", "-std=c++17"] + SAN + [os.path.join(ROOT, "tests", "asan", "reader_driver.cpp"),
                   os.path.join(csrc, "oi_io.cpp"), os.path.join(csrc, "oi_hdf5.cpp"), "-o", reader, "-lz", "-lpthread"], check=True)
    oracle = str(d / "oracle_driver")
    subprocess.run(["gcc", "-std=c11", "-ffp-contract=off"] + SAN + [os.path.join(ROOT, "tests", "asan", "oracle_driver.c"),
                   os.path.join(ROOT, "oracle", "oi_oracle.c"), "-o", oracle, "-lm"], check=True)
    return reader, oracle


def run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stderr[-3000:])
    return r.stdout


def test_bundled_files_under_sanitizers(drivers):
    reader, _ = drivers
    out = run([reader, "tiff", "-"] + [os.path.join(GOLDEN, f) for f in ("SampleData_2Phase_stack_3d_1bit.tif", "SampleData_2Phase_squared.tif", "spheres.tif")])
    assert out.splitlines()[0].startswith("ok 100 100 100") and out.strip().endswith("batch done")
    out = run([reader, "hdf5", "image", os.path.join(GOLDEN, "SampleData_2Phase_3d.hdf5")])
    assert out.startswith("ok 100 100 100")
    out = run([reader, "raw", "100,100,100,1", os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_uint8.raw")])
    assert out.startswith("ok 100 100 100")


def test_parallel_readers_under_thread_sanitizer(tmp_path):
    """Round 4: pages, planes and chunks are decoded and thresholded on several threads (oi_hostpar.h).  The same driver built with
    ThreadSanitizer: no data race on a multi-page TIFF, a chunked + deflated + shuffled HDF5 dataset (every chunk writes its own box
    of the output), a RAW and a DAT volume with more planes than threads."""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    csrc = os.path.join(ROOT, "openimpala_amd", "csrc")
    reader = str(tmp_path / "reader_tsan")
    r = subprocess.run(["g++", "-std=c++17", "-fsanitize=thread", "-g", "-O1", os.path.join(ROOT, "tests", "asan", "reader_driver.cpp"),
                        os.path.join(csrc, "oi_io.cpp"), os.path.join(csrc, "oi_hdf5.cpp"), "-o", reader, "-lz", "-lpthread"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("ThreadSanitizer runtime not available: " + r.stderr[-200:])
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 4000, size=(40, 48, 56)).astype(np.uint16)
    h5 = tmp_path / "chunked.h5"
    h5write.write_h5(str(h5), vol, chunks=(8, 16, 16), gzip=True, shuffle=True)
    raw = tmp_path / "v.raw"
    vol.tofile(raw)
    dat = tmp_path / "v.dat"
    with open(dat, "wb") as f:
        f.write(np.array([56, 48, 40], dtype="<i4").tobytes())
        f.write(vol.astype("<u2").tobytes())
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")

    def run_tsan(cmd):
        rr = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if "FATAL: ThreadSanitizer" in rr.stderr and "data race" not in rr.stderr:
            pytest.skip("ThreadSanitizer cannot run here: " + rr.stderr[-200:])   # (e.g. an address-space layout it does not support)
        assert rr.returncode == 0 and "ThreadSanitizer" not in rr.stderr, (rr.returncode, rr.stderr[-3000:])
        return rr.stdout

    assert run_tsan([reader, "tiff", "-", os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif")]).startswith("ok 100 100 100")
    assert run_tsan([reader, "hdf5", "image", str(h5), os.path.join(GOLDEN, "SampleData_2Phase_3d.hdf5")]).startswith("ok 56 48 40")
    # a crafted index that lists one chunk box twice: refused before any thread decodes (no race on the shared box)
    dup = tmp_path / "dup.h5"
    h5write.write_h5(str(dup), vol, chunks=(8, 16, 16), gzip=True, shuffle=True, craft="duplicate")
    assert run_tsan([reader, "hdf5", "image", str(dup)]).startswith("err")
    assert run_tsan([reader, "raw", "56,48,40,5", str(raw)]).startswith("ok 56 48 40")
    assert run_tsan([reader, "dat", "-", str(dat)]).startswith("ok 56 48 40")


def test_corrupted_and_crafted_files_under_sanitizers(drivers, tmp_path):
    reader, _ = drivers
    rng = random.Random(20251031)
    src = open(os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif"), "rb").read()
    paths = []
    for t in range(60):
        p = tmp_path / f"t{t}.tif"
        p.write_bytes(bytes(corrupt(src, t, rng)))
        paths.append(str(p))
    crafted = {"a.tif": _classic_tiff(2 ** 31, 2 ** 31, 8, 16, pages=4), "b.tif": _classic_tiff(2 ** 32 - 1, 3, 8, 16),
               "c.tif": _classic_tiff(16, 16, 2 ** 32 - 8, 32, bps=8), "d.tif": _bigtiff(16, 16, 2 ** 64 - 8, 256),
               "e.tif": _bigtiff(2 ** 40, 2 ** 40, 16, 16), "f.tif": _classic_tiff(16, 0, 8, 16)}
    for name, data in crafted.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    # JPEG-compressed strips (round 3: the Huffman / inverse-DCT decoder): corrupted and truncated copies of a libtiff-written stack
    try:
        from PIL import Image, features
        if features.check("libtiff"):
            jv = np.clip(np.cumsum(np.random.default_rng(2).integers(-9, 10, size=(2, 40, 70)), axis=2) + 128, 0, 255).astype(np.uint8)
            jp = tmp_path / "jpeg_src.tif"
            Image.fromarray(jv[0]).save(jp, format="TIFF", compression="jpeg", quality=80, save_all=True, append_images=[Image.fromarray(jv[1])],
                                        tiffinfo={278: 16})
            jsrc = jp.read_bytes()
            paths.append(str(jp))
            for t in range(90):
                q = tmp_path / f"j{t}.tif"
                q.write_bytes(bytes(corrupt(jsrc, t, rng)))
                paths.append(str(q))
            # progressive streams (round 5: spectral selection / successive approximation, end-of-band runs): a hand-wrapped strip, corrupted
            import io
            from test_abi_cpu import _wrap_jpeg_tiff
            pb = io.BytesIO()
            Image.fromarray(jv[0]).save(pb, format="JPEG", quality=85, progressive=True)
            psrc = _wrap_jpeg_tiff(pb.getvalue(), jv.shape[2], jv.shape[1])
            (tmp_path / "prog_src.tif").write_bytes(psrc)
            paths.append(str(tmp_path / "prog_src.tif"))
            for t in range(160):
                q = tmp_path / f"p{t}.tif"
                q.write_bytes(bytes(corrupt(psrc, t, rng)))
                paths.append(str(q))
    except Exception:
        pass
    out = run([reader, "tiff", "-"] + paths)
    assert out.strip().endswith("batch done") and len(out.splitlines()) == len(paths) + 1
    arr = (np.random.default_rng(1).random((12, 20, 28)) * 255).astype(np.uint8)
    a, b = tmp_path / "a.h5", tmp_path / "b.h5"
    h5write.write_h5(str(a), arr)
    h5write.write_h5(str(b), arr, chunks=(4, 10, 14), gzip=True, shuffle=True)
    srcs = [a.read_bytes(), b.read_bytes()]
    hp = []
    for t in range(60):
        p = tmp_path / f"h{t}.h5"
        p.write_bytes(bytes(corrupt(srcs[t % 2], t, rng)))
        hp.append(str(p))
    out = run([reader, "hdf5", "image", str(a), str(b)] + hp)
    assert out.splitlines()[0].startswith("ok 28 20 12") and out.strip().endswith("batch done")
    # the "latest format" structures (v2 object headers, fixed / extensible arrays), written by the real libhdf5
    inc, lib = "/opt/conda/include", "/opt/conda/lib"
    if os.path.exists(os.path.join(inc, "hdf5.h")) and os.path.exists(os.path.join(lib, "libhdf5.so")):
        w = str(tmp_path / "h5w")
        if subprocess.run(["gcc", "-O1", os.path.join(ROOT, "tests", "h5_libhdf5_writer.c"), "-I" + inc, "-L" + lib, "-lhdf5",
                           "-Wl,-rpath," + lib, "-o", w], capture_output=True).returncode == 0:
            srcs, good = [], []
            # (round 3: + the version-2 B-tree chunk index, dense groups: fractal heap, name index)
            for k, v in enumerate(("chunk_gz_shuf", "paged", "unlimz_tiny", "unlimx", "single_gz", "implicit", "nested_chunk", "unlim2tiny_gz",
                                   "unlim2", "dense_chunk", "densebig_chunk_gz")):
                f = tmp_path / f"latest{k}.h5"
                subprocess.run([w, str(f), v, "40", "40", "40", "u8"], check=True, timeout=120)
                srcs.append(f.read_bytes())
                if "nested" not in v:
                    good.append(str(f))
            lp = []
            for t in range(220):
                q = tmp_path / f"l{t}.h5"
                q.write_bytes(bytes(corrupt(srcs[t % len(srcs)], t, rng)))
                lp.append(str(q))
            # files cut INSIDE the dataset's object header, message by message: the dataspace, datatype, layout and filter
            # messages end at the end of the file (fixed-offset reads of a message body must stay inside the map)
            for k, src in enumerate(srcs[:4]):
                at = src.find(b"OHDR", src.find(b"OHDR") + 4)   # (the root group's header comes first)
                for cut in range(at + 6, min(len(src), at + 200), 7):
                    q = tmp_path / f"cut{k}_{cut}.h5"
                    q.write_bytes(src[:cut])
                    lp.append(str(q))
            out = run([reader, "hdf5", "image"] + good + lp)
            assert out.splitlines()[0].startswith("ok 40 40 40") and out.strip().endswith("batch done")


def test_oracle_under_sanitizers(drivers):
    _, oracle = drivers
    out = run([oracle])
    assert out.strip().endswith("oracle done")
