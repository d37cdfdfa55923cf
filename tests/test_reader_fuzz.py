"""Corrupted and truncated input files: the readers report OI_IO_ERROR (or read what is still consistent), they never
crash.  Each batch runs in a child process so that a fault would fail the test instead of killing the session."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import h5write
import openimpala_amd as oia
from conftest import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, sys.argv[1])
import openimpala_amd as oia
kind, dataset = sys.argv[2], sys.argv[3]
for path in sys.argv[4:]:
    try:
        r = (oia.TiffReader(path) if kind == "tiff" else oia.HDF5Reader(path, dataset)).threshold(0.5)
        print("ok", r.shape)
    except oia.OpenImpalaError as e:
        assert e.status in (oia.OI_IO_ERROR, oia.OI_BAD_ARG), e.status
        print("err")
print("batch done")
'''


def corrupt(src, t, rng):
    b = bytearray(src)
    mode = t % 3
    if mode == 0:
        return b[:rng.randrange(8, len(b))]
    if mode == 1:
        for _ in range(rng.randrange(1, 16)):
            b[rng.randrange(len(b))] = rng.randrange(256)
        return b
    for _ in range(10):  # header / directory regions
        pos = rng.randrange(0, min(len(b), 4096)) if rng.random() < 0.5 else rng.randrange(max(0, len(b) - 8192), len(b))
        b[pos] = rng.randrange(256)
    return b


def run_batch(tmp_path, kind, dataset, sources, n):
    rng = random.Random(20251031)
    paths = []
    for t in range(n):
        p = tmp_path / f"{kind}{t}"
        p.write_bytes(bytes(corrupt(sources[t % len(sources)], t, rng)))
        paths.append(str(p))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, kind, dataset] + paths, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("batch done"), (r.returncode, r.stderr[-500:])
    assert len(r.stdout.split()) > n  # one verdict per file


def test_corrupted_tiffs_are_errors(tmp_path):
    src = open(os.path.join(GOLDEN, "SampleData_2Phase_stack_3d_1bit.tif"), "rb").read()
    run_batch(tmp_path, "tiff", "-", [src], 45)


def test_corrupted_hdf5_files_are_errors(tmp_path):
    arr = (np.random.default_rng(1).random((12, 20, 28)) * 255).astype(np.uint8)
    a, b = tmp_path / "a.h5", tmp_path / "b.h5"
    h5write.write_h5(str(a), arr)
    h5write.write_h5(str(b), arr, chunks=(4, 10, 14), gzip=True, shuffle=True)
    run_batch(tmp_path, "hdf5", "image", [a.read_bytes(), b.read_bytes()], 45)


def test_crafted_hdf5_chunk_indices_are_refused(tmp_path):
    """chunks are decoded on several threads, each writing 'its own box' of the output -- an index that lists one box twice
    (or a chunk off the chunk grid, overlapping its neighbour) would make two threads write the same bytes.  Both are refused with
    an I/O error; the clean file of the same data still reads."""
    arr = (np.random.default_rng(4).random((12, 20, 28)) * 60000).astype(np.uint16)
    ok = tmp_path / "ok.h5"
    h5write.write_h5(str(ok), arr, chunks=(4, 10, 14), gzip=True, shuffle=True)
    assert np.array_equal(oia.HDF5Reader(str(ok), "image").threshold(30000.0), (arr > 30000).astype(np.uint8))
    for craft, msg in (("duplicate", "listed twice"), ("misaligned", "off the chunk grid")):
        bad = tmp_path / f"{craft}.h5"
        h5write.write_h5(str(bad), arr, chunks=(4, 10, 14), gzip=True, shuffle=True, craft=craft)
        with pytest.raises(oia.OpenImpalaError) as e:
            oia.HDF5Reader(str(bad), "image").threshold(30000.0)
        assert e.value.status == oia.OI_IO_ERROR and msg in str(e.value), str(e.value)


def _classic_tiff(width, height, strip_offset, strip_bytes, bps=1, pages=1):
    """Minimal little-endian TIFF whose directory CLAIMS the given geometry (values are not made consistent on purpose)."""
    import struct
    out = bytearray(b"II*\x00" + struct.pack("<I", 8))
    for k in range(pages):
        tags = [(256, 4, 1, width), (257, 4, 1, height), (258, 3, 1, bps), (259, 3, 1, 1), (262, 3, 1, 1),
                (273, 4, 1, strip_offset), (277, 3, 1, 1), (278, 4, 1, height), (279, 4, 1, strip_bytes)]
        ifd = struct.pack("<H", len(tags)) + b"".join(struct.pack("<HHII", *t) for t in tags)
        nxt = len(out) + len(ifd) + 4 if k + 1 < pages else 0
        out += ifd + struct.pack("<I", nxt)
    return bytes(out) + b"\x00" * 64


def _bigtiff(width, height, strip_offset, strip_bytes):
    import struct
    out = bytearray(b"II+\x00" + struct.pack("<HHQ", 8, 0, 16))
    tags = [(256, 16, 1, width), (257, 16, 1, height), (258, 3, 1, 8), (259, 3, 1, 1), (262, 3, 1, 1),
            (273, 16, 1, strip_offset), (277, 3, 1, 1), (278, 16, 1, height), (279, 16, 1, strip_bytes)]
    out += struct.pack("<Q", len(tags)) + b"".join(struct.pack("<HHQQ", *t) for t in tags) + struct.pack("<Q", 0)
    return bytes(out) + b"\x00" * 64


def test_crafted_tiff_headers_are_refused(tmp_path):
    """64-bit values taken from the file must not wrap: W = H = 2^31 over 4 pages multiplies to 0 bytes, a LONG8 strip
    offset near 2^64 plus its byte count wraps below the file size.  libtiff validates these fields for the reference."""
    cases = {
        "huge_dims.tif": _classic_tiff(2 ** 31, 2 ** 31, 8, 16, pages=4),
        "huge_width.tif": _classic_tiff(2 ** 32 - 1, 3, 8, 16),
        "zero_height.tif": _classic_tiff(16, 0, 8, 16),
        "strip_past_end.tif": _classic_tiff(16, 16, 2 ** 32 - 8, 32, bps=8),
        "big_wrap.tif": _bigtiff(16, 16, 2 ** 64 - 8, 256),
        "big_dims.tif": _bigtiff(2 ** 40, 2 ** 40, 16, 16),
    }
    paths = []
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        paths.append(str(tmp_path / name))
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, "tiff", "-"] + paths, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("batch done"), (r.returncode, r.stderr[-500:])
    assert r.stdout.split()[:len(cases)] == ["err"] * len(cases), r.stdout
