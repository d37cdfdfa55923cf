"""The built-in HDF5 reader against files written by the REAL libhdf5 (1.10.6 in /opt/conda; the reference links
libhdf5 1.12, HDF5Reader.cpp:111-187, and reads whatever it reads): the default ("earliest") format and the "latest"
format -- superblock v3, version-2 object headers, compact link messages, layout message v4 with single-chunk,
implicit, fixed-array (plain, filtered, paged), extensible-array and version-2 B-tree chunk indexes, groups with dense link
storage (fractal heap + name index).  Skipped where libhdf5 is absent."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import openimpala_amd as oia

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC, LIB = "/opt/conda/include", "/opt/conda/lib"


@pytest.fixture(scope="module")
def writer(tmp_path_factory):
    if not (shutil.which("gcc") and os.path.exists(os.path.join(INC, "hdf5.h")) and os.path.exists(os.path.join(LIB, "libhdf5.so"))):
        pytest.skip("no libhdf5 to write the test files with")
    exe = str(tmp_path_factory.mktemp("h5w") / "h5_libhdf5_writer")
    r = subprocess.run(["gcc", "-O1", os.path.join(ROOT, "tests", "h5_libhdf5_writer.c"), "-I" + INC, "-L" + LIB, "-lhdf5",
                        "-Wl,-rpath," + LIB, "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("libhdf5 test writer does not build here: " + r.stderr[-300:])
    return exe


def expected(shape):
    z, y, x = np.indices(shape).astype(np.int64)
    return ((z * 7 + y * 3 + x * 5 + (z * y) % 11) % 251).astype(np.float64)


VARIANTS = ["old_contig", "old_chunk", "old_chunk_gz_shuf", "contig", "compact", "chunk", "chunk_gz", "chunk_gz_shuf", "single",
            "single_gz", "implicit", "paged", "paged_gz_shuf", "unlimz", "unlimz_gz_shuf", "unlimx", "maxbig", "nested_chunk",
            "chunk_partial", "unlimz_partial", "unlimz_tiny", "unlimz_tiny_huge_gz", "chunk_tiny_huge",
            # round 3: version-2 B-tree chunk index (two unlimited dimensions; plain, filtered, thousands of records = internal
            # nodes, partly written) and groups with dense link storage (fractal heap + name-index B-tree: 12 links in the
            # root group, 3000 links = indirect heap blocks and internal B-tree nodes, a dense group below the root)
            "unlim2", "unlim2_gz_shuf", "unlim2tiny", "unlim2tiny_gz", "unlim2_partial", "dense_chunk", "densebig_chunk_gz",
            "dense_in_group_contig"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("dtype", ["u8", "u16", "f32"])
def test_reads_what_libhdf5_writes(writer, tmp_path, variant, dtype):
    shape = (8, 6, 10) if variant == "compact" else ((40, 40, 40) if "paged" in variant else (23, 37, 45))
    if "huge" in variant:
        shape = (100, 100, 100)  # 50 000 chunks of 1 x 4 x 5: extensible-array super blocks and PAGED data blocks
    elif "tiny" in variant:
        shape = (40, 40, 40)     # 4 000 chunks: extensible-array super blocks
    if dtype != "u8" and variant not in ("chunk_gz_shuf", "unlimz_gz_shuf", "contig", "paged_gz_shuf", "old_chunk_gz_shuf", "unlim2_gz_shuf"):
        pytest.skip("element types are covered on a few layouts")
    path = str(tmp_path / f"{variant}_{dtype}.h5")
    subprocess.run([writer, path, variant] + [str(s) for s in shape] + [dtype], check=True, timeout=120)
    want = expected(shape)
    if "partial" in variant:
        want[:2] = 0
        want[5:] = 0
    dset = "g1/g2/image" if ("nested" in variant or "in_group" in variant) else "image"
    for thr in (100.5, 0.5):
        got = oia.HDF5Reader(path, dset).threshold(thr)
        assert got.shape == shape
        assert np.array_equal(got, (want > thr).astype(np.uint8)), (variant, dtype, thr)
    # a z-slab only
    part = oia.HDF5Reader(path, dset).threshold(100.5, z0=3, nz_local=4)
    assert np.array_equal(part, (want[3:7] > 100.5).astype(np.uint8))


def test_what_the_reader_refuses_it_names(writer, tmp_path):
    """Names beyond the dataset's: a missing object in a dense group is reported as missing, not as unreadable."""
    path = str(tmp_path / "dense.h5")
    subprocess.run([writer, path, "dense_chunk", "12", "20", "28", "u8"], check=True, timeout=120)
    with pytest.raises(oia.OpenImpalaError) as e:
        oia.HDF5Reader(path, "no_such_image").threshold(0.5)
    assert e.value.status == oia.OI_IO_ERROR and "no object" in str(e.value)
    with pytest.raises(oia.OpenImpalaError) as e:   # a sibling found through the heap, but not a volume
        oia.HDF5Reader(path, "other05").threshold(0.5)
    assert "rank 1" in str(e.value)
