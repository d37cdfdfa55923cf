"""Minimal classic-format HDF5 writer for reader tests (superblock v0, symbol-table groups,
v1 object headers, contiguous or chunked layout with optional shuffle + deflate).
h5py is not available in the image; this writes just enough of the format spec."""
import struct
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


def _msg(mtype, body):
    body = body + b"\0" * (-len(body) % 8)
    return struct.pack("<HHB3x", mtype, len(body), 0) + body


def _ohdr(msgs):
    body = b"".join(msgs)
    return struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body


def _datatype(dt):
    dt = np.dtype(dt)
    big = dt.byteorder == ">" or (dt.byteorder == "=" and False)
    if dt.kind in "ui":
        bits0 = (1 if big else 0) | (0x08 if dt.kind == "i" else 0)
        return struct.pack("<BBBBI", 0x10 | 0, bits0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    cls = struct.pack("<BBBBI", 0x10 | 1, (1 if big else 0) | 0x20, 0x1F if dt.itemsize == 4 else 0x3F, 0, dt.itemsize)
    if dt.itemsize == 4:
        return cls + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
    return cls + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)


def write_h5(path, arr, names=("image",), chunks=None, gzip=False, shuffle=False, craft=None):
    """arr: 3-D numpy array (Z, Y, X); names: group path ending in the dataset name.
    craft (chunked only): "duplicate" lists the first chunk's box twice in the index, "misaligned" moves the second chunk off the
    chunk grid -- what a corrupt or hostile file could hold; the reader must refuse both."""
    arr = np.ascontiguousarray(arr)
    es = arr.dtype.itemsize
    blob = bytearray(b"\0" * 96)  # superblock (filled last)

    def put(b):
        while len(blob) % 8:
            blob.append(0)
        off = len(blob)
        blob.extend(b)
        return off

    # ---- dataset raw data ----
    filters = b""
    if chunks is None:
        data_addr = put(arr.tobytes())
        layout = struct.pack("<BBQQ", 3, 1, data_addr, arr.nbytes)
    else:
        cz, cy, cx = chunks
        entries = []
        for z0 in range(0, arr.shape[0], cz):
            for y0 in range(0, arr.shape[1], cy):
                for x0 in range(0, arr.shape[2], cx):
                    c = np.zeros(chunks, arr.dtype)
                    part = arr[z0:z0 + cz, y0:y0 + cy, x0:x0 + cx]
                    c[:part.shape[0], :part.shape[1], :part.shape[2]] = part
                    raw = c.tobytes()
                    if shuffle and es > 1:
                        raw = np.frombuffer(raw, np.uint8).reshape(-1, es).T.tobytes()
                    if gzip:
                        raw = zlib.compress(raw, 4)
                    entries.append(((z0, y0, x0), put(raw), len(raw)))
        if craft == "duplicate":
            entries[1] = (entries[0][0], entries[1][1], entries[1][2])
        elif craft == "misaligned":
            (z0, y0, x0), a_, n_ = entries[1]
            entries[1] = ((z0, y0, max(1, x0 - 1)), a_, n_)
        node = struct.pack("<4sBBHQQ", b"TREE", 1, 0, len(entries), UNDEF, UNDEF)
        for (z0, y0, x0), addr, nbytes in entries:
            node += struct.pack("<IIQQQQ", nbytes, 0, z0, y0, x0, 0) + struct.pack("<Q", addr)
        node += struct.pack("<IIQQQQ", 0, 0, arr.shape[0], arr.shape[1], arr.shape[2], 0)  # final key
        btree = put(node)
        layout = struct.pack("<BBBQ", 3, 2, 4, btree) + struct.pack("<4I", cz, cy, cx, es)
        flt = []
        if shuffle and es > 1:
            flt.append(struct.pack("<HHHH", 2, 0, 0, 1) + struct.pack("<I", es) + b"\0" * 4)
        if gzip:
            flt.append(struct.pack("<HHHH", 1, 0, 0, 1) + struct.pack("<I", 4) + b"\0" * 4)
        if flt:
            filters = _msg(0x000B, struct.pack("<BB6x", 1, len(flt)) + b"".join(flt))
    dataspace = struct.pack("<BBB5x", 1, 3, 0) + struct.pack("<3Q", *arr.shape)
    msgs = [_msg(0x0001, dataspace), _msg(0x0003, _datatype(arr.dtype)), _msg(0x0008, layout)]
    if filters:
        msgs.append(filters)
    obj = put(_ohdr(msgs))

    # ---- groups from the innermost outwards ----
    for name in reversed(names):
        heap_data = b"\0" * 8 + name.encode() + b"\0"
        heap_data += b"\0" * (-len(heap_data) % 8)
        heap_data_addr = put(heap_data + b"\0" * 16)
        heap = put(struct.pack("<4sB3xQQQ", b"HEAP", 0, len(heap_data) + 16, len(heap_data), heap_data_addr))
        snod = put(struct.pack("<4sBBH", b"SNOD", 1, 0, 1) + struct.pack("<QQII16x", 8, obj, 0, 0))
        tree = put(struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, snod, 8))
        obj = put(_ohdr([_msg(0x0011, struct.pack("<QQ", tree, heap))]))

    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, len(blob), UNDEF) + struct.pack("<QQII16x", 0, obj, 0, 0)
    blob[:len(sb)] = sb
    with open(path, "wb") as f:
        f.write(bytes(blob))
