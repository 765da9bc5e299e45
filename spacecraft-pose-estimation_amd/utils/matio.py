"""MATLAB Level-5 MAT-file writer / reader for the prediction files of the pipeline.

The reference exchanges `pred*.mat` between its two stages with scipy.io
(landmark_regression/lib/dataset/events.py:121-125 writes {'preds': (N, J, 3) float32};
pose_estimation/export_predicted_poses_real.py:172-173 reads it back).  This module writes and reads that file format
itself -- numeric N-d arrays, one variable per matrix element, the layout scipy.io.savemat(..., do_compression=False)
produces -- so that the file-format boundary does not depend on SciPy.

Level-5 layout: 128-byte header (116 bytes of text, 8 bytes subsystem offset, version 0x0100, endian indicator "IM"),
then data elements <type:uint32, nbytes:uint32, payload padded to 8 bytes>.  A variable is an miMATRIX element
holding the sub-elements array-flags, dimensions, name, real part; values are stored in column-major order.
"""
import struct
import time
import zlib

import numpy as np

MI_INT8, MI_UINT8, MI_INT16, MI_UINT16, MI_INT32, MI_UINT32, MI_SINGLE, MI_DOUBLE, MI_INT64, MI_UINT64 = 1, 2, 3, 4, 5, 6, 7, 9, 12, 13
MI_MATRIX, MI_COMPRESSED = 14, 15
# numpy dtype -> (mx class, mi storage type)
_CLASS = {"float64": (6, MI_DOUBLE), "float32": (7, MI_SINGLE), "int8": (8, MI_INT8), "uint8": (9, MI_UINT8),
          "int16": (10, MI_INT16), "uint16": (11, MI_UINT16), "int32": (12, MI_INT32), "uint32": (13, MI_UINT32),
          "int64": (14, MI_INT64), "uint64": (15, MI_UINT64)}
_MI_DTYPE = {MI_INT8: "i1", MI_UINT8: "u1", MI_INT16: "i2", MI_UINT16: "u2", MI_INT32: "i4", MI_UINT32: "u4",
             MI_SINGLE: "f4", MI_DOUBLE: "f8", MI_INT64: "i8", MI_UINT64: "u8"}
_MX_DTYPE = {6: "f8", 7: "f4", 8: "i1", 9: "u1", 10: "i2", 11: "u2", 12: "i4", 13: "u4", 14: "i8", 15: "u8"}


def _element(mi_type, payload):
    pad = (-len(payload)) % 8
    return struct.pack("<II", mi_type, len(payload)) + payload + b"\0" * pad


def _matrix(name, arr):
    arr = np.asarray(arr)
    if arr.dtype == np.bool_:
        arr = arr.astype(np.uint8)
    key = arr.dtype.name
    if key not in _CLASS:
        raise TypeError("savemat: dtype %s of variable %r is not supported (numeric arrays only)" % (arr.dtype, name))
    if arr.ndim == 0:
        arr = arr.reshape(1, 1)
    elif arr.ndim == 1:
        arr = arr.reshape(1, -1)                      # scipy's oned_as='row'
    mx, mi = _CLASS[key]
    body = _element(MI_UINT32, struct.pack("<II", mx, 0))                                   # array flags: class, no complex/global/logical
    body += _element(MI_INT32, struct.pack("<%di" % arr.ndim, *arr.shape))
    body += _element(MI_INT8, name.encode("ascii"))
    body += _element(mi, np.asfortranarray(arr).astype(arr.dtype.newbyteorder("<"), copy=False).tobytes(order="F"))
    return _element(MI_MATRIX, body)


def savemat(path, mdict):
    """Write the numeric arrays of `mdict` (name -> array) as a Level-5 MAT-file."""
    text = ("MATLAB 5.0 MAT-file Platform: scpose, Created on: %s" % time.asctime()).encode("ascii")
    header = text[:116].ljust(116, b" ") + b"\0" * 8 + struct.pack("<H", 0x0100) + b"IM"
    with open(path, "wb") as fh:
        fh.write(header)
        for name, arr in mdict.items():
            if name.startswith("_"):
                continue
            fh.write(_matrix(name, arr))


def _read_tag(buf, pos):
    word = struct.unpack_from("<I", buf, pos)[0]
    if word >> 16:                                    # small data element: type in the low half, byte count in the high half
        n = word >> 16
        return word & 0xFFFF, n, pos + 4, pos + 8
    mi, n = struct.unpack_from("<II", buf, pos)
    return mi, n, pos + 8, pos + 8 + n + ((-n) % 8)


def _parse_matrix(buf):
    pos = 0
    mi, n, dpos, pos = _read_tag(buf, pos)            # array flags
    flags = struct.unpack_from("<I", buf, dpos)[0]
    mx = flags & 0xFF
    if flags & 0x0800:
        raise ValueError("loadmat: complex arrays are not supported")
    mi, n, dpos, pos = _read_tag(buf, pos)            # dimensions
    dims = struct.unpack_from("<%di" % (n // 4), buf, dpos)
    mi, n, dpos, pos = _read_tag(buf, pos)            # name
    name = bytes(buf[dpos:dpos + n]).decode("ascii")
    if mx not in _MX_DTYPE:
        raise ValueError("loadmat: variable %r has class %d (only numeric arrays are supported)" % (name, mx))
    mi, n, dpos, pos = _read_tag(buf, pos)            # real part (may be stored in a narrower type than the class)
    vals = np.frombuffer(buf, dtype="<" + _MI_DTYPE[mi], count=n // int(_MI_DTYPE[mi][1:]), offset=dpos)
    return name, vals.astype(_MX_DTYPE[mx]).reshape(dims, order="F")


def loadmat(path):
    """Read every numeric variable of a Level-5 MAT-file (plain or zlib-compressed elements) into a dict."""
    with open(path, "rb") as fh:
        data = fh.read()
    if len(data) < 128 or data[126:128] != b"IM":
        raise ValueError("loadmat: %s is not a little-endian Level-5 MAT-file" % path)
    out, pos = {}, 128
    while pos + 8 <= len(data):
        mi, n, dpos, nxt = _read_tag(data, pos)
        chunk = data[dpos:dpos + n]
        if mi == MI_COMPRESSED:
            nxt = dpos + n                             # compressed elements are not padded
            chunk = zlib.decompress(chunk)
            mi, n_in, dpos_in, _ = _read_tag(chunk, 0)
            chunk = chunk[dpos_in:dpos_in + n_in]
        if mi == MI_MATRIX:
            name, arr = _parse_matrix(memoryview(chunk))
            out[name] = arr
        pos = nxt
    return out
