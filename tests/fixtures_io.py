"""Readers for the reference's own test fixtures copied (as data) into tests/golden/.

* test_1.bson                 -- reference test/reftest-files/test_1.bson (test/part1.jl:24-40)
* fortran/*.bin               -- reference test/reftest-files/fortran/*.bin (test/part2.jl:18-38);
                                 format = Int32 nx, Int32 ny, nx*ny Float64 column-major
                                 (scripts-part2/part2_utils.jl:11-19)
"""
import os
import struct

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _parse_doc(b, off):
    n = struct.unpack_from("<i", b, off)[0]
    end = off + n - 1
    off += 4
    out = {}
    while off < end:
        t = b[off]
        off += 1
        e = b.index(b"\0", off)
        name = b[off:e].decode()
        off = e + 1
        if t == 0x01:
            v = struct.unpack_from("<d", b, off)[0]
            off += 8
        elif t == 0x02:
            ln = struct.unpack_from("<i", b, off)[0]
            v = b[off + 4 : off + 4 + ln - 1].decode()
            off += 4 + ln
        elif t in (0x03, 0x04):
            v, off = _parse_doc(b, off)
            if t == 0x04:
                v = [v[k] for k in sorted(v, key=int)]
        elif t == 0x05:
            ln = struct.unpack_from("<i", b, off)[0]
            v = b[off + 5 : off + 5 + ln]
            off += 5 + ln
        elif t == 0x08:
            v = bool(b[off])
            off += 1
        elif t == 0x0A:
            v = None
        elif t == 0x10:
            v = struct.unpack_from("<i", b, off)[0]
            off += 4
        elif t == 0x12:
            v = struct.unpack_from("<q", b, off)[0]
            off += 8
        else:
            raise ValueError("unsupported BSON element type 0x%02x" % t)
        out[name] = v
    return out, end + 1


def load_bson_arrays(path):
    """BSON.jl array documents -> {name: Fortran-ordered float64 ndarray}."""
    with open(path, "rb") as fh:
        doc, _ = _parse_doc(fh.read(), 0)
    out = {}
    for k, v in doc.items():
        assert v["tag"] == "array" and v["type"]["name"] == ["Core", "Float64"]
        out[k] = np.frombuffer(v["data"], dtype="<f8").reshape(v["size"], order="F").copy(order="F")
    return out


def part1_reference():
    return load_bson_arrays(os.path.join(GOLDEN, "test_1.bson"))


def load_bin(name):
    """scripts-part2/part2_utils.jl:11-19 `load`."""
    path = os.path.join(GOLDEN, "fortran", name)
    with open(path, "rb") as fh:
        nx, ny = struct.unpack("<ii", fh.read(8))
        a = np.frombuffer(fh.read(8 * nx * ny), dtype="<f8").reshape((nx, ny), order="F")
    return np.asfortranarray(a, dtype=np.float64)


def splitmix64_uniform(n, seed=1):
    """Counter-based U[0,1): splitmix64 of (linear index + seed*golden), identical on CPU and GPU
    (Julia's @rand stream cannot be reproduced)."""
    z = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    z ^= z >> np.uint64(30)
    z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27)
    z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
