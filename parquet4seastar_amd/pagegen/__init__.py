"""Synthetic Parquet page encoders (ctypes front end of libp4s_gen.so).

They build decoder INPUTS for the tests and the benchmark; nothing here is on
the decode path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libp4s_gen.so")
_lib = None


def build():
    src = os.path.join(_HERE, "p4s_gen.c")
    if os.path.exists(_LIB_PATH) and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src):
        return
    subprocess.run(["gcc", "-O2", "-fPIC", "-std=c99", "-shared", "-o", _LIB_PATH, src], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        i64, vp = C.c_int64, C.c_void_p
        _lib.gen_bitpack.restype = i64
        _lib.gen_bitpack.argtypes = [vp, i64, C.c_int, vp, i64]
        _lib.gen_rle_encode.restype = i64
        _lib.gen_rle_encode.argtypes = [vp, i64, C.c_int, C.c_int, C.c_int, vp, i64]
        _lib.gen_rle_from_runs.restype = i64
        _lib.gen_rle_from_runs.argtypes = [vp, vp, vp, vp, i64, C.c_int, vp, i64]
        _lib.gen_delta_encode.restype = i64
        _lib.gen_delta_encode.argtypes = [vp, C.c_int, i64, C.c_uint32, C.c_uint32, vp, i64]
        _lib.gen_delta_max_size.restype = i64
        _lib.gen_delta_max_size.argtypes = [C.c_int, i64, C.c_uint32, C.c_uint32]
        _lib.gen_bss_encode.restype = None
        _lib.gen_bss_encode.argtypes = [vp, C.c_int, i64, vp]
    return _lib


def bit_width(max_n):
    """bit_width() of encoding.hh:35-37 in the reference: 0 for 0, else floor(log2)+1."""
    return int(max_n).bit_length()


def bitpack(values, w):
    v = np.ascontiguousarray(values, dtype=np.uint32)
    out = np.zeros((v.size * w + 7) // 8 + 8, dtype=np.uint8)
    n = lib().gen_bitpack(v.ctypes.data, v.size, w, out.ctypes.data, out.size)
    assert n >= 0
    return out[:n].tobytes()


def rle_encode(values, w, max_groups=63, min_repeat=8):
    """RLE/bit-packed hybrid stream (no length prefix, no width byte)."""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    cap = (v.size + 8) * ((w + 7) // 8 + 1) + (v.size // 8 + 2) * 6 + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().gen_rle_encode(v.ctypes.data, v.size, w, max_groups, min_repeat, out.ctypes.data, cap)
    assert n >= 0, "rle_encode: buffer too small"
    return out[:n].tobytes()


def rle_from_runs(runs, w):
    """runs: list of ('rle', count, value) or ('lit', [values...]) with len(values) % 8 == 0."""
    kinds, counts, reps, lits = [], [], [], []
    for r in runs:
        if r[0] == "rle":
            kinds.append(0); counts.append(r[1]); reps.append(r[2] & 0xFFFFFFFF)
        else:
            vals = list(r[1])
            assert len(vals) % 8 == 0
            kinds.append(1); counts.append(len(vals) // 8); reps.append(0)
            lits.extend(vals)
    k = np.asarray(kinds, dtype=np.int32)
    c = np.asarray(counts, dtype=np.int64)
    rv = np.asarray(reps, dtype=np.uint32)
    lv = np.asarray(lits + [0], dtype=np.uint32)
    cap = len(lits) * 5 + len(runs) * 16 + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().gen_rle_from_runs(k.ctypes.data, c.ctypes.data, rv.ctypes.data, lv.ctypes.data, len(runs), w,
                                out.ctypes.data, cap)
    assert n >= 0
    return out[:n].tobytes()


def delta_encode(values, block_size=256, n_miniblocks=8, as_array=False):
    """DELTA_BINARY_PACKED page for an int32 or int64 numpy array."""
    v = np.ascontiguousarray(values)
    assert v.dtype in (np.int32, np.int64)
    assert block_size % n_miniblocks == 0 and n_miniblocks <= 4096
    es = v.dtype.itemsize
    cap = lib().gen_delta_max_size(es, v.size, block_size, n_miniblocks)
    out = np.zeros(cap, dtype=np.uint8)
    n = lib().gen_delta_encode(v.ctypes.data, es, v.size, block_size, n_miniblocks, out.ctypes.data, cap)
    assert n >= 0
    return out[:n].copy() if as_array else out[:n].tobytes()


def bss_encode(values):
    v = np.ascontiguousarray(values)
    es = v.dtype.itemsize
    return np.ascontiguousarray(v.view(np.uint8).reshape(-1, es).T).tobytes()


def dlba_encode(strings, block_size=128, n_miniblocks=4):
    """DELTA_LENGTH_BYTE_ARRAY page from a list of bytes objects."""
    lens = np.asarray([len(s) for s in strings], dtype=np.int32)
    return delta_encode(lens, block_size, n_miniblocks) + b"".join(strings)


def dba_encode(strings, block_size=128, n_miniblocks=4):
    """DELTA_BYTE_ARRAY page (maximal shared prefixes) from a list of bytes objects."""
    prefixes, suffixes = [], []
    prev = b""
    for s in strings:
        p = 0
        m = min(len(prev), len(s))
        while p < m and prev[p] == s[p]:
            p += 1
        prefixes.append(p)
        suffixes.append(s[p:])
        prev = s
    return (delta_encode(np.asarray(prefixes, dtype=np.int32), block_size, n_miniblocks)
            + dlba_encode(suffixes, block_size, n_miniblocks))


def dba_encode_arrays(prefix_lens, suffix_lens, suffix_bytes, block_size=128, n_miniblocks=4):
    """DELTA_BYTE_ARRAY page from already-split numpy arrays (bulk generator)."""
    return (delta_encode(np.asarray(prefix_lens, dtype=np.int32), block_size, n_miniblocks)
            + delta_encode(np.asarray(suffix_lens, dtype=np.int32), block_size, n_miniblocks)
            + bytes(suffix_bytes))


def plain_byte_array_encode(strings):
    return b"".join(len(s).to_bytes(4, "little") + s for s in strings)


def dict_page_values_encode(indices, w, max_groups=63):
    """Values section of an RLE_DICTIONARY data page: width byte + hybrid stream."""
    return bytes([w]) + rle_encode(indices, w, max_groups)


def levels_v1_rle_encode(levels, max_level):
    body = rle_encode(levels, bit_width(max_level))
    return len(body).to_bytes(4, "little") + body


# ---------------------------------------------------------------------------
# Column-chunk assembly: Thrift compact-protocol PageHeader + page body
# (field ids: doc/parquet/parquet.thrift:507-659 of the reference).
# ---------------------------------------------------------------------------
PAGE_DATA, PAGE_INDEX, PAGE_DICTIONARY, PAGE_DATA_V2 = 0, 1, 2, 3
ENC_PLAIN, ENC_PLAIN_DICTIONARY, ENC_RLE, ENC_BIT_PACKED = 0, 2, 3, 4
ENC_DELTA_BINARY_PACKED, ENC_DELTA_LENGTH_BYTE_ARRAY, ENC_DELTA_BYTE_ARRAY = 5, 6, 7
ENC_RLE_DICTIONARY, ENC_BYTE_STREAM_SPLIT = 8, 9
TYPE_BOOLEAN, TYPE_INT32, TYPE_INT64, TYPE_INT96, TYPE_FLOAT, TYPE_DOUBLE, TYPE_BYTE_ARRAY, TYPE_FLBA = range(8)


def _uleb(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _zz(v):
    return _uleb((v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1)


class _Struct:
    """One Thrift compact struct under construction (i32 and bool fields, nested structs)."""

    def __init__(self):
        self.out = bytearray()
        self.last = 0

    def _field(self, fid, ftype):
        d = fid - self.last
        if 0 < d <= 15:
            self.out.append((d << 4) | ftype)
        else:
            self.out.append(ftype)
            self.out += _zz(fid)
        self.last = fid

    def i32(self, fid, v):
        self._field(fid, 5)
        self.out += _zz(v)

    def boolean(self, fid, v):
        self._field(fid, 1 if v else 2)

    def struct(self, fid, s):
        self._field(fid, 12)
        self.out += s.done()

    def done(self):
        return bytes(self.out) + b"\x00"


def page_header(page_type, body_size, *, num_values=0, encoding=ENC_PLAIN, def_encoding=ENC_RLE, rep_encoding=ENC_RLE,
                num_nulls=0, num_rows=0, def_len=0, rep_len=0, is_compressed=None, with_stats=False, uncompressed_size=None):
    """body_size = bytes that follow the header (compressed_page_size); uncompressed_size defaults to the same."""
    h = _Struct()
    h.i32(1, page_type)
    h.i32(2, body_size if uncompressed_size is None else uncompressed_size)
    h.i32(3, body_size)
    if page_type == PAGE_DATA:
        d = _Struct()
        d.i32(1, num_values)
        d.i32(2, encoding)
        d.i32(3, def_encoding)
        d.i32(4, rep_encoding)
        if with_stats:      # a Statistics struct the reader has to skip: null_count (i64) and a binary max
            s = _Struct()
            s._field(3, 6)
            s.out += _zz(5)
            s._field(5, 8)
            s.out += _uleb(3) + b"xyz"
            d.struct(5, s)
        h.struct(5, d)
    elif page_type == PAGE_DICTIONARY:
        d = _Struct()
        d.i32(1, num_values)
        d.i32(2, encoding)
        h.struct(7, d)
    elif page_type == PAGE_DATA_V2:
        d = _Struct()
        d.i32(1, num_values)
        d.i32(2, num_nulls)
        d.i32(3, num_rows)
        d.i32(4, encoding)
        d.i32(5, def_len)
        d.i32(6, rep_len)
        if is_compressed is not None:
            d.boolean(7, is_compressed)
        h.struct(8, d)
    return h.done()


def data_page_v1(values_section, num_values, encoding, *, rep_levels=None, max_rep=0, def_levels=None, max_def=0,
                 level_encoding=ENC_RLE, with_stats=False, compress=None):
    """A v1 data page: [rep levels][def levels][values]; a level section exists only when its maximum is > 0.
    compress: bytes -> bytes applied to the whole body (a page of a compressed column chunk)."""
    body = b""
    for lv, mx in ((rep_levels, max_rep), (def_levels, max_def)):
        if mx > 0:
            if level_encoding == ENC_RLE:
                body += levels_v1_rle_encode(lv, mx)
            else:
                body += bitpack(lv, bit_width(mx))
    body += values_section
    stored = compress(body) if compress else body
    return page_header(PAGE_DATA, len(stored), num_values=num_values, encoding=encoding, def_encoding=level_encoding,
                       rep_encoding=level_encoding, with_stats=with_stats, uncompressed_size=len(body)) + stored


def data_page_v2(values_section, num_values, encoding, *, rep_levels=None, max_rep=0, def_levels=None, max_def=0,
                 num_nulls=0, compress=None):
    """compress: bytes -> bytes applied to the values (the level sections of a v2 page are never compressed)."""
    rep = rle_encode(rep_levels, bit_width(max_rep)) if max_rep > 0 else b""
    dfn = rle_encode(def_levels, bit_width(max_def)) if max_def > 0 else b""
    body = rep + dfn + values_section
    stored = rep + dfn + compress(values_section) if compress else body
    return page_header(PAGE_DATA_V2, len(stored), num_values=num_values, encoding=encoding, num_nulls=num_nulls,
                       num_rows=num_values, def_len=len(dfn), rep_len=len(rep), is_compressed=bool(compress),
                       uncompressed_size=len(body)) + stored


def dictionary_page(plain_values, num_values, compress=None):
    stored = compress(plain_values) if compress else plain_values
    return page_header(PAGE_DICTIONARY, len(stored), num_values=num_values, encoding=ENC_PLAIN,
                       uncompressed_size=len(plain_values)) + stored
