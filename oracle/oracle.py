"""ctypes front end of the CPU oracle (oracle/libp4s_oracle.so).

TEST INFRASTRUCTURE ONLY.  Importers: tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libp4s_oracle.so")
_LIB_O3_PATH = os.path.join(_HERE, "libp4s_oracle_o3.so")
# second build of the same restatement for the CPU baseline (x86-64-v3 = AVX2 / BMI2 / FMA: what "-march=native"
# would pick on any recent server CPU, without tying the binary to the CPU it was compiled on)
O3_FLAGS = "gcc -O3 -march=x86-64-v3"
_REF_PATH = os.path.join(_HERE, "_ref", "libp4s_ref_l0.so")


def build():
    """Compile the restatement (and, where /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


class OrcError(C.Structure):
    _fields_ = [("code", C.c_int32), ("aux0", C.c_uint64), ("aux1", C.c_uint64)]


# message templates of the reference, keyed by error code (encoding.cc, exception.hh:42-56)
_CORRUPT = "Invalid or corrupted parquet file: "
MESSAGES = {
    1: "Unexpected end of DELTA_BINARY_PACKED header",
    2: "In DELTA_BINARY_PACKED number miniblocks per block is 0",
    3: "Unexpected end of DELTA_BINARY_PACKED block header",
    4: "Unexpected end of data in DELTA_BINARY_PACKED",
    5: "DELTA_BINARY_PACKED miniblock bit width {0} exceeds 64",
    6: "DELTA_BINARY_PACKED block size {0} smaller than miniblock count {1}",
    7: "Unexpected end of values in DELTA_LENGTH_BYTE_ARRAY",
    8: "Invalid prefix length in DELTA_BYTE_ARRAY",
    9: "Data size in BYTE_STREAM_SPLIT is not divisible by size of data type",
    10: _CORRUPT + "Illegal dictionary index bit width (should be 0 <= bit width <= 32, got {0})",
    11: _CORRUPT + "Dict index exceeds dict size (dict size = {0}, index = {1})",
    12: _CORRUPT + "End of page while reading BYTE_ARRAY length (needed {0}B, got {1}B)",
    13: _CORRUPT + "End of page while reading BYTE_ARRAY (needed {0}B, got {1}B)",
    14: _CORRUPT + "End of page while reading FIXED_LEN_BYTE_ARRAY (needed {0}B, got {1}B)",
    15: _CORRUPT + "End of page while reading levels (needed {0}B, got {1}B)",
    16: _CORRUPT + "Negative RLE levels length ({0})",
    17: _CORRUPT + "BIT_PACKED length exceeds int ({0}B)",
    18: "Unknown level encoding ({0})",
    19: _CORRUPT + "Levels length exceeds int ({0}B)",
    20: _CORRUPT + "No dictionary page found before a dictionary-encoded page",
    22: _CORRUPT + "Empty dictionary-encoded data page",
    24: _CORRUPT + "Corrupt snappy data",
    25: _CORRUPT + "Uncompression buffer size too small",
    26: "Could not decompress snappy.",
}


class OracleError(Exception):
    def __init__(self, code, aux0=0, aux1=0):
        self.code, self.aux0, self.aux1 = code, aux0, aux1
        a0 = aux0 - (1 << 64) if code == 16 and aux0 >= (1 << 63) else aux0
        super().__init__(MESSAGES.get(code, "error {}".format(code)).format(a0, aux1))


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        i64, vp, u8p = C.c_int64, C.c_void_p, C.c_void_p
        ep = C.POINTER(OrcError)
        _lib.orc_rle_decode.restype = i64
        _lib.orc_rle_decode.argtypes = [u8p, i64, C.c_int, vp, C.c_int, i64]
        _lib.orc_rle_decode_batches.restype = None
        _lib.orc_rle_decode_batches.argtypes = [u8p, i64, C.c_int, vp, C.c_int, vp, C.c_int, vp]
        _lib.orc_bitpacked_decode.restype = i64
        _lib.orc_bitpacked_decode.argtypes = [u8p, i64, C.c_int, vp, C.c_int, i64]
        _lib.orc_levels_v1.restype = i64
        _lib.orc_levels_v1.argtypes = [u8p, i64, C.c_int, C.c_uint32, C.c_uint32, i64, vp, C.c_int,
                                       C.POINTER(i64), ep]
        _lib.orc_levels_v2.restype = i64
        _lib.orc_levels_v2.argtypes = [u8p, i64, C.c_uint32, C.c_uint32, i64, vp, C.c_int, ep]
        _lib.orc_delta_decode.restype = i64
        _lib.orc_delta_decode.argtypes = [u8p, i64, vp, C.c_int, i64, C.POINTER(i64), ep]
        _lib.orc_rle_encode.restype = i64
        _lib.orc_rle_encode.argtypes = [vp, i64, C.c_int, vp, i64]
        _lib.orc_delta_encode.restype = i64
        _lib.orc_delta_encode.argtypes = [vp, C.c_int, i64, vp, i64]
        for name in ("orc_dlba_decode", "orc_dba_decode", "orc_plain_byte_array"):
            f = getattr(_lib, name)
            f.restype = i64
            f.argtypes = [u8p, i64, i64, vp, vp, i64, ep]
        _lib.orc_dba_sizes.restype = i64
        _lib.orc_dba_sizes.argtypes = [u8p, i64, i64, C.POINTER(i64), ep]
        _lib.orc_bss_decode.restype = i64
        _lib.orc_bss_decode.argtypes = [u8p, i64, C.c_int, vp, i64, ep]
        _lib.orc_plain_fixed.restype = i64
        _lib.orc_plain_fixed.argtypes = [u8p, i64, C.c_int, vp, i64]
        _lib.orc_plain_boolean.restype = i64
        _lib.orc_plain_boolean.argtypes = [u8p, i64, vp, i64]
        _lib.orc_rle_boolean.restype = i64
        _lib.orc_rle_boolean.argtypes = [u8p, i64, vp, i64]
        _lib.orc_plain_flba.restype = i64
        _lib.orc_plain_flba.argtypes = [u8p, i64, C.c_uint32, i64, vp, i64, ep]
        _lib.orc_snappy_decompress.restype = i64
        _lib.orc_snappy_decompress.argtypes = [u8p, i64, vp, i64, ep]
        _lib.orc_dict_decode.restype = i64
        _lib.orc_dict_decode.argtypes = [u8p, i64, vp, C.c_uint64, C.c_int, vp, i64, ep]
    return _lib


def lib_o3():
    """The -O3 build for timing, or None when the host CPU lacks what it was compiled for."""
    if not os.path.exists(_LIB_O3_PATH):
        return None
    try:
        flags = ""
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = line
                break
        if not all((" " + f + " ") in (flags + " ") for f in ("avx2", "bmi2", "fma")):
            return None
    except OSError:
        return None
    L = C.CDLL(_LIB_O3_PATH)
    L.orc_delta_decode.restype = C.c_int64
    L.orc_delta_decode.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int64),
                                   C.POINTER(OrcError)]
    return L


def ref():
    """The reference's own bit-level layer (oracle/_ref), or None when it was never built."""
    global _ref
    if _ref is None and os.path.exists(_REF_PATH):
        _ref = C.CDLL(_REF_PATH)
        i64, vp = C.c_int64, C.c_void_p
        _ref.ref_rle_decode_batches.restype = None
        _ref.ref_rle_decode_batches.argtypes = [vp, i64, C.c_int, vp, C.c_int, vp, C.c_int, vp]
        _ref.ref_bitpacked_decode.restype = i64
        _ref.ref_bitpacked_decode.argtypes = [vp, i64, C.c_int, vp, C.c_int, i64]
        _ref.ref_br_script.restype = None
        _ref.ref_br_script.argtypes = [vp, i64, vp, C.c_int, vp]
        _ref.ref_rle_encode.restype = i64
        _ref.ref_rle_encode.argtypes = [vp, i64, C.c_int, vp, i64]
    return _ref


_DT = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}


def _buf(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    a = np.ascontiguousarray(a, dtype=np.uint8)
    # keep a padded copy so that a zero-length page still has a valid pointer
    p = np.zeros(a.size + 16, dtype=np.uint8)
    p[: a.size] = a
    return p, a.size


def _check(err):
    if err.code != 0:
        raise OracleError(err.code, err.aux0, err.aux1)


def rle_decode(data, bit_width, n, elem_size=4, cap=None):
    """One RleDecoder::GetBatch<T>(out, n) call over the whole stream."""
    p, ln = _buf(data)
    cap = n if cap is None else cap
    out = np.zeros(max(cap, 1), dtype=_DT[elem_size])
    got = lib().orc_rle_decode(p.ctypes.data, ln, bit_width, out.ctypes.data, elem_size, n)
    return out[:got].copy()


def rle_decode_batches(data, bit_width, batch_sizes, elem_size=4, use_ref=False):
    p, ln = _buf(data)
    bs = np.asarray(batch_sizes, dtype=np.int64)
    out = np.zeros(int(bs.sum()) + 1, dtype=_DT[elem_size])
    counts = np.zeros(len(bs), dtype=np.int64)
    fn = ref().ref_rle_decode_batches if use_ref else lib().orc_rle_decode_batches
    fn(p.ctypes.data, ln, bit_width, out.ctypes.data, elem_size, bs.ctypes.data, len(bs), counts.ctypes.data)
    return out[: int(counts.sum())].copy(), counts


def bitpacked_decode(data, bit_width, n, elem_size=4, use_ref=False):
    p, ln = _buf(data)
    out = np.zeros(max(n, 1), dtype=_DT[elem_size])
    fn = ref().ref_bitpacked_decode if use_ref else lib().orc_bitpacked_decode
    got = fn(p.ctypes.data, ln, bit_width, out.ctypes.data, elem_size, n)
    return out[:got].copy()


def levels_v1(data, encoding, max_level, num_values, n, elem_size=4):
    p, ln = _buf(data)
    out = np.zeros(max(n, 1), dtype=_DT[elem_size])
    consumed = C.c_int64(0)
    err = OrcError()
    got = lib().orc_levels_v1(p.ctypes.data, ln, encoding, max_level, num_values, n, out.ctypes.data,
                              elem_size, C.byref(consumed), C.byref(err))
    _check(err)
    return out[:got].copy(), consumed.value


def levels_v2(data, max_level, num_values, n, elem_size=4):
    p, ln = _buf(data)
    out = np.zeros(max(n, 1), dtype=_DT[elem_size])
    err = OrcError()
    got = lib().orc_levels_v2(p.ctypes.data, ln, max_level, num_values, n, out.ctypes.data, elem_size,
                              C.byref(err))
    _check(err)
    return out[:got].copy()


def delta_decode(data, n, elem_size=8, return_bytes_left=False):
    p, ln = _buf(data)
    out = np.zeros(max(n, 1), dtype=_DT[elem_size])
    left = C.c_int64(0)
    err = OrcError()
    got = lib().orc_delta_decode(p.ctypes.data, ln, out.ctypes.data, elem_size, n, C.byref(left), C.byref(err))
    _check(err)
    r = out[:got].copy()
    return (r, left.value) if return_bytes_left else r


def delta_encode(values, elem_size=8):
    """The page delta_binary_packed_encoder<T> writes for `values` (one put_batch, one flush), padding bits 0."""
    v = np.ascontiguousarray(values, dtype=np.int32 if elem_size == 4 else np.int64)
    cap = 64 + (len(v) // 256 + 1) * (18 + 256 * elem_size)
    out = np.zeros(cap, dtype=np.uint8)
    got = lib().orc_delta_encode(v.ctypes.data, elem_size, len(v), out.ctypes.data, cap)
    if got < 0:
        raise ValueError("orc_delta_encode: capacity")
    return out[:got].tobytes()


def rle_encode_bound(n, w):
    """RleEncoder::MaxBufferSize + MinBufferSize (rle_encoding.hh:167-190), as src/encoding.cc:783-784 sizes its sink."""
    n, w = int(n), int(w)
    bytes_per_run = w
    num_runs = (n + 7) // 8
    literal_max = num_runs + num_runs * bytes_per_run
    min_repeated_run = 1 + (w + 7) // 8
    repeated_max = ((n + 7) // 8) * min_repeated_run
    max_literal_run = 1 + (512 * w + 7) // 8
    max_repeated_run = 5 + (w + 7) // 8
    return max(literal_max, repeated_max) + max(max_literal_run, max_repeated_run)


def rle_encode(values, w):
    """The hybrid stream RleEncoder writes for `values` (Put for each, then Flush) at bit width w."""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    cap = rle_encode_bound(len(v), w) + 16
    out = np.zeros(cap, dtype=np.uint8)
    got = lib().orc_rle_encode(v.ctypes.data, len(v), int(w), out.ctypes.data, cap)
    if got < 0:
        raise ValueError("orc_rle_encode: capacity")
    return out[:got].tobytes()


def ref_rle_encode(values, w):
    """The same from the reference's own RleEncoder (oracle/_ref)."""
    v = np.ascontiguousarray(values, dtype=np.uint64)
    cap = rle_encode_bound(len(v), w) + 64
    out = np.zeros(cap, dtype=np.uint8)
    got = ref().ref_rle_encode(v.ctypes.data, len(v), int(w), out.ctypes.data, cap)
    if got < 0:
        raise ValueError("ref_rle_encode: capacity")
    return out[:got].tobytes()


def _strings(fn, data, n, bytes_cap):
    p, ln = _buf(data)
    offsets = np.zeros(n + 1, dtype=np.int64)
    out = np.zeros(max(bytes_cap, 1), dtype=np.uint8)
    err = OrcError()
    got = fn(p.ctypes.data, ln, n, offsets.ctypes.data, out.ctypes.data, bytes_cap, C.byref(err))
    _check(err)
    if got == -2:
        raise MemoryError("bytes_cap too small")
    offsets = offsets[: got + 1].copy()
    return offsets, out[: int(offsets[-1])].copy()


def dlba_decode(data, n, bytes_cap=None):
    return _strings(lib().orc_dlba_decode, data, n, len(data) if bytes_cap is None else bytes_cap)


def dba_sizes(data, n):
    p, ln = _buf(data)
    total = C.c_int64(0)
    err = OrcError()
    got = lib().orc_dba_sizes(p.ctypes.data, ln, n, C.byref(total), C.byref(err))
    _check(err)
    return got, total.value


def dba_decode(data, n):
    got, total = dba_sizes(data, n)
    return _strings(lib().orc_dba_decode, data, got, total)


def plain_byte_array(data, n):
    return _strings(lib().orc_plain_byte_array, data, n, len(data))


def snappy_decompress(data, out_cap):
    """snappy_compressor::decompress into a buffer of out_cap bytes (src/compression.cc:49-65)."""
    p, ln = _buf(data)
    out = np.zeros(max(int(out_cap), 1), dtype=np.uint8)
    err = OrcError()
    got = lib().orc_snappy_decompress(p.ctypes.data, ln, out.ctypes.data, int(out_cap), C.byref(err))
    _check(err)
    return out[:got].copy()


def plain_flba(data, fixed_len, n):
    p, ln = _buf(data)
    out = np.zeros(max(ln, 1), dtype=np.uint8)
    err = OrcError()
    got = lib().orc_plain_flba(p.ctypes.data, ln, fixed_len, n, out.ctypes.data, ln, C.byref(err))
    _check(err)
    return got, out[: got * fixed_len].copy()


def bss_decode(data, elem_size, n):
    p, ln = _buf(data)
    out = np.zeros(max(ln, 8), dtype=np.uint8)
    err = OrcError()
    got = lib().orc_bss_decode(p.ctypes.data, ln, elem_size, out.ctypes.data, n, C.byref(err))
    _check(err)
    return out[: got * elem_size].copy()


def plain_fixed(data, elem_size, n):
    p, ln = _buf(data)
    out = np.zeros(max(ln, 16), dtype=np.uint8)
    got = lib().orc_plain_fixed(p.ctypes.data, ln, elem_size, out.ctypes.data, n)
    return out[: got * elem_size].copy()


def plain_boolean(data, n):
    p, ln = _buf(data)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    got = lib().orc_plain_boolean(p.ctypes.data, ln, out.ctypes.data, n)
    return out[:got].copy()


def rle_boolean(data, n):
    p, ln = _buf(data)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    got = lib().orc_rle_boolean(p.ctypes.data, ln, out.ctypes.data, n)
    return out[:got].copy()


def dict_decode(data, dictionary, n):
    """dictionary: 1-D numpy array of fixed-width elements (a zero pad element is appended)."""
    p, ln = _buf(data)
    d = np.ascontiguousarray(dictionary)
    es = d.dtype.itemsize
    padded = np.zeros(d.size + 1, dtype=d.dtype)
    padded[: d.size] = d
    out = np.zeros(max(n, 1), dtype=d.dtype)
    err = OrcError()
    got = lib().orc_dict_decode(p.ctypes.data, ln, padded.ctypes.data, d.size, es, out.ctypes.data, n,
                                C.byref(err))
    _check(err)
    return out[:got].copy()


def br_script(data, ops, use_ref=False):
    """Scripted BitReader session: ops = [(opcode, arg)], see orc_br_script."""
    p, ln = _buf(data)
    o = np.ascontiguousarray(np.asarray(ops, dtype=np.int32).reshape(-1, 2))
    res = np.zeros((len(o), 2), dtype=np.int64)
    if use_ref:
        ref().ref_br_script(p.ctypes.data, ln, o.ctypes.data, len(o), res.ctypes.data)
    else:
        f = lib().orc_br_script
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        f(p.ctypes.data, ln, o.ctypes.data, len(o), res.ctypes.data)
    return res
