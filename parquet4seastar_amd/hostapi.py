"""ctypes binding of the C++ host layer (host/libp4s_host.so: value_decoder<T>,
level_decoder, column_chunk_reader<T>, column_chunk_decoder<T> over the GPU kernels).

Used by the parity tests and bench.py.  There is no CPU fallback: without the
library or a GPU every call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "host", "libp4s_host.so")

EXPORTS = ["p4sh_value_decode", "p4sh_levels_decode", "p4sh_chunk_open", "p4sh_chunk_open_codec", "p4sh_chunk_close", "p4sh_chunk_read_batch",
           "p4sh_chunk_plan", "p4sh_chunk_launch", "p4sh_chunk_collect", "p4sh_chunk_page_error", "p4sh_page_index",
           "p4sh_chunk_device_ptr", "p4sh_current_ctx", "p4sh_chunk_positional",
           "p4sh_flat_open", "p4sh_flat_close", "p4sh_flat_add_field", "p4sh_flat_read_all"]

# physical types (parquet.thrift Type) -> numpy dtype of a fixed-width value
FIXED_DTYPES = {0: np.dtype(np.uint8), 1: np.dtype("<i4"), 2: np.dtype("<i8"), 3: np.dtype(("<i4", 3)),
                4: np.dtype("<f4"), 5: np.dtype("<f8")}
_ERR = 1024
_lib = None


class HostError(RuntimeError):
    """An exception of the C++ layer; the text is what the reference's decoder throws."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HostError("host library missing: build it with `make -C parquet4seastar_amd/host` (expected %s)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u64, i64, u32, i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_uint32, C.c_int32
        L.p4sh_value_decode.restype = i64
        L.p4sh_value_decode.argtypes = [C.c_int, C.c_int, i32, vp, u64, i64, vp, u64, vp, C.c_int, vp, u64, vp, vp, u64, vp,
                                        C.c_char_p, u64]
        L.p4sh_levels_decode.restype = i64
        L.p4sh_levels_decode.argtypes = [C.c_int, C.c_int, u32, u32, vp, u64, vp, C.c_int, vp, u64, C.POINTER(i64),
                                         C.c_char_p, u64]
        L.p4sh_chunk_open.restype = vp
        L.p4sh_chunk_open.argtypes = [C.c_int, vp, u64, u32, u32, i32, C.c_char_p, u64]
        L.p4sh_chunk_open_codec.restype = vp
        L.p4sh_chunk_open_codec.argtypes = [C.c_int, vp, u64, u32, u32, i32, C.c_int, C.c_char_p, u64]
        L.p4sh_chunk_close.restype = None
        L.p4sh_chunk_close.argtypes = [vp]
        L.p4sh_chunk_read_batch.restype = i64
        L.p4sh_chunk_read_batch.argtypes = [vp, u64, vp, vp, vp, vp, vp, u64, C.POINTER(u64), C.c_char_p, u64]
        for name in ("p4sh_chunk_plan", "p4sh_chunk_launch"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [vp, C.c_char_p, u64]
        L.p4sh_chunk_collect.restype = C.c_int
        L.p4sh_chunk_collect.argtypes = [vp, vp, C.c_char_p, u64]
        L.p4sh_flat_open.restype = vp
        L.p4sh_flat_open.argtypes = []
        L.p4sh_flat_close.restype = None
        L.p4sh_flat_close.argtypes = [vp]
        L.p4sh_flat_add_field.restype = C.c_int
        L.p4sh_flat_add_field.argtypes = [vp, C.c_int, C.c_char_p, vp, u64, C.c_uint32, C.c_char_p, u64]
        L.p4sh_flat_read_all.restype = C.c_int64
        L.p4sh_flat_read_all.argtypes = [vp, C.c_char_p, u64, vp, C.c_char_p, u64]
        L.p4sh_chunk_positional.restype = C.c_int64
        L.p4sh_chunk_positional.argtypes = [vp, vp, vp, u64, vp, C.c_char_p, u64]
        L.p4sh_chunk_page_error.restype = C.c_int
        L.p4sh_chunk_page_error.argtypes = [vp, u64, C.c_char_p, u64]
        L.p4sh_chunk_device_ptr.restype = vp
        L.p4sh_chunk_device_ptr.argtypes = [vp, C.c_int]
        L.p4sh_current_ctx.restype = vp
        L.p4sh_current_ctx.argtypes = []
        L.p4sh_page_index.restype = C.c_int
        L.p4sh_page_index.argtypes = [vp, u64, vp, u64, C.POINTER(u64), C.c_char_p, u64]
        _lib = L
    return _lib


def _arr(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    p = np.zeros(a.size + 16, dtype=np.uint8)
    p[:a.size] = a
    return p, a.size


def value_decode(ptype, encoding, page, batches, *, type_length=None, dict_page=None, dict_count=-1, cap=None,
                 bytes_cap=None):
    """value_decoder<T>: reset_dict (optional), reset(page, encoding), one read_batch per entry of `batches`.
    Returns (values, batch_counts): a numpy array for fixed-width types, a list of bytes for the string types."""
    pg, pn = _arr(page)
    dp, dn = _arr(dict_page if dict_page is not None else b"")
    bs = np.asarray(batches, dtype=np.uint64)
    cap = int(bs.sum()) if cap is None else cap
    counts = np.zeros(len(bs), dtype=np.int64)
    err = C.create_string_buffer(_ERR)
    tl = -1 if type_length is None else type_length
    if ptype in FIXED_DTYPES:
        dt = FIXED_DTYPES[ptype]
        out = np.zeros(max(cap, 1), dtype=dt)
        got = lib().p4sh_value_decode(ptype, encoding, tl, dp.ctypes.data, dn, dict_count, pg.ctypes.data, pn,
                                      bs.ctypes.data, len(bs), out.ctypes.data, cap, None, None, 0, counts.ctypes.data,
                                      err, _ERR)
        if got < 0:
            raise HostError(err.value.decode())
        return out[:got].copy(), counts
    bytes_cap = (64 * pn + 1024) if bytes_cap is None else bytes_cap
    offsets = np.zeros(cap + 2, dtype=np.int64)
    data = np.zeros(max(bytes_cap, 1), dtype=np.uint8)
    got = lib().p4sh_value_decode(ptype, encoding, tl, dp.ctypes.data, dn, dict_count, pg.ctypes.data, pn, bs.ctypes.data,
                                  len(bs), None, cap, offsets.ctypes.data, data.ctypes.data, bytes_cap, counts.ctypes.data,
                                  err, _ERR)
    if got < 0:
        raise HostError(err.value.decode())
    raw = data.tobytes()
    return [raw[offsets[i]:offsets[i + 1]] for i in range(got)], counts


def levels_decode(buf, max_level, num_values, batches, *, v2=False, encoding=3):
    """level_decoder: reset_v1 / reset_v2 then one read_batch<int32_t> per batch. Returns (levels, consumed)."""
    b, n = _arr(buf)
    bs = np.asarray(batches, dtype=np.uint64)
    cap = int(bs.sum())
    out = np.zeros(max(cap, 1), dtype=np.int32)
    consumed = C.c_int64(0)
    err = C.create_string_buffer(_ERR)
    got = lib().p4sh_levels_decode(1 if v2 else 0, encoding, max_level, num_values, b.ctypes.data, n, bs.ctypes.data,
                                   len(bs), out.ctypes.data, cap, C.byref(consumed), err, _ERR)
    if got < 0:
        raise HostError(err.value.decode())
    return out[:got].copy(), consumed.value


class Chunk:
    """One column chunk: column_chunk_reader<T> (read_batch) and column_chunk_decoder<T> (plan/launch/collect)."""

    COUNT_NAMES = ("pages", "data_pages", "levels", "values", "failed_pages", "value_slots", "string_bytes")

    def __init__(self, ptype, chunk, max_def, max_rep, type_length=None, codec=0):
        """codec: 0 UNCOMPRESSED, 1 SNAPPY (format::CompressionCodec)."""
        self.ptype = ptype
        self.max_def = max_def
        self._buf, n = _arr(chunk)
        err = C.create_string_buffer(_ERR)
        self.h = lib().p4sh_chunk_open_codec(ptype, self._buf.ctypes.data, n, max_def, max_rep,
                                             -1 if type_length is None else type_length, codec, err, _ERR)
        if not self.h:
            raise HostError(err.value.decode())

    def close(self):
        if self.h:
            lib().p4sh_chunk_close(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def read_batch(self, n, bytes_cap=1 << 20):
        """Returns (def levels, rep levels, values) of one batch; values as in value_decode."""
        d = np.zeros(max(n, 1), dtype=np.int32)
        r = np.zeros(max(n, 1), dtype=np.int32)
        nv = C.c_uint64(0)
        err = C.create_string_buffer(_ERR)
        if self.ptype in FIXED_DTYPES:
            v = np.zeros(max(n, 1), dtype=FIXED_DTYPES[self.ptype])
            got = lib().p4sh_chunk_read_batch(self.h, n, d.ctypes.data, r.ctypes.data, v.ctypes.data, None, None, 0,
                                              C.byref(nv), err, _ERR)
            if got < 0:
                raise HostError(err.value.decode())
            return d[:got].copy(), r[:got].copy(), v[:nv.value].copy()
        off = np.zeros(n + 2, dtype=np.int64)
        data = np.zeros(bytes_cap, dtype=np.uint8)
        got = lib().p4sh_chunk_read_batch(self.h, n, d.ctypes.data, r.ctypes.data, None, off.ctypes.data, data.ctypes.data,
                                          bytes_cap, C.byref(nv), err, _ERR)
        if got < 0:
            raise HostError(err.value.decode())
        raw = data.tobytes()
        return d[:got].copy(), r[:got].copy(), [raw[off[i]:off[i + 1]] for i in range(nv.value)]

    def _call(self, fn, *args):
        err = C.create_string_buffer(_ERR)
        if fn(self.h, *args, err, _ERR) != 0:
            raise HostError(err.value.decode())

    def plan(self):
        self._call(lib().p4sh_chunk_plan)

    def launch(self):
        self._call(lib().p4sh_chunk_launch)

    def collect(self):
        counts = np.zeros(7, dtype=np.uint64)
        self._call(lib().p4sh_chunk_collect, counts.ctypes.data)
        return dict(zip(self.COUNT_NAMES, (int(x) for x in counts)))

    def positional(self, cap_rows):
        """column_chunk_decoder<T>::positional(): (values with one slot per level, nulls 0; validity as bools; nulls)."""
        v = np.zeros(max(cap_rows, 1), dtype=FIXED_DTYPES[self.ptype])
        ok = np.zeros(max(cap_rows, 1), dtype=np.uint8)
        nulls = C.c_uint64(0)
        err = C.create_string_buffer(_ERR)
        rows = lib().p4sh_chunk_positional(self.h, v.ctypes.data, ok.ctypes.data, cap_rows, C.addressof(nulls), err, _ERR)
        if rows < 0:
            raise HostError(err.value.decode())
        return v[:rows].copy(), ok[:rows].astype(bool), int(nulls.value)

    def page_error(self, i):
        buf = C.create_string_buffer(_ERR)
        lib().p4sh_chunk_page_error(self.h, i, buf, _ERR)
        return buf.value.decode()

    def device_ptr(self, which):
        return lib().p4sh_chunk_device_ptr(self.h, {"def": 0, "rep": 1, "values": 2, "offsets": 3, "bytes": 4}[which])


def flat_records(fields, text_cap=1 << 26):
    """flat_record_reader::read_all over `fields` = [(name, physical type, chunk bytes, max_def)]: (records as text, one
    per line: values or "null" joined by '|'; record count; error text or None - the records before an error are
    still returned)."""
    h = lib().p4sh_flat_open()
    err = C.create_string_buffer(_ERR)
    keep = []
    try:
        for name, ptype, chunk, max_def in fields:
            buf, n = _arr(chunk)
            keep.append(buf)
            if lib().p4sh_flat_add_field(h, ptype, name.encode(), buf.ctypes.data, n, max_def, err, _ERR) != 0:
                raise HostError(err.value.decode())
        out = C.create_string_buffer(text_cap)
        records = C.c_uint64(0)
        got = lib().p4sh_flat_read_all(h, out, text_cap, C.addressof(records), err, _ERR)
        if got < 0:
            return out.raw[:-1 - got].decode(), None, err.value.decode()
        return out.raw[:got].decode(), int(records.value), None
    finally:
        lib().p4sh_flat_close(h)


def current_ctx():
    return lib().p4sh_current_ctx()


def page_index(chunk, max_pages=None):
    """The host layer's own header walk (page_reader::next_page) over a chunk, in the records the device walk writes:
    (PAGE_INFO_DTYPE array, error text or None)."""
    from . import capi
    raw = np.frombuffer(bytes(chunk), dtype=np.uint8) if len(chunk) else np.zeros(0, dtype=np.uint8)
    cap = int(max_pages) if max_pages is not None else len(chunk) // 8 + 1
    out = np.zeros(max(cap, 1), dtype=capi.PAGE_INFO_DTYPE)
    count = C.c_uint64(0)
    err = C.create_string_buffer(_ERR)
    rc = lib().p4sh_page_index(raw.ctypes.data if raw.size else None, raw.size, out.ctypes.data, cap, C.byref(count), err, _ERR)
    return out[:count.value].copy(), (err.value.decode("utf-8", "replace") if rc else None)
