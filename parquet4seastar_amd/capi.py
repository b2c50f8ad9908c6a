"""ctypes binding of the C ABI (include/p4s_gpu.h, built into csrc/libp4s_gpu.so).

This is plumbing for Python callers (tests, bench.py, the smoke check): device
buffers, page batches and status decoding.  There is no CPU fallback: without the
HIP library, or without a GPU, every entry point raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# P4S_GPU_LIB selects another build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("P4S_GPU_LIB") or os.path.join(_HERE, "csrc", "libp4s_gpu.so")

P4S_OK = 0
LEVELS_V1_RLE, LEVELS_V1_BIT_PACKED, LEVELS_V2 = 0, 1, 2


class PageDesc(C.Structure):
    _fields_ = [("in_offset", C.c_uint64), ("in_size", C.c_uint64), ("out_offset", C.c_uint64),
                ("aux_offset", C.c_uint64), ("num_values", C.c_uint32), ("param0", C.c_uint32),
                ("param1", C.c_uint32), ("flags", C.c_uint32)]


class PageStatus(C.Structure):
    _fields_ = [("code", C.c_int32), ("reserved", C.c_uint32), ("values_decoded", C.c_uint64),
                ("aux0", C.c_uint64), ("aux1", C.c_uint64), ("bytes_consumed", C.c_uint64),
                ("out_bytes", C.c_uint64)]


PAGE_DESC_DTYPE = np.dtype([("in_offset", "<u8"), ("in_size", "<u8"), ("out_offset", "<u8"), ("aux_offset", "<u8"),
                            ("num_values", "<u4"), ("param0", "<u4"), ("param1", "<u4"), ("flags", "<u4")])
PAGE_STATUS_DTYPE = np.dtype([("code", "<i4"), ("reserved", "<u4"), ("values_decoded", "<u8"), ("aux0", "<u8"),
                              ("aux1", "<u8"), ("bytes_consumed", "<u8"), ("out_bytes", "<u8")])
assert PAGE_DESC_DTYPE.itemsize == C.sizeof(PageDesc) == 48
assert PAGE_STATUS_DTYPE.itemsize == C.sizeof(PageStatus) == 48

# every symbol include/p4s_gpu.h declares
# p4s_page_info / p4s_chunk_desc of include/p4s_gpu.h (p4s_walk_page_headers)
PAGE_INFO_DTYPE = np.dtype([("header_offset", "<u8"), ("body_offset", "<u8"), ("type", "<i4"),
                            ("uncompressed_page_size", "<i4"), ("compressed_page_size", "<i4"), ("num_values", "<i4"),
                            ("encoding", "<i4"), ("definition_level_encoding", "<i4"), ("repetition_level_encoding", "<i4"),
                            ("definition_levels_byte_length", "<i4"), ("repetition_levels_byte_length", "<i4"),
                            ("num_nulls", "<i4"), ("num_rows", "<i4"), ("flags", "<u4"), ("v1_rep_prefix", "<i4"),
                            ("v1_def_prefix", "<i4"), ("reserved", "<u4", (2,))])
CHUNK_DESC_DTYPE = np.dtype([("offset", "<u8"), ("size", "<u8"), ("first_page", "<u8"), ("max_pages", "<u4"), ("flags", "<u4")])
assert PAGE_INFO_DTYPE.itemsize == 80 and CHUNK_DESC_DTYPE.itemsize == 32
PAGE_HAS_DATA_HEADER, PAGE_HAS_DICT_HEADER, PAGE_HAS_DATA_HEADER_V2, PAGE_IS_COMPRESSED, PAGE_IS_SORTED, PAGE_HAS_CRC = 1, 2, 4, 8, 16, 32
CHUNK_HAS_REP, CHUNK_HAS_DEF, CHUNK_UNCOMPRESSED = 1, 2, 4

EXPORTS = [
    "p4s_version", "p4s_ctx_create", "p4s_ctx_destroy", "p4s_ctx_set_stream", "p4s_ctx_sync", "p4s_last_error",
    "p4s_format_error", "p4s_dev_alloc", "p4s_dev_free", "p4s_host_alloc", "p4s_host_free", "p4s_copy_h2d",
    "p4s_copy_d2h", "p4s_memset", "p4s_timer_start", "p4s_timer_stop",
    "p4s_decode_byte_stream_split", "p4s_decode_plain_fixed", "p4s_decode_plain_boolean",
    "p4s_decode_delta_binary_packed", "p4s_decode_rle", "p4s_decode_levels",
    "p4s_pages_set_num_values_from_levels", "p4s_decode_dictionary",
    "p4s_decode_delta_length_byte_array", "p4s_delta_byte_array_sizes", "p4s_decode_delta_byte_array",
    "p4s_decode_delta_byte_array_static",
    "p4s_decode_plain_byte_array", "p4s_decode_plain_fixed_len_byte_array", "p4s_scatter_by_levels",
    "p4s_decompress_snappy", "p4s_walk_page_headers", "p4s_encode_delta_binary_packed", "p4s_encode_rle",
]

_lib = None


class P4SError(RuntimeError):
    pass


class PageDecodeError(P4SError):
    """A page failed to decode; the text is the reference's exception message."""

    def __init__(self, page, code, message):
        self.page, self.code = page, code
        super().__init__(message)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P4SError("HIP extension missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(expected %s)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        common = [vp, vp, u64, vp, u32]
        sigs = {
            "p4s_version": ([], C.c_char_p),
            "p4s_ctx_create": ([i32, C.POINTER(vp)], i32),
            "p4s_ctx_destroy": ([vp], None),
            "p4s_ctx_set_stream": ([vp, vp], i32),
            "p4s_ctx_sync": ([vp], i32),
            "p4s_last_error": ([vp], C.c_char_p),
            "p4s_format_error": ([vp, C.c_char_p, C.c_size_t], i32),
            "p4s_dev_alloc": ([vp, u64, C.POINTER(vp)], i32),
            "p4s_dev_free": ([vp, vp], i32),
            "p4s_host_alloc": ([vp, u64, C.POINTER(vp)], i32),
            "p4s_host_free": ([vp, vp], i32),
            "p4s_copy_h2d": ([vp, vp, vp, u64], i32),
            "p4s_copy_d2h": ([vp, vp, vp, u64], i32),
            "p4s_memset": ([vp, vp, i32, u64], i32),
            "p4s_timer_start": ([vp], i32),
            "p4s_timer_stop": ([vp, C.POINTER(C.c_float)], i32),
            "p4s_encode_delta_binary_packed": ([vp, vp, u64, vp, u32, vp, u64, vp, i32], i32),
            "p4s_encode_rle": ([vp, vp, u64, vp, u32, vp, u64, vp], i32),
            "p4s_decode_byte_stream_split": (common + [vp, vp, i32], i32),
            "p4s_decode_plain_fixed": (common + [vp, vp, i32], i32),
            "p4s_decode_plain_boolean": (common + [vp, vp], i32),
            "p4s_decode_delta_binary_packed": (common + [vp, vp, i32], i32),
            "p4s_decode_rle": (common + [vp, vp, i32], i32),
            "p4s_decode_levels": (common + [vp, vp, i32], i32),
            "p4s_pages_set_num_values_from_levels": ([vp, vp, vp, u32], i32),
            "p4s_decode_dictionary": (common + [vp, u64, i32, vp, vp], i32),
            "p4s_decode_delta_length_byte_array": (common + [vp, vp, u64, vp], i32),
            "p4s_delta_byte_array_sizes": (common + [vp, u64, vp], i32),
            "p4s_decode_delta_byte_array": (common + [vp, vp, u64, vp, u64, vp], i32),
            "p4s_decode_delta_byte_array_static": (common + [vp, u64, vp, u64, vp], i32),
            "p4s_decode_plain_byte_array": (common + [vp, vp, u64, vp], i32),
            "p4s_decode_plain_fixed_len_byte_array": (common + [vp, u64, vp], i32),
            "p4s_scatter_by_levels": ([vp, vp, vp, vp, u32, i32, vp, vp, vp], i32),
            "p4s_decompress_snappy": (common + [vp, u64, vp], i32),
            "p4s_walk_page_headers": ([vp, vp, u64, vp, u32, vp, vp], i32),
        }
        assert set(sigs) == set(EXPORTS)
        for name, (argtypes, restype) in sigs.items():
            # a symbol the library lacks surfaces as AttributeError at the call site
            # (and in the export test), not here
            if hasattr(L, name):
                f = getattr(L, name)
                f.argtypes, f.restype = argtypes, restype
        _lib = L
    return _lib


def format_error(status_row):
    st = PageStatus(int(status_row["code"]), 0, int(status_row["values_decoded"]), int(status_row["aux0"]),
                    int(status_row["aux1"]), int(status_row["bytes_consumed"]), int(status_row["out_bytes"]))
    buf = C.create_string_buffer(512)
    lib().p4s_format_error(C.byref(st), buf, 512)
    return buf.value.decode()


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        ctx._check(lib().p4s_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr:
            lib().p4s_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.ctx._check(lib().p4s_copy_h2d(self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes))
        self.ctx.sync()   # the source is pageable host memory that may go away

    def download(self, dtype, count, byte_offset=0):
        out = np.empty(count, dtype=dtype)
        assert byte_offset + out.nbytes <= self.nbytes
        self.ctx._check(lib().p4s_copy_d2h(self.ctx.handle, out.ctypes.data, self.ptr + byte_offset, out.nbytes))
        self.ctx.sync()
        return out

    def fill(self, value):
        self.ctx._check(lib().p4s_memset(self.ctx.handle, self.ptr, value, self.nbytes))


class Context:
    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib().p4s_ctx_create(device, C.byref(h))
        if rc != P4S_OK:
            raise P4SError("p4s_ctx_create(device=%d) failed with %d (is a GPU visible?)" % (device, rc))
        self.handle = h.value

    def close(self):
        if self.handle:
            lib().p4s_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != P4S_OK:
            raise P4SError("p4s call failed (%d): %s" % (rc, lib().p4s_last_error(self.handle).decode()))

    def sync(self):
        self._check(lib().p4s_ctx_sync(self.handle))

    def set_stream(self, raw_stream):
        self._check(lib().p4s_ctx_set_stream(self.handle, raw_stream))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        b = DeviceBuffer(self, max(a.nbytes, 16))
        if a.nbytes:
            b.upload(a)
        return b

    def timer_start(self):
        self._check(lib().p4s_timer_start(self.handle))

    def timer_stop(self):
        ms = C.c_float()
        self._check(lib().p4s_timer_stop(self.handle, C.byref(ms)))
        return ms.value


def pack_pages(pages, align=1, lead=0):
    """Concatenate encoded pages into one input buffer; returns (uint8 array, offsets, sizes).
    `lead` bytes of filler precede the first page and each page start is rounded up to `align`."""
    offs, sizes = [], []
    pos = lead
    for pg in pages:
        pos = (pos + align - 1) // align * align
        offs.append(pos)
        sizes.append(len(pg))
        pos += len(pg)
    buf = np.full(pos + 0, 0xA5, dtype=np.uint8)
    for o, pg in zip(offs, pages):
        buf[o:o + len(pg)] = np.frombuffer(bytes(pg), dtype=np.uint8)
    return buf, offs, sizes


def make_descs(offs, sizes, out_offsets, num_values, param0=0, param1=0, flags=0, aux_offsets=0):
    n = len(offs)
    d = np.zeros(n, dtype=PAGE_DESC_DTYPE)
    d["in_offset"], d["in_size"], d["out_offset"] = offs, sizes, out_offsets
    d["aux_offset"] = aux_offsets
    d["num_values"], d["param0"], d["param1"], d["flags"] = num_values, param0, param1, flags
    return d


def delta_encode_bound(n, elem_size):
    """P4S_DELTA_ENCODE_BOUND of include/p4s_gpu.h."""
    return 32 + ((int(n) + 255) // 256) * (18 + 256 * elem_size)


def encode_delta_binary_packed(ctx, values, counts, elem_size=8, caps=None, guard=None):
    """p4s_encode_delta_binary_packed over consecutive runs of `values` (counts[i] values each): ([page bytes], status
    rows).  caps: per-page slot capacities (default: the bound).  guard: a byte the output buffer is filled with first;
    every byte of a slot behind its page must still hold it afterwards (checked here)."""
    v = np.ascontiguousarray(values, dtype=np.int32 if elem_size == 4 else np.int64)
    counts = [int(c) for c in counts]
    n = len(counts)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
    caps = [delta_encode_bound(c, elem_size) for c in counts] if caps is None else [int(c) for c in caps]
    slots = [(c + 15) // 16 * 16 for c in caps]
    out_offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
    descs = make_descs(starts, [c * elem_size for c in counts], out_offs, counts, param0=caps)
    out_bytes = int(sum(slots))
    d_v = ctx.to_device(v.view(np.uint8) if v.size else np.zeros(8, dtype=np.uint8))
    d_desc = ctx.to_device(descs.view(np.uint8))
    d_out, d_st = ctx.alloc(max(out_bytes, 16)), ctx.alloc(max(n, 1) * PAGE_STATUS_DTYPE.itemsize)
    try:
        if guard is not None:
            d_out.fill(guard)
        ctx._check(lib().p4s_encode_delta_binary_packed(ctx.handle, d_v.ptr, v.size, d_desc.ptr, n, d_out.ptr, out_bytes,
                                                        d_st.ptr, elem_size))
        ctx.sync()
        st = d_st.download(PAGE_STATUS_DTYPE, n) if n else np.zeros(0, dtype=PAGE_STATUS_DTYPE)
        raw = d_out.download(np.uint8, out_bytes) if out_bytes else np.zeros(0, dtype=np.uint8)
    finally:
        for b in (d_v, d_desc, d_out, d_st):
            b.free()
    pages = [raw[int(o):int(o) + int(s["out_bytes"])].tobytes() if s["code"] == 0 else None for o, s in zip(out_offs, st)]
    if guard is not None:
        for o, sl, cp, s in zip(out_offs, slots, caps, st):
            used = int(s["out_bytes"]) if s["code"] == 0 else cp          # a page that did not fit may have written up to its capacity
            tail = raw[int(o) + used:int(o) + sl]
            assert np.all(tail == guard), "encoder wrote behind a page's %d bytes" % used
    return pages, st


def rle_encode_bound(n, w):
    """P4S_RLE_ENCODE_BOUND of include/p4s_gpu.h."""
    return ((int(n) + 7) // 8) * (1 + int(w)) + 64 * int(w) + 16


def encode_rle(ctx, values, counts, widths, caps=None, guard=None):
    """p4s_encode_rle over consecutive runs of `values` (uint32; counts[i] values at bit width widths[i] each): ([stream
    bytes or None], status rows).  caps / guard as in encode_delta_binary_packed."""
    v = np.ascontiguousarray(values, dtype=np.uint32)
    counts = [int(c) for c in counts]
    n = len(counts)
    widths = [int(w) for w in (widths if isinstance(widths, (list, tuple, np.ndarray)) else [widths] * n)]
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
    caps = [rle_encode_bound(c, w) for c, w in zip(counts, widths)] if caps is None else [int(c) for c in caps]
    slots = [(c + 15) // 16 * 16 + 16 for c in caps]
    out_offs = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.uint64) if n else np.zeros(0, dtype=np.uint64)
    descs = make_descs(starts, [c * 4 for c in counts], out_offs, counts, param0=caps, param1=widths)
    out_bytes = int(sum(slots))
    d_v = ctx.to_device(v.view(np.uint8) if v.size else np.zeros(8, dtype=np.uint8))
    d_desc = ctx.to_device(descs.view(np.uint8))
    d_out, d_st = ctx.alloc(max(out_bytes, 16)), ctx.alloc(max(n, 1) * PAGE_STATUS_DTYPE.itemsize)
    try:
        if guard is not None:
            d_out.fill(guard)
        L = lib()
        checked = hasattr(L, "p4s_debug_set_bounds")      # a library built with -DP4S_DEBUG_BOUNDS: the buffers of the call
        if checked:
            bufs = (d_v, d_desc, d_out, d_st)
            lo = (C.c_uint64 * 6)(*([b.ptr for b in bufs] + [0, 0]))
            hi = (C.c_uint64 * 6)(*([b.ptr + b.nbytes for b in bufs] + [0, 0]))
            assert L.p4s_debug_set_bounds(lo, hi) == 0
        ctx._check(L.p4s_encode_rle(ctx.handle, d_v.ptr, v.size, d_desc.ptr, n, d_out.ptr, out_bytes, d_st.ptr))
        ctx.sync()
        if checked:
            hits = (C.c_uint64 * 65)()
            assert L.p4s_debug_get_hits(hits) == 0
            assert hits[0] == 0, "%d accesses outside the call's buffers (first: address 0x%x, site %d)" % (
                hits[0], hits[1], hits[2] & 0xFFFFFFFF)
        st = d_st.download(PAGE_STATUS_DTYPE, n) if n else np.zeros(0, dtype=PAGE_STATUS_DTYPE)
        raw = d_out.download(np.uint8, out_bytes) if out_bytes else np.zeros(0, dtype=np.uint8)
    finally:
        for b in (d_v, d_desc, d_out, d_st):
            b.free()
    pages = [raw[int(o):int(o) + int(s["out_bytes"])].tobytes() if s["code"] == 0 else None for o, s in zip(out_offs, st)]
    if guard is not None:
        for o, sl, cp, s in zip(out_offs, slots, caps, st):
            used = int(s["out_bytes"]) if s["code"] == 0 else cp          # a page that did not fit may have written up to its capacity
            tail = raw[int(o) + used:int(o) + sl]
            assert np.all(tail == guard), "encoder wrote behind a page's %d bytes" % used
    return pages, st


def walk_page_headers(ctx, chunks, flags=CHUNK_UNCOMPRESSED, max_pages=None):
    """p4s_walk_page_headers over a list of column chunks (bytes): ([PAGE_INFO_DTYPE array per chunk], status rows).
    `flags` is one value for all chunks or a list; `max_pages` likewise (default: size / 8 + 1 per chunk)."""
    n = len(chunks)
    buf, offs, sizes = pack_pages(chunks, align=8)
    fl = flags if isinstance(flags, (list, tuple)) else [flags] * n
    mp = max_pages if isinstance(max_pages, (list, tuple)) else [max_pages] * n
    mp = [int(m) if m is not None else sizes[i] // 8 + 1 for i, m in enumerate(mp)]
    cd = np.zeros(n, dtype=CHUNK_DESC_DTYPE)
    cd["offset"], cd["size"], cd["max_pages"], cd["flags"] = offs, sizes, mp, fl
    cd["first_page"] = np.concatenate([[0], np.cumsum(mp)[:-1]])
    total = int(sum(mp))
    d_in = ctx.to_device(buf if buf.size else np.zeros(8, dtype=np.uint8))
    d_cd = ctx.to_device(cd.view(np.uint8))
    d_pages = ctx.alloc(max(total, 1) * PAGE_INFO_DTYPE.itemsize)
    d_status = ctx.alloc(max(n, 1) * PAGE_STATUS_DTYPE.itemsize)
    try:
        ctx._check(lib().p4s_walk_page_headers(ctx.handle, d_in.ptr, buf.size, d_cd.ptr, n, d_pages.ptr, d_status.ptr))
        status = d_status.download(PAGE_STATUS_DTYPE, n) if n else np.zeros(0, dtype=PAGE_STATUS_DTYPE)
        infos = d_pages.download(PAGE_INFO_DTYPE, total) if total else np.zeros(0, dtype=PAGE_INFO_DTYPE)
    finally:
        for b in (d_in, d_cd, d_pages, d_status):
            b.free()
    out = []
    for i in range(n):
        f = int(cd["first_page"][i])
        out.append(infos[f:f + int(status["values_decoded"][i])].copy())
    return out, status


def raise_on_page_errors(status):
    for i, st in enumerate(status):
        if st["code"] != 0:
            raise PageDecodeError(i, int(st["code"]), format_error(st))
