"""Helpers for the GPU parity tests: run one C-ABI decode call over a list of pages
and hand back per-page outputs, with guard bytes around every page's output slot.

All device buffers of a call are carved from one arena that lives as long as the process, with a
guard band before and after each of them: the bands are filled with a pattern before the call and
must hold it afterwards, so a kernel that writes outside a buffer fails the test that ran it (also on
pages that end in an error, whose output slots are not compared), and the suite does not depend on
where the allocator happens to place a buffer."""
import ctypes as C

import numpy as np

from parquet4seastar_amd import capi

GUARD = 0xCD
BAND = 0xE7                 # fill of the guard bands between buffers
BAND_BYTES = 64 << 10
ARENA_BYTES = 1 << 30

_ctx = None
_arena = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


class _View(capi.DeviceBuffer):
    """A piece of the arena with the DeviceBuffer interface."""

    def __init__(self, c, ptr, nbytes):
        self.ctx, self.nbytes, self.ptr = c, int(nbytes), ptr

    def free(self):
        self.ptr = None


class Arena:
    def __init__(self, c):
        self.c = c
        self.buf = c.alloc(ARENA_BYTES)
        self.pos = 0
        self.bands = []

    def begin(self):
        self.pos = 0
        self.bands = []
        self._band()

    def _band(self):
        self.bands.append(self.pos)
        self.pos += BAND_BYTES

    def take(self, nbytes):
        """nbytes of device memory, 256-byte aligned, followed by a guard band; None when the arena is full."""
        nbytes = max(int(nbytes), 16)
        if self.pos + nbytes + 256 + BAND_BYTES > ARENA_BYTES:
            return None
        v = _View(self.c, self.buf.ptr + self.pos, nbytes)
        self.pos = (self.pos + nbytes + 255) // 256 * 256
        self._band()
        return v

    def arm(self):
        for b in self.bands:
            _View(self.c, self.buf.ptr + b, BAND_BYTES).fill(BAND)

    def check(self):
        for i, b in enumerate(self.bands):
            got = _View(self.c, self.buf.ptr + b, BAND_BYTES).download(np.uint8, BAND_BYTES)
            bad = np.flatnonzero(got != BAND)
            assert bad.size == 0, "guard band %d was written at byte %d (of %d bytes written)" % (i, bad[0], bad.size)


def arena():
    global _arena
    if _arena is None:
        _arena = Arena(ctx())
    _arena.begin()
    return _arena


_dbg_ranges = []


def _dbg_arm(buffers):
    """With a library built with -DP4S_DEBUG_BOUNDS: tell it the buffers of the call."""
    L = capi.lib()
    if not hasattr(L, "p4s_debug_set_bounds"):
        return
    lo = (C.c_uint64 * 6)(*([0] * 6))
    hi = (C.c_uint64 * 6)(*([0] * 6))
    _dbg_ranges[:] = []
    for i, b in enumerate(buffers):
        lo[i], hi[i] = b.ptr, b.ptr + b.nbytes
        _dbg_ranges.append((b.ptr, b.nbytes))
    assert L.p4s_debug_set_bounds(lo, hi) == 0


def _dbg_check():
    L = capi.lib()
    if not hasattr(L, "p4s_debug_get_hits"):
        return
    out = (C.c_uint64 * 65)()
    assert L.p4s_debug_get_hits(out) == 0
    if out[0]:
        lines = ["%d accesses outside the call's buffers" % out[0]]
        lines += ["  buffer %d: 0x%x + %d" % (i, p, n) for i, (p, n) in enumerate(_dbg_ranges)]
        for k in range(min(int(out[0]), 16)):
            a, ss, wg, t = (int(out[1 + 4 * k + j]) for j in range(4))
            rel = ", ".join("b%d%+d" % (i, a - p) for i, (p, n) in enumerate(_dbg_ranges))
            lines.append("  %d bytes at 0x%x (%s), site %d, page %d, thread %d" % (ss >> 32, a, rel, ss & 0xFFFFFFFF, wg, t))
        raise AssertionError("\n".join(lines))


class _Buffers:
    """The device buffers of one decode call: from the arena when they fit, else plain allocations."""

    def __init__(self):
        self.c = ctx()
        self.ar = arena()
        self.own = []

    def alloc(self, nbytes):
        v = self.ar.take(nbytes) if self.ar is not None else None
        if v is None:
            v = self.c.alloc(max(int(nbytes), 16))
            self.own.append(v)
        return v

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        v = self.alloc(a.nbytes)
        if a.nbytes:
            v.upload(a)
        return v

    def arm(self):
        self.ar.arm()

    def check_and_free(self):
        self.ar.check()
        for b in self.own:
            b.free()


def run_fixed(fn_name, pages, num_values, elem_size, caps, *, param0=0, param1=0, flags=0, extra=(),
              align=1, lead=0, dict_args=None, out_gap=0):
    """Decode `pages` (list of bytes) with a fixed-width-output entry point.
    caps[i] = elements reserved for page i (>= what the page can produce).
    Returns (list of uint8 arrays holding each page's produced bytes, status array)."""
    c = ctx()
    bufs = _Buffers()
    L = capi.lib()
    buf, offs, sizes = capi.pack_pages(pages, align=align, lead=lead)
    caps = [int(x) for x in caps]
    out_offs, pos = [], 0
    for cp in caps:
        out_offs.append(pos)
        pos += cp + out_gap
    total = pos + 8
    descs = capi.make_descs(offs, sizes, out_offs, num_values, param0, param1, flags)
    d_in = bufs.to_device(buf)
    d_pages = bufs.to_device(descs.view(np.uint8))
    d_status = bufs.alloc(48 * len(pages))
    d_status.fill(0xFF)
    d_out = bufs.alloc(total * elem_size)
    d_out.fill(GUARD)
    bufs.arm()
    _dbg_arm([d_in, d_pages, d_status, d_out] + ([dict_args[0]] if dict_args is not None else []))
    fn = getattr(L, fn_name)
    if dict_args is not None:
        d_dict, dict_size = dict_args
        rc = fn(c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages), d_dict.ptr, dict_size, elem_size, d_out.ptr,
                d_status.ptr)
    else:
        rc = fn(c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages), d_out.ptr, d_status.ptr, *extra)
    c._check(rc)
    c.sync()
    _dbg_check()
    status = d_status.download(capi.PAGE_STATUS_DTYPE, len(pages))
    raw = d_out.download(np.uint8, total * elem_size)
    outs = []
    for i, st in enumerate(status):
        lo = out_offs[i] * elem_size
        got = int(st["values_decoded"]) if st["code"] == 0 else 0
        assert got <= caps[i], "page %d decoded %d values into a slot of %d" % (i, got, caps[i])
        outs.append(raw[lo: lo + got * elem_size].copy())
        if st["code"] == 0:
            tail = raw[lo + got * elem_size: lo + (caps[i] + out_gap) * elem_size]
            assert np.all(tail == GUARD), "page %d wrote past its %d decoded values" % (i, got)
    bufs.check_and_free()
    return outs, status


def run_strings(kind, pages, num_values, value_caps, byte_caps, *, flba_len=None, align=1, lead=0):
    """Decode string pages. kind in {"dlba", "dba", "dba_sizes", "plain_ba", "flba"}.
    value_caps[i] / byte_caps[i] = value slots / byte capacity reserved for page i.
    Returns (list of (offsets int64 array, bytes uint8 array), status)."""
    c = ctx()
    bufs = _Buffers()
    L = capi.lib()
    buf, offs, sizes = capi.pack_pages(pages, align=align, lead=lead)
    slot_offs, pos = [], 0
    for cp in value_caps:
        slot_offs.append(pos)
        pos += int(cp) + 1
    n_slots = pos + 1
    byte_offs, bpos = [], 0
    for bc in byte_caps:
        byte_offs.append(bpos)
        bpos += int(bc) + 3
    n_bytes = bpos + 8
    param0 = byte_caps if flba_len is None else flba_len
    descs = capi.make_descs(offs, sizes, slot_offs, num_values, param0, aux_offsets=byte_offs)
    d_in = bufs.to_device(buf)
    d_pages = bufs.to_device(descs.view(np.uint8))
    d_status = bufs.alloc(48 * len(pages))
    d_status.fill(0xFF)
    d_offsets = bufs.alloc(n_slots * 8)
    d_offsets.fill(GUARD)
    d_bytes = bufs.alloc(n_bytes)
    d_bytes.fill(GUARD)
    # the kernels must not depend on what the scratch buffer held before the call
    d_scratch = bufs.alloc(n_slots * 8)
    d_scratch.upload(np.full(n_slots * 2, 41, dtype=np.uint32))
    bufs.arm()
    _dbg_arm([d_in, d_pages, d_status, d_offsets, d_bytes, d_scratch])
    common = (c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages))
    if kind == "dlba":
        rc = L.p4s_decode_delta_length_byte_array(*common, d_offsets.ptr, d_bytes.ptr, n_bytes, d_status.ptr)
    elif kind == "plain_ba":
        rc = L.p4s_decode_plain_byte_array(*common, d_offsets.ptr, d_bytes.ptr, n_bytes, d_status.ptr)
    elif kind == "dba":
        rc = L.p4s_decode_delta_byte_array(*common, d_offsets.ptr, d_bytes.ptr, n_bytes, d_scratch.ptr, n_slots * 8,
                                           d_status.ptr)
    elif kind == "dba_sizes":
        rc = L.p4s_delta_byte_array_sizes(*common, d_scratch.ptr, n_slots * 8, d_status.ptr)
    elif kind == "flba":
        rc = L.p4s_decode_plain_fixed_len_byte_array(*common, d_bytes.ptr, n_bytes, d_status.ptr)
    else:
        raise ValueError(kind)
    c._check(rc)
    c.sync()
    _dbg_check()
    status = d_status.download(capi.PAGE_STATUS_DTYPE, len(pages))
    raw_off = d_offsets.download(np.int64, n_slots)
    raw_bytes = d_bytes.download(np.uint8, n_bytes)
    outs = []
    for i, st in enumerate(status):
        if st["code"] != 0:
            outs.append((None, None))
            continue
        nv, nb = int(st["values_decoded"]), int(st["out_bytes"])
        assert nv <= value_caps[i] and nb <= byte_caps[i], (i, st)
        o = raw_off[slot_offs[i]: slot_offs[i] + nv + 1].copy()
        b = raw_bytes[byte_offs[i]: byte_offs[i] + nb].copy()
        if kind not in ("dba_sizes",):
            assert np.all(raw_bytes[byte_offs[i] + nb: byte_offs[i] + int(byte_caps[i]) + 3] == GUARD), \
                "page %d wrote past its %d bytes: %r %r n=%r" % (
                    i, nb, st, raw_bytes[byte_offs[i]: byte_offs[i] + 24], num_values[i] if hasattr(num_values, "__len__") else num_values)
        outs.append((o, b))
    bufs.check_and_free()
    return outs, status


def run_snappy(pages, sizes, caps, *, align=1, lead=0, out_gap=3):
    """Decompress Snappy `pages`; sizes[i] = uncompressed size the page header would announce (param0),
    caps[i] = bytes reserved for page i in the output.  Returns (list of uint8 arrays, status)."""
    c = ctx()
    bufs = _Buffers()
    L = capi.lib()
    buf, offs, in_sizes = capi.pack_pages(pages, align=align, lead=lead)
    out_offs, pos = [], 0
    for cp in caps:
        out_offs.append(pos)
        pos += int(cp) + out_gap
    total = pos + 8
    descs = capi.make_descs(offs, in_sizes, out_offs, 0, [int(s) for s in sizes])
    d_in = bufs.to_device(buf)
    d_pages = bufs.to_device(descs.view(np.uint8))
    d_status = bufs.alloc(48 * len(pages))
    d_status.fill(0xFF)
    d_out = bufs.alloc(total)
    d_out.fill(GUARD)
    bufs.arm()
    _dbg_arm([d_in, d_pages, d_status, d_out])
    c._check(L.p4s_decompress_snappy(c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages), d_out.ptr, total,
                                     d_status.ptr))
    c.sync()
    _dbg_check()
    status = d_status.download(capi.PAGE_STATUS_DTYPE, len(pages))
    raw = d_out.download(np.uint8, total)
    outs = []
    for i, st in enumerate(status):
        got = int(st["out_bytes"]) if st["code"] == 0 else 0
        assert got <= caps[i], (i, st)
        outs.append(raw[out_offs[i]: out_offs[i] + got].copy())
        # a page never writes past the size its header announces, whether it succeeds or not
        lim = min(int(sizes[i]), int(caps[i])) if st["code"] != 0 else got
        assert np.all(raw[out_offs[i] + lim: out_offs[i] + int(caps[i]) + out_gap] == GUARD), \
            "page %d wrote past byte %d of its slot" % (i, lim)
    bufs.check_and_free()
    return outs, status


def walk_page_headers(chunks, flags=None, max_pages=None, lead=0):
    """p4s_walk_page_headers over `chunks` (list of bytes) with every buffer taken from the guarded arena (and made
    known to a bounds-checked build): ([PAGE_INFO_DTYPE array per chunk], status rows)."""
    c = ctx()
    bufs = _Buffers()
    n = len(chunks)
    buf, offs, sizes = capi.pack_pages(chunks, align=1, lead=lead)
    fl = flags if isinstance(flags, (list, tuple)) else [capi.CHUNK_UNCOMPRESSED if flags is None else flags] * n
    mp = max_pages if isinstance(max_pages, (list, tuple)) else [max_pages] * n
    mp = [int(m) if m is not None else sizes[i] // 8 + 1 for i, m in enumerate(mp)]
    cd = np.zeros(n, dtype=capi.CHUNK_DESC_DTYPE)
    cd["offset"], cd["size"], cd["max_pages"], cd["flags"] = offs, sizes, mp, fl
    cd["first_page"] = np.concatenate([[0], np.cumsum(mp)[:-1]]) if n else []
    total = int(sum(mp))
    d_in = bufs.to_device(buf if buf.size else np.zeros(8, dtype=np.uint8))
    d_cd = bufs.to_device(cd.view(np.uint8) if n else np.zeros(32, dtype=np.uint8))
    d_pages = bufs.alloc(max(total, 1) * capi.PAGE_INFO_DTYPE.itemsize)
    d_pages.fill(GUARD)
    d_status = bufs.alloc(max(n, 1) * 48)
    d_status.fill(0xFF)
    bufs.arm()
    _dbg_arm([d_in, d_cd, d_pages, d_status])
    c._check(capi.lib().p4s_walk_page_headers(c.handle, d_in.ptr, buf.size, d_cd.ptr, n, d_pages.ptr, d_status.ptr))
    c.sync()
    _dbg_check()
    status = d_status.download(capi.PAGE_STATUS_DTYPE, n) if n else np.zeros(0, dtype=capi.PAGE_STATUS_DTYPE)
    infos = d_pages.download(capi.PAGE_INFO_DTYPE, total) if total else np.zeros(0, dtype=capi.PAGE_INFO_DTYPE)
    bufs.check_and_free()
    out = []
    for i in range(n):
        f = int(cd["first_page"][i])
        out.append(infos[f:f + int(status["values_decoded"][i])].copy())
        # records past the chunk's count are untouched
        rest = infos[f + int(status["values_decoded"][i]):f + mp[i]]
        assert (rest.view(np.uint8) == GUARD).all()
    return out, status
