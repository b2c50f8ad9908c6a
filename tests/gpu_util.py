"""Helpers for the GPU parity tests: run one C-ABI decode call over a list of pages
and hand back per-page outputs, with guard bytes around every page's output slot."""
import ctypes as C

import numpy as np

from parquet4seastar_amd import capi

GUARD = 0xCD

_ctx = None


def ctx():
    global _ctx
    if _ctx is None:
        _ctx = capi.Context(0)
    return _ctx


def run_fixed(fn_name, pages, num_values, elem_size, caps, *, param0=0, param1=0, flags=0, extra=(),
              align=1, lead=0, dict_args=None, out_gap=0):
    """Decode `pages` (list of bytes) with a fixed-width-output entry point.
    caps[i] = elements reserved for page i (>= what the page can produce).
    Returns (list of uint8 arrays holding each page's produced bytes, status array)."""
    c = ctx()
    L = capi.lib()
    buf, offs, sizes = capi.pack_pages(pages, align=align, lead=lead)
    caps = [int(x) for x in caps]
    out_offs, pos = [], 0
    for cp in caps:
        out_offs.append(pos)
        pos += cp + out_gap
    total = pos + 8
    descs = capi.make_descs(offs, sizes, out_offs, num_values, param0, param1, flags)
    d_in = c.to_device(buf) if buf.size else c.alloc(16)
    d_pages = c.to_device(descs.view(np.uint8))
    d_status = c.alloc(48 * len(pages))
    d_status.fill(0xFF)
    d_out = c.alloc(total * elem_size)
    d_out.fill(GUARD)
    fn = getattr(L, fn_name)
    if dict_args is not None:
        d_dict, dict_size = dict_args
        rc = fn(c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages), d_dict.ptr, dict_size, elem_size, d_out.ptr,
                d_status.ptr)
    else:
        rc = fn(c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages), d_out.ptr, d_status.ptr, *extra)
    c._check(rc)
    c.sync()
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
    for b in (d_in, d_pages, d_status, d_out):
        b.free()
    return outs, status


def run_strings(kind, pages, num_values, value_caps, byte_caps, *, flba_len=None, align=1, lead=0):
    """Decode string pages. kind in {"dlba", "dba", "dba_sizes", "plain_ba", "flba"}.
    value_caps[i] / byte_caps[i] = value slots / byte capacity reserved for page i.
    Returns (list of (offsets int64 array, bytes uint8 array), status)."""
    c = ctx()
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
    d_in = c.to_device(buf) if buf.size else c.alloc(16)
    d_pages = c.to_device(descs.view(np.uint8))
    d_status = c.alloc(48 * len(pages))
    d_status.fill(0xFF)
    d_offsets = c.alloc(n_slots * 8)
    d_offsets.fill(GUARD)
    d_bytes = c.alloc(n_bytes)
    d_bytes.fill(GUARD)
    d_scratch = c.alloc(n_slots * 8)
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
    for b in (d_in, d_pages, d_status, d_offsets, d_bytes, d_scratch):
        b.free()
    return outs, status
