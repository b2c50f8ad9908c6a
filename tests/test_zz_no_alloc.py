"""p4s_decode_delta_byte_array_static, the entry point for callers that cannot allocate: through the C ABI, its
offsets, bytes and statuses must equal those of p4s_decode_delta_byte_array on the same pages, on two calls in a
row (the second on a smaller batch, so the static scratch still holds the first call's contents), and a batch
that spans more slots than the static array is refused."""
import numpy as np
import pytest

from parquet4seastar_amd import capi
from parquet4seastar_amd.pagegen import synth
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


def _decode(ctx, pages, n, totals, static):
    L = capi.lib()
    buf, offs, sizes = capi.pack_pages(pages, align=16)
    n_pages = len(pages)
    caps = [(t + 15) // 16 * 16 for t in totals]
    aux = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
    n_slots = n_pages * (n + 1)
    descs = capi.make_descs(offs, sizes, [i * (n + 1) for i in range(n_pages)], n, caps, aux_offsets=aux)
    d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
    d_status = ctx.alloc(48 * n_pages)
    d_offsets = ctx.alloc(n_slots * 8)
    bytes_cap = int(sum(caps)) + 16
    d_bytes = ctx.alloc(bytes_cap)
    for b in (d_status, d_offsets, d_bytes):
        b.fill(0xCD)
    common = (ctx.handle, d_in.ptr, buf.size, d_pages.ptr, n_pages)
    if static:
        rc = L.p4s_decode_delta_byte_array_static(*common, d_offsets.ptr, n_slots, d_bytes.ptr, bytes_cap, d_status.ptr)
    else:
        d_scratch = ctx.alloc(n_slots * 8)
        rc = L.p4s_decode_delta_byte_array(*common, d_offsets.ptr, d_bytes.ptr, bytes_cap, d_scratch.ptr, n_slots * 8,
                                           d_status.ptr)
    ctx._check(rc)
    ctx.sync()
    out = (d_status.download(capi.PAGE_STATUS_DTYPE, n_pages), d_offsets.download(np.int64, n_slots),
           d_bytes.download(np.uint8, bytes_cap))
    for b in (d_in, d_pages, d_status, d_offsets, d_bytes):
        b.free()
    if not static:
        d_scratch.free()
    return out


def test_static_scratch_entry_point_matches_the_workspace_one():
    ctx = gu.ctx()
    rng = np.random.default_rng(707)
    # first call: 6 pages of 20000 strings; second call: 2 pages of 3000 (smaller, scratch holds the first call's data)
    for n, n_pages in ((20000, 6), (3000, 2)):
        sp = synth.string_pages(rng, n, n_pages)
        pages = [p["dba"] for p in sp]
        totals = [int(p["offsets"][-1]) for p in sp]
        # one page that fails (prefix longer than the string before it) rides along: statuses must agree too
        bad = bytes([0x80, 0x01, 0x04, 0x01, 0x06]) + bytes([0x80, 0x01, 0x04, 0x01, 0x02]) + b"x"
        pages.append(bad + b"\x00" * 8)
        totals.append(64)
        st_s, off_s, by_s = _decode(ctx, pages, n, totals, static=True)
        st_w, off_w, by_w = _decode(ctx, pages, n, totals, static=False)
        assert np.array_equal(st_s, st_w)
        assert st_s["code"][:-1].tolist() == [0] * n_pages and st_s["code"][-1] != 0
        pos = 0
        for i, p in enumerate(sp):
            o = off_s[i * (n + 1):(i + 1) * (n + 1)]
            assert np.array_equal(o, off_w[i * (n + 1):(i + 1) * (n + 1)]) and np.array_equal(o, p["offsets"])
            nb = int(o[-1])
            assert np.array_equal(by_s[pos:pos + nb], by_w[pos:pos + nb]) and np.array_equal(by_s[pos:pos + nb], p["bytes"])
            pos += (totals[i] + 15) // 16 * 16


def _free_device_memory():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_scratch_entry_point_allocates_nothing_and_keeps_to_its_slots():
    """Free device memory is the same before and after a call, and a page whose closing offset would fall behind the
    declared offsets slots (the caller forgot the + 1) reports P4S_E_OUT_CAPACITY instead of writing there."""
    ctx = gu.ctx()
    L = capi.lib()
    rng = np.random.default_rng(708)
    n, n_pages = 5000, 3
    sp = synth.string_pages(rng, n, n_pages)
    pages = [p["dba"] for p in sp]
    totals = [int(p["offsets"][-1]) for p in sp]
    buf, offs, sizes = capi.pack_pages(pages, align=16)
    caps = [(t + 15) // 16 * 16 for t in totals]
    aux = np.concatenate([[0], np.cumsum(caps)[:-1]]).astype(np.uint64)
    n_slots = n_pages * (n + 1)
    descs = capi.make_descs(offs, sizes, [i * (n + 1) for i in range(n_pages)], n, caps, aux_offsets=aux)
    d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
    d_status, d_offsets, d_bytes = ctx.alloc(48 * n_pages), ctx.alloc((n_slots + 8) * 8), ctx.alloc(int(sum(caps)) + 16)
    d_offsets.fill(0xCD)
    ctx.sync()
    before = _free_device_memory()
    for declared in (n_slots, n_slots - 1):
        ctx._check(L.p4s_decode_delta_byte_array_static(ctx.handle, d_in.ptr, buf.size, d_pages.ptr, n_pages, d_offsets.ptr,
                                                        declared, d_bytes.ptr, int(sum(caps)) + 16, d_status.ptr))
        ctx.sync()
        assert _free_device_memory() == before
        st = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
        tail = d_offsets.download(np.int64, n_slots + 8)[declared:]
        if declared == n_slots:
            assert st["code"].tolist() == [0, 0, 0]
        else:
            assert st["code"].tolist() == [0, 0, capi.E_OUT_CAPACITY if hasattr(capi, "E_OUT_CAPACITY") else 23]
        assert (tail.view(np.uint8) == 0xCD).all()          # nothing written at or behind the declared end
        d_offsets.fill(0xCD)
    for b in (d_in, d_pages, d_status, d_offsets, d_bytes):
        b.free()


def test_static_scratch_entry_point_refuses_oversized_batches():
    ctx = gu.ctx()
    L = capi.lib()
    d = ctx.alloc(4096)
    rc = L.p4s_decode_delta_byte_array_static(ctx.handle, d.ptr, 4096, d.ptr, 1, d.ptr, (2 << 20) + 1, d.ptr, 16, d.ptr)
    assert rc == -1 and b"P4S_STATIC_SCRATCH_SLOTS" in L.p4s_last_error(ctx.handle)
    # ... and batches of more pages than its per-page records
    rc = L.p4s_decode_delta_byte_array_static(ctx.handle, d.ptr, 4096, d.ptr, (1 << 15) + 1, d.ptr, 1 << 20, d.ptr, 16, d.ptr)
    assert rc == -1 and b"P4S_STATIC_SCRATCH_PAGES" in L.p4s_last_error(ctx.handle)
    d.free()
