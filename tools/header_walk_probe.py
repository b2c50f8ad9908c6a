#!/usr/bin/env python3
"""Times p4s_walk_page_headers (device time, HIP events): one chunk of many pages (one wavefront walks it, a page per
dependent trip to memory) and many chunks of few pages (one wavefront each).  usage: tools/header_walk_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet4seastar_amd import capi, pagegen as g          # noqa: E402


def main():
    rng = np.random.default_rng(3)
    L = capi.lib()
    body = rng.integers(0, 256, size=int(os.environ.get("P4S_PROBE_BODY", 128 * 1024)), dtype=np.uint8).tobytes()
    page = bytes(g.data_page_v1(body, 32768, g.ENC_PLAIN))
    with capi.Context(0) as ctx:
        for n_chunks, pages_per_chunk in ((1, 2048), (16, 128), (256, 8), (2048, 1)):
            chunk = page * pages_per_chunk
            buf = np.frombuffer(chunk * n_chunks, dtype=np.uint8)
            cd = np.zeros(n_chunks, dtype=capi.CHUNK_DESC_DTYPE)
            cd["offset"] = np.arange(n_chunks, dtype=np.uint64) * len(chunk)
            cd["size"], cd["max_pages"], cd["flags"] = len(chunk), pages_per_chunk, capi.CHUNK_UNCOMPRESSED
            cd["first_page"] = np.arange(n_chunks, dtype=np.uint64) * pages_per_chunk
            d_in, d_cd = ctx.to_device(buf), ctx.to_device(cd.view(np.uint8))
            d_pages = ctx.alloc(n_chunks * pages_per_chunk * capi.PAGE_INFO_DTYPE.itemsize)
            d_st = ctx.alloc(n_chunks * capi.PAGE_STATUS_DTYPE.itemsize)

            def step():
                ctx._check(L.p4s_walk_page_headers(ctx.handle, d_in.ptr, buf.size, d_cd.ptr, n_chunks, d_pages.ptr, d_st.ptr))
            step(); ctx.sync()
            ms = C.c_float()
            L.p4s_timer_start(ctx.handle)
            for _ in range(5):
                step()
            L.p4s_timer_stop(ctx.handle, C.byref(ms))
            st = d_st.download(capi.PAGE_STATUS_DTYPE, n_chunks)
            assert (st["code"] == 0).all() and (st["values_decoded"] == pages_per_chunk).all()
            t = ms.value / 5
            print("%5d chunks x %5d pages of %d KiB: %.3f ms per launch, %.2f us per page and chain" % (
                n_chunks, pages_per_chunk, len(page) >> 10, t, 1e3 * t / pages_per_chunk), flush=True)
            for b in (d_in, d_cd, d_pages, d_st):
                b.free()


if __name__ == "__main__":
    main()
