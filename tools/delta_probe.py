#!/usr/bin/env python3
"""Times p4s_decode_delta_binary_packed over a page-count sweep (device time, HIP events), with and without the
split of pages over several workgroups (P4S_DELTA_SPLIT is read once per process: run twice).
usage: tools/delta_probe.py [page counts]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet4seastar_amd import capi                         # noqa: E402
from parquet4seastar_amd.pagegen import synth               # noqa: E402


def main():
    counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,8,64,256,512,1024,2048").split(",")]
    n = 131072
    pages, values = synth.delta_pages(8, n, 3)
    L = capi.lib()
    with capi.Context(0) as ctx:
        row = []
        for k in counts:
            use = [pages[i % 8] for i in range(k)]
            buf, offs, sizes = capi.pack_pages(use, align=16)
            descs = capi.make_descs(offs, sizes, [i * n for i in range(k)], n)
            d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
            d_status, d_out = ctx.alloc(48 * k), ctx.alloc(8 * n * k)

            def step():
                ctx._check(L.p4s_decode_delta_binary_packed(ctx.handle, d_in.ptr, buf.size, d_pages.ptr, k, d_out.ptr, d_status.ptr, 8))
            step(); ctx.sync()
            ms = C.c_float()
            L.p4s_timer_start(ctx.handle)
            for _ in range(10):
                step()
            L.p4s_timer_stop(ctx.handle, C.byref(ms))
            t = ms.value / 10
            got = d_out.download(np.int64, n, byte_offset=8 * n * (k - 1))
            ok = np.array_equal(got, values[((k - 1) % 8) * n:((k - 1) % 8 + 1) * n])
            row.append("%d:%.3fms %.0fGB/s%s" % (k, t, (sum(sizes) + 8 * n * k) / t / 1e6, "" if ok else " WRONG"))
            for b in (d_in, d_pages, d_status, d_out):
                b.free()
        print("delta int64 split<=%s | %s" % (os.environ.get("P4S_DELTA_SPLIT", "64"), " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
