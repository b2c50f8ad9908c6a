#!/usr/bin/env python3
"""Prints the per-page cycle counters a -DP4S_EXP_TIMING build of the DELTA kernel leaves in the status words
(aux0 / aux1, two 32-bit sums each), averaged over the pages of the headline workload, next to the launch time.
usage: P4S_GPU_LIB=exp_build/lib_x.so tools/delta_ticks.py [pages]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet4seastar_amd import capi                         # noqa: E402
from parquet4seastar_amd.pagegen import synth               # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n = 131072
    pages, _ = synth.delta_pages(8, n, 3)
    L = capi.lib()
    with capi.Context(0) as ctx:
        use = [pages[i % 8] for i in range(k)]
        buf, offs, sizes = capi.pack_pages(use, align=16)
        descs = capi.make_descs(offs, sizes, [i * n for i in range(k)], n)
        d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
        d_status, d_out = ctx.alloc(48 * k), ctx.alloc(8 * n * k)

        def step():
            ctx._check(L.p4s_decode_delta_binary_packed(ctx.handle, d_in.ptr, buf.size, d_pages.ptr, k, d_out.ptr,
                                                        d_status.ptr, 8))
        step(); ctx.sync()
        ms = C.c_float()
        L.p4s_timer_start(ctx.handle)
        for _ in range(10):
            step()
        L.p4s_timer_stop(ctx.handle, C.byref(ms))
        st = d_status.download(capi.PAGE_STATUS_DTYPE, k)
        names = st.dtype.names
        a0 = st[[x for x in names if "aux0" in x or x == "aux0"][0]].astype(np.uint64)
        a1 = st[[x for x in names if "aux1" in x or x == "aux1"][0]].astype(np.uint64)
        f = [float(np.mean(a0 & np.uint64(0xFFFFFFFF))), float(np.mean(a0 >> np.uint64(32))),
             float(np.mean(a1 & np.uint64(0xFFFFFFFF))), float(np.mean(a1 >> np.uint64(32)))]
        print("%s pages=%d %.4f ms | aux0.lo %.0f aux0.hi %.0f aux1.lo %.0f aux1.hi %.0f" % (
            os.environ.get("P4S_GPU_LIB", "default"), k, ms.value / 10, *f), flush=True)


if __name__ == "__main__":
    main()
