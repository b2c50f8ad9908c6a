#!/usr/bin/env python3
"""Times p4s_decode_rle / p4s_decode_delta_binary_packed over a page-count sweep (device time, HIP events):
how the kernels behave when few pages are in flight.  usage: tools/rle_probe.py [widths] [page counts] [structures]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from parquet4seastar_amd import capi, pagegen as g          # noqa: E402
from parquet4seastar_amd.pagegen import synth               # noqa: E402

MI = 1 << 20


def time_call(ctx, fn, steps=5):
    L = capi.lib()
    fn()
    ctx.sync()
    ms = C.c_float()
    L.p4s_timer_start(ctx.handle)
    for _ in range(steps):
        fn()
    L.p4s_timer_stop(ctx.handle, C.byref(ms))
    return ms.value / steps


def main():
    widths = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,8,12,16,20").split(",")]
    counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,8,64,256,2048").split(",")]
    structs = (sys.argv[3] if len(sys.argv) > 3 else "alt,lit,rep").split(",")
    rng = np.random.default_rng(5)
    L = capi.lib()
    with capi.Context(0) as ctx:
        for st in structs:
            for w in widths:
                distinct = 8
                if st == "alt":
                    vals = [synth.run_values(rng, MI, w) for _ in range(distinct)]
                elif st == "nulls":      # a definition-level stream: 10 % zeros at random places, the rest at the maximum
                    vals = [np.where(rng.random(MI) >= 0.1, (1 << w) - 1, 0).astype(np.uint32) for _ in range(distinct)]
                elif st == "lit":
                    vals = [rng.integers(0, 1 << w, size=MI, dtype=np.uint64).astype(np.uint32) for _ in range(distinct)]
                else:
                    vals = [np.repeat(rng.integers(0, 1 << w, size=MI // 1024 + 1, dtype=np.uint64).astype(np.uint32), 1024)[:MI]
                            for _ in range(distinct)]
                pages = [g.rle_encode(v, w) for v in vals]
                row = []
                for n_pages in counts:
                    reps = (n_pages + distinct - 1) // distinct
                    use = [pages[i % distinct] for i in range(n_pages)]
                    buf, offs, sizes = capi.pack_pages(use, align=16)
                    pad = int(os.environ.get("P4S_PROBE_OUTPAD", "0"))      # values between the pages' output slots
                    stride = MI + pad
                    levels = bool(os.environ.get("P4S_PROBE_LEVELS"))     # the same streams as v2 level sections
                    if levels:
                        descs = capi.make_descs(offs, sizes, [i * stride for i in range(n_pages)], MI, (1 << w) - 1, MI, 2)
                    else:
                        descs = capi.make_descs(offs, sizes, [i * stride for i in range(n_pages)], MI, w)
                    d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
                    d_status, d_out = ctx.alloc(48 * n_pages), ctx.alloc(4 * stride * n_pages)

                    def step():
                        fn = L.p4s_decode_levels if levels else L.p4s_decode_rle
                        ctx._check(fn(ctx.handle, d_in.ptr, buf.size, d_pages.ptr, n_pages, d_out.ptr, d_status.ptr, 4))
                    ms = time_call(ctx, step)
                    got = d_out.download(np.uint32, MI, byte_offset=4 * stride * (n_pages - 1))
                    if os.environ.get("P4S_PROBE_AUX"):      # -DP4S_EXP_TIMING builds: cycle counts in the status words
                        stt = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
                        a = [stt[k].astype(np.uint64) for k in ("aux0", "aux1", "bytes_consumed")]
                        print("   cycles/page: walk %.0f expand %.0f barrierA %.0f stageB %.0f other %.0f" % (
                            (a[0] & 0xFFFFFFFF).mean(), (a[0] >> 32).mean(), (a[1] & 0xFFFFFFFF).mean(), (a[1] >> 32).mean(),
                            a[2].mean()))
                        print("   counts/page: iterations %.0f fallbacks %.0f rounds %.0f" % (
                            (a[2] & 0xFFFFFF).mean(), ((a[2] >> 24) & 0xFFFFF).mean(), (a[2] >> 44).mean()))
                    ok = np.array_equal(got, vals[(n_pages - 1) % distinct])
                    alg = sum(sizes) + 4 * MI * n_pages
                    row.append("%d:%.3fms %.0fGB/s %.2fGv/s%s" % (n_pages, ms, alg / ms / 1e6, MI * n_pages / ms / 1e6,
                                                                  "" if ok else " WRONG"))
                    for b in (d_in, d_pages, d_status, d_out):
                        b.free()
                print("rle %s w=%d enc=%dKB | %s" % (st, w, len(pages[0]) >> 10, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
