#!/usr/bin/env python3
"""Times column_chunk_decoder over one SNAPPY chunk (plan + launch + collect: header walk, page decompression, level
and value decode; results stay on the device) with the decompressed pages kept on the device (default) and with
the uncompressed chunk rebuilt on the host (P4S_SNAPPY_HOST_IMAGE=1).
usage: tools/snappy_chunk_probe.py [pages] [values per page]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parquet4seastar_amd import hostapi, pagegen as g      # noqa: E402
from tests.test_gpu_host import _snappy                     # noqa: E402


def main():
    pages = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
    z = _snappy()
    rng = np.random.default_rng(5)
    dl = (rng.random(n) < 0.9).astype(np.uint32)
    nn = int(dl.sum())
    v = rng.integers(0, 50, size=nn).cumsum().astype(np.int64)
    page = g.data_page_v1(v.tobytes(), n, g.ENC_PLAIN, def_levels=dl, max_def=1, compress=z)
    chunk = page * pages
    raw = pages * (nn * 8 + n // 8)
    for mode in ("0", "1", "0", "1"):
        os.environ["P4S_SNAPPY_HOST_IMAGE"] = mode
        t0 = time.perf_counter()
        with hostapi.Chunk(g.TYPE_INT64, chunk, 1, 0, codec=1) as c:
            c.plan(); c.launch()
            counts = c.collect()
        dt = time.perf_counter() - t0
        assert counts["values"] == pages * nn and counts["failed_pages"] == 0
        print("host image" if mode == "1" else "on device ", "%d pages, %.1f MB compressed, %.1f MB uncompressed: %.1f ms" % (
            pages, len(chunk) / 1e6, raw / 1e6, dt * 1e3), flush=True)


if __name__ == "__main__":
    main()
