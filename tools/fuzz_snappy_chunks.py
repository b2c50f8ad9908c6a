#!/usr/bin/env python3
"""Robustness run for SNAPPY chunks through the host layer: chunks with damaged bytes (headers and compressed
bodies) are read to their end; every failure must arrive as a HostError with a text, never as a crash or a hang.
Undamaged chunks must still decode exactly.

    python3 tools/fuzz_snappy_chunks.py N_CHUNKS

With P4S_FUZZ_LOG=FILE every outcome (error texts, digests of the values read) is written to FILE, so that two runs
(P4S_SNAPPY_HOST_IMAGE=0 and =1: pages kept on the device / chunk rebuilt on the host) can be compared line by line.
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from parquet4seastar_amd import hostapi, pagegen as g      # noqa: E402
from tests.test_gpu_host import _snappy                     # noqa: E402

# CODEC=0: the same chunks UNCOMPRESSED (compare P4S_DEVICE_CHUNK=0 and =1: host walk / chunk decoded from device memory)
codec = int(os.environ.get("CODEC", "1"))
z = _snappy() if codec == 1 else None
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
n_chunks = int(sys.argv[1])
errors = ok = 0
log = []
for it in range(n_chunks):
    n = int(rng.choice([4, 300, 3000]))
    dl = (rng.random(n) < 0.7).astype(np.uint32)
    nn = int(dl.sum())
    v = rng.integers(-1000, 1000, size=nn).cumsum().astype(np.int64)
    kw = dict(def_levels=dl, max_def=1)
    ckw = dict(compress=z) if z else {}
    chunk = (g.data_page_v1(v.tobytes(), n, g.ENC_PLAIN, **ckw, **kw)
             + g.data_page_v2(g.delta_encode(v), n, g.ENC_DELTA_BINARY_PACKED, num_nulls=n - nn, **ckw, **kw)
             + g.data_page_v1(v.tobytes(), n, g.ENC_PLAIN, **ckw, **kw))
    damaged = it % 5 != 0
    b = bytearray(chunk)
    if damaged:
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(b)))
            k = int(rng.integers(0, 3))
            if k == 0:
                b[pos] = int(rng.integers(0, 256))
            elif k == 1:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            else:
                del b[pos:pos + int(rng.integers(1, 10))]
    got = []
    try:
        with hostapi.Chunk(g.TYPE_INT64, bytes(b), 1, 0, codec=codec) as c:
            for _ in range(64):                # every page, failed ones included
                try:
                    d, r, vals = c.read_batch(100000)
                except hostapi.HostError as e:
                    assert str(e)
                    errors += 1
                    log.append("%d E %s" % (it, e))
                    continue
                if len(d) == 0:
                    break
                got.append(vals)
                log.append("%d V %d %s %s" % (it, len(d), hashlib.sha1(np.ascontiguousarray(d).tobytes()).hexdigest()[:12],
                                             hashlib.sha1(np.ascontiguousarray(vals).tobytes()).hexdigest()[:12]))
    except hostapi.HostError as e:
        assert str(e)
        errors += 1
        log.append("%d X %s" % (it, e))
    if not damaged:
        assert len(got) == 3 and all(np.array_equal(x, v) for x in got), it
        ok += 1
    if it % 50 == 49:
        print("chunk %d: %d intact chunks exact, %d errors reported" % (it + 1, ok, errors), flush=True)
print("done: %d intact chunks exact, %d errors reported" % (ok, errors))
if os.environ.get("P4S_FUZZ_LOG"):
    with open(os.environ["P4S_FUZZ_LOG"], "w") as f:
        f.write("\n".join(log) + "\n")
