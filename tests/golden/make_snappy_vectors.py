#!/usr/bin/env python3
"""Writes tests/golden/snappy_vectors.json: raw Snappy streams made by libsnappy (the library the reference's
snappy_compressor wraps, src/compression.cc:49-77; here the copy bundled with pyarrow) with the size and SHA-256 of
what they decompress to, and damaged variants with libsnappy's verdict on them.  The reference tree holds no Snappy
fixture of its own (its parquet-testing data directory is empty), so these pin the restatement in oracle/ against
the library itself.

    python3 tests/golden/make_snappy_vectors.py
"""
import hashlib
import json
import os

import numpy as np
import pyarrow as pa

codec = pa.Codec("snappy")
rng = np.random.default_rng(20261111)
vectors = []


def add(name, data, damaged=6):
    comp = codec.compress(data, asbytes=True)
    vectors.append({"name": name, "input": comp.hex(), "size": len(data), "sha256": hashlib.sha256(data).hexdigest()})
    # damaged variants: one byte changed / the stream cut short; libsnappy says whether they still decompress
    for k in range(damaged):
        b = bytearray(comp)
        if len(b) < 3:
            break
        if k < 4:
            pos = int(rng.integers(1, len(b)))
            b[pos] = int(rng.integers(0, 256))
        else:
            del b[int(rng.integers(1, len(b))):]
        try:
            out = codec.decompress(bytes(b), decompressed_size=len(data), asbytes=True)
            vectors.append({"name": "%s/damaged%d" % (name, k), "input": bytes(b).hex(), "size": len(data),
                            "sha256": hashlib.sha256(out).hexdigest()})
        except Exception:
            vectors.append({"name": "%s/damaged%d" % (name, k), "input": bytes(b).hex(), "size": len(data),
                            "sha256": None})


add("empty", b"")
add("one_byte", b"x")
add("run", b"a" * 1000)
add("pattern3", b"abc" * 700)
add("text", b"http://www.example.com/index.html?page=" * 40 + b"tail")
add("random", rng.integers(0, 256, size=3000, dtype=np.uint8).tobytes())
add("low_entropy", rng.integers(0, 4, size=5000, dtype=np.uint8).tobytes())
add("int64_steps", np.cumsum(rng.integers(-3, 4, size=700)).astype(np.int64).tobytes())
add("long_literal", rng.integers(0, 256, size=66000, dtype=np.uint8).tobytes(), damaged=0)   # 3-byte literal length
add("far_copy", rng.integers(0, 256, size=20000, dtype=np.uint8).tobytes() * 2, damaged=2)   # offsets near 20000
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "snappy_vectors.json")
json.dump({"pyarrow": pa.__version__, "vectors": vectors}, open(out, "w"))
print(len(vectors), "vectors,", os.path.getsize(out), "bytes")
