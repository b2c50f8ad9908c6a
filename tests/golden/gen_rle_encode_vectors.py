#!/usr/bin/env python3
"""Records what the REFERENCE's own RleEncoder (include/parquet4seastar/rle_encoding.hh, compiled untouched into
oracle/_ref/libp4s_ref_l0.so) writes for a set of inputs: tests/golden/rle_encode_ref_vectors.json.  Run in the build
container, where /root/reference exists:  python3 tests/golden/gen_rle_encode_vectors.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc      # noqa: E402


def shapes(rng, w, n):
    hi = (1 << w) if w < 32 else (1 << 32)
    yield "random", rng.integers(0, hi, size=n, dtype=np.uint64)
    yield "runs_1_19", np.repeat(rng.integers(0, hi, size=n // 5 + 1, dtype=np.uint64), rng.integers(1, 20, size=n // 5 + 1))[:n]
    yield "few_values_runs_1_11", np.repeat(rng.integers(0, min(hi, 3), size=n // 3 + 1, dtype=np.uint64),
                                           rng.integers(1, 12, size=n // 3 + 1))[:n]
    yield "one_value", np.full(n, rng.integers(0, hi), dtype=np.uint64)
    parts = [np.zeros(0, dtype=np.uint64)]
    while sum(len(p) for p in parts) < n:
        parts.append(np.full(int(rng.choice([7, 8, 9, 15, 16, 17, 64, 513])), rng.integers(0, hi), dtype=np.uint64))
        parts.append(rng.integers(0, hi, size=int(rng.integers(0, 20)), dtype=np.uint64))
    yield "runs_7_8_9_between_literals", np.concatenate(parts)[:n]


def main():
    rng = np.random.default_rng(20261112)
    out = []
    for w in (0, 1, 2, 3, 7, 8, 9, 12, 16, 20, 31, 32):
        for n in (0, 1, 7, 8, 9, 16, 17, 63, 64, 65, 503, 504, 505, 512, 513, 1100):
            for name, v in shapes(rng, w, n):
                enc = orc.ref_rle_encode(v, w)
                out.append({"name": "%s_w%d_n%d" % (name, w, n), "width": w, "values": [int(x) for x in v], "hex": enc.hex()})
    with open(os.path.join(HERE, "rle_encode_ref_vectors.json"), "w") as f:
        json.dump({"what": "outputs of the reference's RleEncoder (Put x n, Flush)", "vectors": out}, f, separators=(",", ":"))
    print(len(out), "vectors")


if __name__ == "__main__":
    main()
