"""The Snappy restatement in oracle/ against libsnappy: the committed vectors (streams made, damaged and judged by
libsnappy, tests/golden/make_snappy_vectors.py) and, where pyarrow is importable, fresh round trips.  The reference
holds no Snappy fixture of its own."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def test_snappy_vectors():
    vs = json.load(open(os.path.join(HERE, "golden", "snappy_vectors.json")))["vectors"]
    assert len(vs) > 40
    accepted = 0
    for v in vs:
        data = bytes.fromhex(v["input"])
        if v["sha256"] is None:
            with pytest.raises(orc.OracleError) as e:
                orc.snappy_decompress(data, v["size"])
            assert e.value.code in (24, 26), v["name"]
        else:
            out = orc.snappy_decompress(data, v["size"]).tobytes()
            assert len(out) == v["size"] and hashlib.sha256(out).hexdigest() == v["sha256"], v["name"]
            accepted += 1
    assert accepted > 15


def test_snappy_errors():
    # compression.cc:53-63: bad preamble, buffer smaller than the announced size, anything else
    for bad in (b"", b"\x80", b"\xff\xff\xff\xff\x7f", b"\x80\x80\x80\x80\x80\x01"):
        with pytest.raises(orc.OracleError) as e:
            orc.snappy_decompress(bad, 100)
        assert e.value.code == 24 and str(e.value) == "Invalid or corrupted parquet file: Corrupt snappy data"
    lit = bytes([5, 4 << 2]) + b"hello"
    assert orc.snappy_decompress(lit, 5).tobytes() == b"hello"
    assert orc.snappy_decompress(lit, 9).tobytes() == b"hello"
    with pytest.raises(orc.OracleError) as e:
        orc.snappy_decompress(lit, 4)
    assert e.value.code == 25 and str(e.value) == "Invalid or corrupted parquet file: Uncompression buffer size too small"
    for bad in (bytes([5, 4 << 2]) + b"hell",                      # literal cut short
                bytes([6, 4 << 2]) + b"hello",                     # fewer bytes than announced
                bytes([4, 4 << 2]) + b"hello",                     # more bytes than announced
                bytes([5, 0 << 2]) + b"h" + bytes([0x01, 0x00]),   # copy with offset 0
                bytes([5, 0 << 2]) + b"h" + bytes([0x01, 0x02]),   # copy from before the start
                bytes([5, 0 << 2]) + b"h" + bytes([0x01])):        # copy tag without its offset byte
        with pytest.raises(orc.OracleError) as e:
            orc.snappy_decompress(bad, 8)
        assert e.value.code == 26 and str(e.value) == "Could not decompress snappy."
    # offset < length repeats the pattern: "ab" then copy of 6 from 2 back
    assert orc.snappy_decompress(bytes([8, 1 << 2]) + b"ab" + bytes([((6 - 4) << 2) | 1, 2]), 8).tobytes() == b"abababab"


def test_snappy_round_trips():
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("snappy")
    rng = np.random.default_rng(3)
    for it in range(120):
        n = int(rng.choice([0, 1, 5, 100, 5000, 70000, 200000]))
        if it % 4 == 0:
            data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif it % 4 == 1:
            data = rng.integers(0, 4, size=n, dtype=np.uint8).tobytes()
        elif it % 4 == 2:
            data = (b"http://www.example.com/" * (n // 23 + 1))[:n]
        else:
            data = np.cumsum(rng.integers(-3, 4, size=n // 8 + 1)).astype(np.int64).tobytes()[:n]
        comp = codec.compress(data, asbytes=True)
        assert orc.snappy_decompress(comp, len(data)).tobytes() == data
        b = bytearray(comp)
        if len(b) > 3:
            b[int(rng.integers(1, len(b)))] = int(rng.integers(0, 256))
            try:
                ref = codec.decompress(bytes(b), decompressed_size=len(data), asbytes=True)
            except Exception:
                ref = None
            try:
                mine = orc.snappy_decompress(bytes(b), len(data)).tobytes()
            except orc.OracleError:
                mine = None
            assert ref == mine, it
