"""GPU parity: Snappy page decompression through the C ABI against the oracle's restatement of libsnappy and the
committed libsnappy vectors.  Byte-exact; failures carry the reference's three texts (src/compression.cc:53-63)."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import capi
from tests.gpu_util import run_snappy

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _check(pages, sizes, **kw):
    expect = []
    for pg, sz in zip(pages, sizes):
        try:
            expect.append((orc.snappy_decompress(pg, sz), None))
        except orc.OracleError as e:
            expect.append((None, e))
    outs, status = run_snappy(pages, sizes, [int(s) + 2 for s in sizes], **kw)
    for i, ((res, err), o) in enumerate(zip(expect, outs)):
        st = status[i]
        if err is not None:
            assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
            continue
        assert st["code"] == 0 and st["values_decoded"] == len(res) and st["out_bytes"] == len(res), (i, st, len(res))
        assert st["bytes_consumed"] == len(pages[i]), (i, st)
        assert np.array_equal(o, res), "page %d differs" % i
    return outs, status


def test_snappy_vectors():
    vs = json.load(open(os.path.join(HERE, "golden", "snappy_vectors.json")))["vectors"]
    pages = [bytes.fromhex(v["input"]) for v in vs]
    outs, status = _check(pages, [v["size"] for v in vs])
    for v, o, st in zip(vs, outs, status):
        if v["sha256"] is None:
            assert st["code"] in (24, 26), v["name"]
        else:
            assert st["code"] == 0 and hashlib.sha256(o.tobytes()).hexdigest() == v["sha256"], v["name"]


def _literal(data):
    n = len(data) - 1
    if n < 60:
        return bytes([n << 2]) + data
    nb = (n.bit_length() + 7) // 8
    return bytes([(59 + nb) << 2]) + n.to_bytes(nb, "little") + data


def _copy(length, offset):
    if 4 <= length <= 11 and offset < 2048:
        return bytes([((length - 4) << 2) | ((offset >> 8) << 5) | 1, offset & 0xFF])
    if offset < 65536:
        return bytes([((length - 1) << 2) | 2]) + offset.to_bytes(2, "little")
    return bytes([((length - 1) << 2) | 3]) + offset.to_bytes(4, "little")


def _varint(v):
    out = bytearray()
    while True:
        out.append((v & 0x7F) | (0x80 if v > 0x7F else 0))
        v >>= 7
        if not v:
            return bytes(out)


def _synth(rng, n_target, far=False):
    """A hand-made stream of random elements (literals of every length class, copies of every kind, offsets shorter
    than the length, 4-byte offsets) and the size it decompresses to."""
    body, produced = bytearray(), 0
    while produced < n_target:
        if produced == 0 or rng.random() < 0.4:
            ln = int(rng.choice([1, 2, 59, 60, 61, 255, 256, 257, 3000]))
            body += _literal(rng.integers(0, 256, size=ln, dtype=np.uint8).tobytes())
            produced += ln
        else:
            ln = int(rng.integers(1, 65))
            off = int(rng.integers(1, min(produced, 70000 if far else 3000) + 1))
            if far and rng.random() < 0.3 and off < 65536:
                body += bytes([((ln - 1) << 2) | 3]) + off.to_bytes(4, "little")
            else:
                body += _copy(ln, off)
            produced += ln
    return _varint(produced) + bytes(body), produced


def test_snappy_synthetic_streams():
    rng = np.random.default_rng(5)
    pages, sizes = [], []
    for it in range(60):
        pg, n = _synth(rng, int(rng.choice([1, 50, 3000, 40000, 150000])), far=(it % 3 == 0))
        pages.append(pg)
        sizes.append(n + (7 if it % 5 == 0 else 0))            # a header may announce a larger buffer
    # preamble and capacity errors
    pages += [b"", b"\x80", b"\xff\xff\xff\xff\x7f", bytes([5, 4 << 2]) + b"hello", bytes([5, 4 << 2]) + b"hello"]
    sizes += [10, 10, 10, 4, 5]
    _check(pages, sizes)
    _check(pages, sizes, align=16, lead=3)


def test_snappy_compressor_output():
    pa = pytest.importorskip("pyarrow")
    codec = pa.Codec("snappy")
    rng = np.random.default_rng(6)
    pages, sizes = [], []
    for it in range(40):
        n = int(rng.choice([0, 1, 100, 5000, 70000, 1 << 20]))
        if it % 4 == 0:
            data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif it % 4 == 1:
            data = rng.integers(0, 4, size=n, dtype=np.uint8).tobytes()
        elif it % 4 == 2:
            data = (b"http://www.example.com/" * (n // 23 + 1))[:n]
        else:
            data = np.cumsum(rng.integers(-3, 4, size=n // 8 + 1)).astype(np.int64).tobytes()[:n]
        pages.append(codec.compress(data, asbytes=True))
        sizes.append(len(data))
    outs, _ = _check(pages, sizes)
