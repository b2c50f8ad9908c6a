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


def test_snappy_then_delta_on_device():
    """The two steps of column_chunk_reader.cc:104-110 + value decode chained on the device: compressed pages ->
    p4s_decompress_snappy -> the buffer is the input of p4s_decode_delta_binary_packed."""
    from parquet4seastar_amd import pagegen
    from tests.gpu_util import _dbg_arm, _dbg_check, ctx
    try:
        import pyarrow as pa
        compress = lambda b: pa.Codec("snappy").compress(b, asbytes=True)
    except ImportError:
        compress = lambda b: _varint(len(b)) + (_literal(b) if b else b"")
    rng = np.random.default_rng(8)
    cols = [rng.integers(-50, 2000, size=n).cumsum().astype(np.int64) for n in (1, 700, 5000, 40000)]
    raw = [pagegen.delta_encode(c_, 256, 8) for c_ in cols]
    comp = [compress(p) for p in raw]
    c, L = ctx(), capi.lib()
    # compressed pages in, raw pages out at 16-byte aligned offsets of one device buffer
    buf, offs, sizes = capi.pack_pages(comp, align=16)
    raw_offs, pos = [], 0
    for p in raw:
        raw_offs.append(pos)
        pos = (pos + len(p) + 15) // 16 * 16
    raw_bytes = pos + 16
    d_comp = c.to_device(buf)
    d_raw = c.alloc(raw_bytes)
    d_desc1 = c.to_device(capi.make_descs(offs, sizes, raw_offs, 0, [len(p) for p in raw]).view(np.uint8))
    d_st1 = c.alloc(48 * len(raw))
    _dbg_arm([d_comp, d_raw, d_desc1, d_st1])          # (only a bounds-checked build of the library looks at these)
    c._check(L.p4s_decompress_snappy(c.handle, d_comp.ptr, buf.size, d_desc1.ptr, len(raw), d_raw.ptr, raw_bytes, d_st1.ptr))
    if hasattr(L, "p4s_debug_get_hits"):
        c.sync()
        _dbg_check()
    out_offs = np.concatenate([[0], np.cumsum([len(x) for x in cols])])[:-1]
    d_desc2 = c.to_device(capi.make_descs(raw_offs, [len(p) for p in raw], [int(o) for o in out_offs],
                                          [len(x) for x in cols]).view(np.uint8))
    d_st2 = c.alloc(48 * len(raw))
    total = int(sum(len(x) for x in cols))
    d_out = c.alloc(total * 8)
    _dbg_arm([d_raw, d_desc2, d_st2, d_out])
    c._check(L.p4s_decode_delta_binary_packed(c.handle, d_raw.ptr, raw_bytes, d_desc2.ptr, len(raw), d_out.ptr, d_st2.ptr, 8))
    c.sync()
    _dbg_check()
    st1 = d_st1.download(capi.PAGE_STATUS_DTYPE, len(raw))
    st2 = d_st2.download(capi.PAGE_STATUS_DTYPE, len(raw))
    got = d_out.download(np.int64, total)
    for b in (d_comp, d_raw, d_desc1, d_st1, d_desc2, d_st2, d_out):
        b.free()
    assert all(s["code"] == 0 for s in st1) and all(s["code"] == 0 for s in st2), (st1, st2)
    assert np.array_equal(got, np.concatenate(cols))
