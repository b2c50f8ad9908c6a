"""p4s_walk_page_headers (the page-header walk of a column chunk on the device) against the host layer's own parse
(page_reader::next_page over thrift_compact_reader) and the CPU oracle's (oracle/chunk.py): the reference's
page_reader::next_page, src/column_chunk_reader.cc:27-50, on the reference's own column chunks, on generated chunks,
on headers with statistics and unknown fields, and on damaged chunks (same pages, same error text)."""
import json
import os

import numpy as np
import pytest

from oracle import chunk as ck
from parquet4seastar_amd import capi, hostapi, pagegen as g
from tests import alltypes_util as au, gpu_util as gu

pytestmark = pytest.mark.gpu

FIELDS = ["header_offset", "body_offset", "type", "uncompressed_page_size", "compressed_page_size", "num_values", "encoding",
          "definition_level_encoding", "repetition_level_encoding", "definition_levels_byte_length",
          "repetition_levels_byte_length", "num_nulls", "num_rows", "flags"]


def _same_index(dev, host, where=""):
    assert len(dev) == len(host), (where, len(dev), len(host))
    for f in FIELDS:
        a, b = dev[f], host[f]
        if f == "flags":                 # the host's PageHeader keeps the crc without an is-set bit
            a, b = a & ~np.uint32(capi.PAGE_HAS_CRC), b & ~np.uint32(capi.PAGE_HAS_CRC)
        assert np.array_equal(a, b), (where, f, a.tolist()[:8], b.tolist()[:8])


def _walk_both(ctx, chunks, flags=capi.CHUNK_UNCOMPRESSED, max_pages=None):
    """Device and host walks of every chunk: checks that they agree (pages and error text); returns the device's.
    The device's buffers sit between guard bands (tests/gpu_util.py); chunks start at odd addresses."""
    infos, status = gu.walk_page_headers(chunks, flags, max_pages, lead=3)
    for i, chunk in enumerate(chunks):
        host, err = hostapi.page_index(chunk)
        _same_index(infos[i], host, "chunk %d" % i)
        if err is None:
            assert status["code"][i] == 0, (i, capi.format_error(status[i]))
            assert int(status["bytes_consumed"][i]) == len(chunk)
        else:
            assert status["code"][i] != 0, (i, err)
            assert capi.format_error(status[i]) == err, (i, capi.format_error(status[i]), err)
    return infos, status


def test_alltypes_chunks():
    """Every column chunk of the reference's tests/test_data/alltypes files, all in one launch."""
    cols = au.load()
    chunks = [bytes(c["chunk"]) for c in cols]
    flags = [capi.CHUNK_UNCOMPRESSED | (capi.CHUNK_HAS_REP if c["max_rep"] else 0) | (capi.CHUNK_HAS_DEF if c["max_def"] else 0)
             for c in cols]
    with capi.Context(0) as ctx:
        infos, status = _walk_both(ctx, chunks, flags)
    for col, info, chunk in zip(cols, infos, chunks):
        pos = 0
        for pi in info:
            h, body = ck.read_page_header(chunk, pos)            # the oracle's parse of the same header
            assert (int(pi["header_offset"]), int(pi["body_offset"])) == (pos, body)
            assert (pi["type"], pi["uncompressed_page_size"], pi["compressed_page_size"]) == (
                h["type"], h["uncompressed_page_size"], h["compressed_page_size"])
            sub = h.get("data_page_header") or h.get("dictionary_page_header") or h.get("data_page_header_v2")
            assert pi["num_values"] == sub["num_values"] and pi["encoding"] == sub["encoding"]
            if "data_page_header" in h:
                # the level-section prefixes are the bytes level_decoder::reset_v1 would read (encoding.cc:35-50)
                p, want_rep, want_def = body, -1, -1
                if col["max_rep"] and h["data_page_header"]["repetition_level_encoding"] == ck.RLE:
                    want_rep = int.from_bytes(chunk[p:p + 4], "little", signed=True)
                    p += 4 + want_rep
                if col["max_def"] and h["data_page_header"]["definition_level_encoding"] == ck.RLE:
                    want_def = int.from_bytes(chunk[p:p + 4], "little", signed=True)
                assert (int(pi["v1_rep_prefix"]), int(pi["v1_def_prefix"])) == (want_rep, want_def), col["leaf"]
            pos = body + h["compressed_page_size"]
        assert pos == len(chunk)


def _generated_chunk(rng, n_pages, v2_every=3):
    dict_vals = rng.integers(-1000, 1000, size=37).astype(np.int32)
    parts = [g.dictionary_page(dict_vals.tobytes(), len(dict_vals))]
    for p in range(n_pages):
        n = int(rng.integers(1, 300))
        dl = (rng.random(n) >= 0.2).astype(np.uint32)
        vals = rng.integers(0, 1 << 20, size=int(dl.sum())).astype(np.int32)
        make = g.data_page_v2 if v2_every and p % v2_every == 2 else g.data_page_v1
        parts.append(make(vals.tobytes(), n, g.ENC_PLAIN, def_levels=dl, max_def=1))
    return b"".join(bytes(x) for x in parts)


def test_generated_chunks_many_pages_and_capacity():
    rng = np.random.default_rng(11)
    chunks = [_generated_chunk(rng, k) for k in (1, 7, 300, 2000)] + [b""]
    with capi.Context(0) as ctx:
        infos, status = _walk_both(ctx, chunks, capi.CHUNK_UNCOMPRESSED | capi.CHUNK_HAS_DEF)
        assert [len(x) for x in infos] == [2, 8, 301, 2001, 0]
        # fewer records than pages: the walk stops there and says so
        infos2, status2 = gu.walk_page_headers(chunks[2:3], capi.CHUNK_UNCOMPRESSED, max_pages=100)
        assert status2["code"][0] == 23 and len(infos2[0]) == 100 and int(status2["aux0"][0]) == 100
        _same_index(infos2[0], infos[2][:100])


# ---- a small Thrift compact-protocol writer, for headers the generators do not make
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _zz(v):
    return _varint((v << 1) ^ (v >> 63))


class _Struct:
    def __init__(self):
        self.b, self.last = bytearray(), 0

    def _head(self, fid, t):
        d = fid - self.last
        if 0 < d <= 15:
            self.b.append((d << 4) | t)
        else:
            self.b.append(t)
            self.b += _zz(fid)
        self.last = fid

    def i32(self, fid, v):
        self._head(fid, 5); self.b += _zz(v); return self

    def i64(self, fid, v):
        self._head(fid, 6); self.b += _zz(v); return self

    def boolean(self, fid, v):
        self._head(fid, 1 if v else 2); return self

    def byte(self, fid, v):
        self._head(fid, 3); self.b.append(v & 0xFF); return self

    def double(self, fid, raw8):
        self._head(fid, 7); self.b += raw8; return self

    def binary(self, fid, data):
        self._head(fid, 8); self.b += _varint(len(data)) + data; return self

    def struct(self, fid, inner):
        self._head(fid, 12); self.b += inner.done(); return self

    def raw(self, fid, t, payload):
        self._head(fid, t); self.b += payload; return self

    def done(self):
        return bytes(self.b) + b"\x00"


def _v1_header(body_len, num_values, stats=None, extra=None):
    d = _Struct().i32(1, num_values).i32(2, 0).i32(3, 3).i32(4, 3)
    if stats is not None:
        d.struct(5, stats)
    h = _Struct().i32(1, 0).i32(2, body_len).i32(3, body_len).i32(4, 12345).struct(5, d)
    if extra is not None:
        extra(h)
    return h.done()


def test_statistics_and_unknown_fields():
    """Headers far longer than the 512-byte window (statistics with long min / max), and fields the reader does not
    know, of every wire type, nested: skipped exactly as the host's reader skips them."""
    big = bytes(range(256)) * 5
    stats = _Struct().binary(1, big).binary(2, big[:700]).i64(3, 7).i64(4, 9).binary(5, b"mx").binary(6, b"mn")
    lst = bytes([0x35]) + b"".join(_zz(v) for v in (1, -2, 300))                       # list<i32> of 3
    lst_long = bytes([0xF8]) + _varint(20) + b"".join(_varint(3) + b"abc" for _ in range(20))     # list<binary> of 20
    lst_struct = bytes([0x2C]) + _Struct().i32(1, 5).done() + _Struct().boolean(1, True).boolean(2, False).done()
    lst_bool = bytes([0x41, 1, 2, 1, 1])                                              # list<bool>: one byte each
    mp = _varint(2) + bytes([0x58]) + _zz(1) + _varint(1) + b"x" + _zz(2) + _varint(0)  # map<i32, binary>
    nested = _Struct().struct(1, _Struct().struct(1, _Struct().raw(2, 9, lst)))

    def extra(h):
        h.raw(9, 9, lst).raw(10, 10, lst_long).raw(11, 9, lst_struct).raw(12, 9, lst_bool).raw(13, 11, mp)
        h.raw(14, 11, _varint(0)).double(15, b"\x00" * 8).byte(16, 200).boolean(17, True).struct(18, nested)
        h.i64(300, -5).binary(1000, b"tail")
    body = b"\x01" * 40
    pages = [_v1_header(len(body), 10, stats, extra) + body, _v1_header(len(body), 3) + body,
             _v1_header(len(body), 4, None, extra) + body]
    chunk = b"".join(pages)
    with capi.Context(0) as ctx:
        infos, status = _walk_both(ctx, [chunk, chunk[:-3]])
    assert len(infos[0]) == 3 and infos[0]["num_values"].tolist() == [10, 3, 4]
    assert all(int(f) & capi.PAGE_HAS_CRC for f in infos[0]["flags"])


def test_nesting_limit_and_bad_wire_types():
    body = b"\x00" * 8

    def deep(levels):
        def extra(h):
            s = _Struct().i32(1, 1)
            for _ in range(levels):
                s = _Struct().struct(1, s)
            h.struct(20, s)
        return _v1_header(len(body), 1, None, extra) + body

    def bad_type(h):
        h.raw(21, 13, b"")
    chunks = [deep(30), deep(31), deep(32), deep(33), deep(40), _v1_header(len(body), 1, None, bad_type) + body,
              _v1_header(len(body), 1, None, lambda h: h.raw(22, 5, b"\xff" * 11)) + body]
    with capi.Context(0) as ctx:
        infos, status = _walk_both(ctx, chunks)
    assert status["code"].tolist()[-2:] == [30, 28]
    assert 29 in status["code"].tolist()


def _mutate(rng, chunk, header_spans):
    b = bytearray(chunk)
    kind = int(rng.integers(0, 5))
    lo, hi = header_spans[int(rng.integers(0, len(header_spans)))]
    at = int(rng.integers(lo, hi))
    if kind == 0:
        b[at] = int(rng.integers(0, 256))
    elif kind == 1:
        b[at] ^= 1 << int(rng.integers(0, 8))
    elif kind == 2:
        del b[at:at + int(rng.integers(1, 4))]
    elif kind == 3:
        b[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 4)), dtype=np.uint8))
    else:
        del b[int(rng.integers(at, len(b))):]
    return bytes(b)


@pytest.mark.parametrize("seed", range(4))
def test_mutated_headers(seed):
    """Bytes of page headers overwritten, flipped, removed, inserted; chunks cut short: the device's walk finds the
    pages the host's finds and ends with the same message."""
    rng = np.random.default_rng(100 + seed)
    base = _generated_chunk(rng, 12)
    with capi.Context(0) as ctx:
        infos, _ = gu.walk_page_headers([base])
        spans = [(int(p["header_offset"]), int(p["body_offset"])) for p in infos[0]]
        chunks = [_mutate(rng, base, spans) for _ in range(400)]
        _, status = _walk_both(ctx, chunks, capi.CHUNK_UNCOMPRESSED | capi.CHUNK_HAS_DEF)
    assert len(set(status["code"].tolist())) >= 4          # several different endings were met


def test_column_chunk_reader_over_the_device_walk():
    """column_chunk_reader<T> with the header walk on the device (P4S_DEVICE_HEADER_WALK): the reference's alltypes
    chunks give the levels and tokens of the golden text, and a damaged chunk the host walk's page errors."""
    os.environ["P4S_DEVICE_HEADER_WALK"] = "1"
    try:
        for col in au.load():
            with hostapi.Chunk(col["physical_type"], col["chunk"], col["max_def"], col["max_rep"],
                               type_length=col["type_length"]) as c:
                d_all, r_all, v_all = [], [], []
                while True:
                    d, r, v = c.read_batch(1024)
                    if len(d) == 0:
                        break
                    d_all += d.tolist(); r_all += r.tolist(); v_all += list(v) if not isinstance(v, np.ndarray) else [x for x in v]
            where = "%s:%s" % (col["file"], col["leaf"])
            assert d_all == col["def"] and r_all == col["rep"], where
            assert [au.render(col["render"], x) for x in v_all] == col["tokens"], where
        rng = np.random.default_rng(5)
        base = _generated_chunk(rng, 6, v2_every=0)
        cut = base[:len(base) - 7]                       # the last page's body is short
        texts = []
        for walk in ("1", "0"):
            os.environ["P4S_DEVICE_HEADER_WALK"] = walk
            with hostapi.Chunk(g.TYPE_INT32, cut, 1, 0) as c:
                c.plan(); c.launch()
                counts = c.collect()
                texts.append((counts["pages"], counts["failed_pages"], c.page_error(counts["pages"] - 1)))
        assert texts[0] == texts[1] and texts[0][1] == 1 and "Unexpected end of column chunk" in texts[0][2]
    finally:
        os.environ.pop("P4S_DEVICE_HEADER_WALK", None)
