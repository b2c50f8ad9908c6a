"""The benchmark configurations at their real page shapes, as parity tests.

Config 4: dictionary-encoded INT32 column chunks, 65535-entry / 16-bit and 65536-entry / 17-bit dictionaries
(the reference's encoder sizes indices by bit_width(cardinality), src/encoding.cc:769-771), 64 v1 pages of 1Mi
indices each, uniform and Zipf, with and without max_def = 1, through column_chunk_decoder<INT32> (every page
compared with the generated column) and column_chunk_reader<INT32>::read_batch; a page sample is also walked by
the CPU oracle, so generator, oracle and GPU agree.

Config 5: DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY pages of 65536 strings (the benchmark's page shape)
against the generated strings and the oracle.

Config 1 as literally stated: ONE page of 1Mi int32 values, every width 1-20, against the generated values and
(for a sample of widths) the oracle."""
import numpy as np
import pytest

from oracle import chunk as ck
from oracle import oracle as orc
from parquet4seastar_amd import capi, hostapi, pagegen as g
from parquet4seastar_amd.pagegen import synth
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu

MI = 1 << 20


def _download(ptr, dtype, count):
    out = np.empty(count, dtype=dtype)
    L, h = capi.lib(), hostapi.current_ctx()
    assert L.p4s_copy_d2h(h, out.ctypes.data, ptr, out.nbytes) == 0
    assert L.p4s_ctx_sync(h) == 0
    return out


@pytest.mark.parametrize("dsize,dist,max_def", [(65535, "uniform", 0), (65536, "zipf", 1), (65536, "uniform", 0),
                                                (65535, "zipf", 1)])
def test_config4_dictionary_chunk(dsize, dist, max_def):
    rng = np.random.default_rng(4000 + dsize + max_def)
    n, distinct, reps = MI, 8, 8
    dict_page, pages, dictionary, truth_vals, truth_def = synth.dict_chunk_pages(rng, dsize, dist, max_def, n, distinct)
    assert synth.dict_index_width(dsize) == (17 if dsize == 65536 else 16)
    body = np.frombuffer(b"".join(pages), dtype=np.uint8)
    chunk = np.concatenate([np.frombuffer(dict_page, dtype=np.uint8), np.tile(body, reps)])
    n_pages = distinct * reps
    # generator vs oracle on a sample (dictionary page + first and last distinct page)
    sample = ck.decode_chunk(ck.INT32, dict_page + pages[0] + pages[-1], max_def, 0)
    for got, p in zip(sample, (0, distinct - 1)):
        assert np.array_equal(got["values"], truth_vals[p]) and np.array_equal(got["def"], truth_def[p])
    # batched decode of the whole chunk: every page against the generated column
    with hostapi.Chunk(g.TYPE_INT32, chunk, max_def, 0) as c:
        c.plan()
        c.launch()
        counts = c.collect()
        assert counts["failed_pages"] == 0 and counts["data_pages"] == n_pages and counts["pages"] == n_pages + 1
        assert counts["levels"] == n * n_pages
        assert counts["values"] == sum(len(v) for v in truth_vals) * reps
        vals = _download(c.device_ptr("values"), np.int32, n * n_pages)
        for p in range(n_pages):
            tv = truth_vals[p % distinct]
            assert np.array_equal(vals[p * n: p * n + len(tv)], tv), "values of page %d" % p
        del vals
        if max_def:
            defs = _download(c.device_ptr("def"), np.int32, n * n_pages)
            for p in range(n_pages):
                assert np.array_equal(defs[p * n:(p + 1) * n], truth_def[p % distinct]), "def levels of page %d" % p
    # the reference's interface: triplet batches that never span pages (column_chunk_reader.hh:113-123)
    with hostapi.Chunk(g.TYPE_INT32, chunk[: len(dict_page) + len(pages[0]) + len(pages[1])], max_def, 0) as c:
        for p in range(2):
            got_d, got_v, left = [], [], n
            for b in (1024, 1000, 65536, n):            # the last read comes back short
                d, r, v = c.read_batch(b)
                assert len(d) == min(b, left) and not r.any()
                left -= len(d)
                got_d.append(d); got_v.append(v)
            assert left == 0
            assert np.array_equal(np.concatenate(got_v), truth_vals[p])
            assert np.array_equal(np.concatenate(got_d), truth_def[p])
        assert len(c.read_batch(10)[0]) == 0


def test_config5_string_pages_of_65536():
    rng = np.random.default_rng(5005)
    n, distinct = 65536, 4
    pages = synth.string_pages(rng, n, distinct)
    for kind, key in (("dlba", "dlba"), ("dba", "dba")):
        enc = [p[key] for p in pages]
        caps = [int(p["offsets"][-1]) + 16 for p in pages]
        if kind == "dba":
            sized, st = gu.run_strings("dba_sizes", enc, n, [n] * distinct, caps)
            assert not st["code"].any()
            assert [int(s["out_bytes"]) for s in st] == [int(p["offsets"][-1]) for p in pages]
        outs, st = gu.run_strings(kind, enc, n, [n] * distinct, caps)
        assert not st["code"].any() and all(int(s["values_decoded"]) == n for s in st)
        for (o, b), p, e in zip(outs, pages, enc):
            assert np.array_equal(o, p["offsets"]) and np.array_equal(b, p["bytes"])
            wo, wb = (orc.dlba_decode if kind == "dlba" else orc.dba_decode)(e, n)
            assert np.array_equal(o, wo) and np.array_equal(b, wb)


@pytest.mark.parametrize("w", list(range(1, 21)))
def test_config1_one_page_of_1mi_values(w):
    rng = np.random.default_rng(1000 + w)
    vals = synth.run_values(rng, MI, w)
    page = g.rle_encode(vals, w)
    outs, st = gu.run_fixed("p4s_decode_rle", [page], MI, 4, [MI], param0=w, extra=(4,))
    assert st[0]["code"] == 0 and int(st[0]["values_decoded"]) == MI
    got = outs[0].view(np.uint32)
    assert np.array_equal(got, vals)
    if w in (1, 7, 12, 16, 20):
        assert np.array_equal(got, orc.rle_decode(page, w, MI, 4).view(np.uint32))
