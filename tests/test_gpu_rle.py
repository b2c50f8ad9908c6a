"""GPU parity: RLE/bit-packed hybrid, level decoder and dictionary pages through the C ABI
against the oracle, the reference's unit-test vectors and outputs recorded from the
reference's own RleDecoder.  Bit-exact."""
import struct

import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import capi, pagegen
from tests.gpu_util import ctx, run_fixed

pytestmark = pytest.mark.gpu

DT = {1: np.uint8, 2: np.int16, 4: np.int32}


def _run_rle(streams, widths, ns, es, **kw):
    expect = [orc.rle_decode(s, w, n, es) for s, w, n in zip(streams, widths, ns)]
    outs, status = run_fixed("p4s_decode_rle", streams, ns, es, [len(e) + 2 for e in expect], param0=widths,
                             extra=(es,), **kw)
    for i, (o, e) in enumerate(zip(outs, expect)):
        assert status[i]["code"] == 0, (i, status[i])
        assert status[i]["values_decoded"] == len(e), (i, widths[i], ns[i], status[i], len(e))
        assert np.array_equal(o.view(DT[es]), e), (i, widths[i], ns[i])
    return outs


def test_rle_reference_vectors(reference_vectors, l0_vectors):
    by_es = {}
    for v in reference_vectors:
        if v["kind"] == "rle":
            n = min(sum(v["batches"]), 200000)
            by_es.setdefault(v["elem_size"], []).append((bytes.fromhex(v["input"]), v["bit_width"], n, None))
    for v in l0_vectors:
        if v["kind"] == "rle" and v["bit_width"] <= 32:
            n = min(sum(v["batches"]), 200000)
            exp = v["values"] if len(v["batches"]) == 1 else None
            by_es.setdefault(v["elem_size"], []).append((bytes.fromhex(v["input"]), v["bit_width"], n, exp))
    assert sum(len(x) for x in by_es.values()) > 150
    for es, items in by_es.items():
        outs = _run_rle([i[0] for i in items], [i[1] for i in items], [i[2] for i in items], es)
        for (stream, w, n, exp), o in zip(items, outs):
            if exp is not None:   # the reference's own recorded output for this single call
                assert np.array_equal(o.view(DT[es]), np.asarray(exp).astype(DT[es])), (w, n)


def _random_stream(rng, w, n_target, big=False):
    runs, total = [], 0
    while total < n_target:
        if rng.random() < 0.5:
            c = int(rng.integers(0, 3000 if big else 150))
            runs.append(("rle", c, int(rng.integers(0, 1 << 32))))
            total += c
        else:
            g = int(rng.integers(0, 100 if big else 12))
            vals = rng.integers(0, 1 << w, size=8 * g) if w else np.zeros(8 * g, dtype=np.int64)
            runs.append(("lit", [int(x) for x in vals]))
            total += 8 * g
    return pagegen.rle_from_runs(runs, w), total


@pytest.mark.parametrize("es", [1, 2, 4])
def test_rle_random(es):
    rng = np.random.default_rng(40 + es)
    streams, widths, ns = [], [], []
    maxw = {1: 8, 2: 16, 4: 32}[es]
    for it in range(150):
        w = int(rng.integers(0, maxw + 1))
        s, total = _random_stream(rng, w, int(rng.integers(0, 6000)), big=(it % 10 == 0))
        if it % 4 == 1 and len(s):
            s = s[: int(rng.integers(0, len(s)))]          # truncated stream
        streams.append(s)
        widths.append(w)
        ns.append(int(rng.choice([0, 1, 7, 8, 9, total // 2, total, total + 100])))
    _run_rle(streams, widths, ns, es)
    _run_rle(streams, widths, ns, es, align=16, lead=7)


def test_rle_long_pages():
    rng = np.random.default_rng(77)
    streams, widths, ns = [], [], []
    for w in (1, 3, 8, 12, 17, 20, 32):
        n = 300000
        vals = []
        while len(vals) < n:
            ln = int(rng.integers(8, 108))
            if rng.random() < 0.5:
                vals.extend([int(rng.integers(0, 1 << w))] * ln)
            else:
                vals.extend(int(x) for x in rng.integers(0, 1 << w, size=ln))
        vals = np.asarray(vals[:n], dtype=np.uint32)
        streams.append(pagegen.rle_encode(vals, w))
        widths.append(w)
        ns.append(n)
        out = orc.rle_decode(streams[-1], w, n, 4)
        assert np.array_equal(out.view(np.uint32), vals)
        # long literal runs with multi-byte headers, and one enormous repeated run
        streams.append(pagegen.rle_encode(rng.integers(0, 1 << w, size=100000).astype(np.uint32), w, max_groups=5000))
        widths.append(w)
        ns.append(100000)
        streams.append(pagegen.rle_from_runs([("rle", 2000000, 5), ("lit", [1] * 8)], w))
        widths.append(w)
        ns.append(1500000)
    _run_rle(streams, widths, ns, 4)


def test_levels():
    rng = np.random.default_rng(91)
    pages, ns, p0, p1, flags, expect = [], [], [], [], [], []

    def add(page, max_level, num_values, n, kind):
        pages.append(page); ns.append(n); p0.append(max_level); p1.append(num_values); flags.append(kind)
        try:
            if kind == capi.LEVELS_V2:
                expect.append((orc.levels_v2(page, max_level, num_values, n), len(page), None))
            else:
                lv, used = orc.levels_v1(page, 3 if kind == capi.LEVELS_V1_RLE else 4, max_level, num_values, n)
                expect.append((lv, used, None))
        except orc.OracleError as e:
            expect.append((None, None, e))

    add(bytes([0x02, 0, 0, 0, 0x14, 0x01, 0xAA]), 1, 7, 100, capi.LEVELS_V1_RLE)
    add(b"whatever", 0, 5, 100, capi.LEVELS_V1_RLE)
    add(bytes([0x88, 0xC6, 0xFA]), 7, 8, 8, capi.LEVELS_V1_BIT_PACKED)
    add(bytes([0x05, 0x39, 0x77]), 7, 8, 8, capi.LEVELS_V1_BIT_PACKED)
    add(b"ab", 1, 5, 5, capi.LEVELS_V1_RLE)
    add(struct.pack("<i", -1) + b"xx", 1, 5, 5, capi.LEVELS_V1_RLE)
    add(struct.pack("<i", 100) + b"xx", 1, 5, 5, capi.LEVELS_V1_RLE)
    add(bytes([0x88]), 7, 8, 8, capi.LEVELS_V1_BIT_PACKED)
    add(bytes([0x14, 0x01]), 1, 7, 100, capi.LEVELS_V2)
    add(bytes([0x14, 0x01]), 1, 7, 3, capi.LEVELS_V2)
    for it in range(60):
        max_level = int(rng.choice([1, 1, 2, 3, 7, 100, 65535]))
        nv = int(rng.integers(0, 20000))
        lv = rng.integers(0, max_level + 1, size=nv)
        if it % 2:
            lv[rng.random(nv) < 0.9] = max_level            # mostly non-null: long runs
        n = int(rng.choice([nv, nv + 10, nv // 2]))
        kind = it % 3
        w = pagegen.bit_width(max_level)
        if kind == capi.LEVELS_V1_RLE:
            page = pagegen.levels_v1_rle_encode(lv, max_level) + b"values follow"
        elif kind == capi.LEVELS_V1_BIT_PACKED:
            page = pagegen.bitpack(lv.astype(np.uint32), w) + b"values follow"
        else:
            page = pagegen.rle_encode(lv.astype(np.uint32), w)
        add(page, max_level, nv, n, kind)
    for es in (4, 2):
        outs, status = run_fixed("p4s_decode_levels", pages, ns, es, [max(p1[i], 1) + 2 for i in range(len(pages))],
                                 param0=p0, param1=p1, flags=flags, extra=(es,))
        for i, ((lv, used, err), o) in enumerate(zip(expect, outs)):
            st = status[i]
            if err is not None:
                assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
                continue
            assert st["code"] == 0, (i, st)
            assert st["values_decoded"] == len(lv), (i, st, len(lv))
            assert np.array_equal(o.view(DT[es]), lv.astype(DT[es])), i     # static_cast<LevelT>, as the reference stores
            assert st["aux0"] == int(np.count_nonzero(lv == p0[i])), i     # values_to_read
            if flags[i] != capi.LEVELS_V2:
                assert st["bytes_consumed"] == used, (i, st, used)


def test_levels_narrow_streams_one_workgroup_per_page():
    """Level streams of 1 to 4 bits in a batch too large for the small-batch split (one workgroup per page: the layout
    with the pointer-jumping walk, the bit map of run starts and byte staging): values, the count of levels equal to the
    maximum and the count of levels ABOVE it (column_chunk_reader.hh:124-139), for streams of short runs, long repeated
    runs, literal runs only, and values the bit width can hold but the maximum level forbids."""
    rng = np.random.default_rng(404)
    pages, ns, p0, p1, truth = [], [], [], [], []
    for it in range(540):
        max_level = int(rng.choice([1, 1, 1, 2, 3, 5, 7, 9, 15]))
        w = pagegen.bit_width(max_level)
        top = (1 << w) - 1                                   # what the stream can hold
        nv = int(rng.integers(1, 40000)) if it % 9 else int(rng.integers(1, 9))
        shape = it % 4
        if shape == 0:                                       # 10 % below the maximum at random places: a header every few bytes
            lv = np.where(rng.random(nv) >= 0.1, max_level, rng.integers(0, max_level + 1, size=nv))
        elif shape == 1:                                     # long repeated runs
            lv = np.repeat(rng.integers(0, top + 1, size=nv // 3000 + 1), 3000)[:nv]
        elif shape == 2:                                     # literal runs only
            lv = rng.integers(0, top + 1, size=nv)
        else:                                                # runs of 1 to 40 values
            lv = np.repeat(rng.integers(0, top + 1, size=nv // 2 + 1), rng.integers(1, 41, size=nv // 2 + 1))[:nv]
        lv = lv.astype(np.uint32)
        n = nv if it % 5 else max(nv // 2, 1)
        pages.append(pagegen.rle_encode(lv, w)); ns.append(n); p0.append(max_level); p1.append(nv); truth.append(lv[:n])
    for es in (4, 2):
        outs, status = run_fixed("p4s_decode_levels", pages, ns, es, [nv + 2 for nv in p1], param0=p0, param1=p1,
                                 flags=[capi.LEVELS_V2] * len(pages), extra=(es,))
        for i, (lv, o) in enumerate(zip(truth, outs)):
            st = status[i]
            assert st["code"] == 0 and st["values_decoded"] == len(lv), (i, st, len(lv))
            assert np.array_equal(o.view(DT[es]), lv.astype(DT[es])), (i, p0[i])
            assert st["aux0"] == int(np.count_nonzero(lv == p0[i])), (i, p0[i], st)
            assert st["aux1"] == int(np.count_nonzero(lv > p0[i])), (i, p0[i], st)
            if i % 60 == 0:
                assert np.array_equal(orc.levels_v2(pages[i], p0[i], p1[i], ns[i]), lv), i


@pytest.mark.parametrize("es,dt", [(4, np.int32), (8, np.int64), (12, None)])
def test_dictionary(es, dt):
    rng = np.random.default_rng(300 + es)
    dsize = 1000
    if dt is None:
        dictionary = rng.integers(0, 256, size=(dsize, 12), dtype=np.uint8)
    else:
        dictionary = rng.integers(-2**31, 2**31 - 1, size=dsize).astype(dt)
    raw = np.ascontiguousarray(dictionary).view(np.uint8).reshape(dsize, es)
    padded = np.zeros((dsize + 1, es), dtype=np.uint8)
    padded[:dsize] = raw
    c = ctx()
    d_dict = c.to_device(padded)
    pages, ns, expect = [], [], []

    def add(page, n):
        pages.append(page); ns.append(n)
        try:
            # oracle over the dictionary {1..dsize}: returns index+1, or 0 (the pad element) for index == dsize
            got = orc.dict_decode(page, np.arange(1, dsize + 1, dtype=np.int32), n)
            expect.append((np.where(got == 0, dsize, got - 1), None))
        except orc.OracleError as e:
            expect.append((None, e))

    w = pagegen.bit_width(dsize - 1)
    for it in range(40):
        n = int(rng.integers(0, 30000))
        if it % 2:
            idx = rng.integers(0, dsize, size=n)
        else:
            idx = np.minimum((rng.zipf(1.3, size=n) - 1), dsize - 1)
        page = pagegen.dict_page_values_encode(idx.astype(np.uint32), w, max_groups=63 if it % 3 else 4000)
        if it % 5 == 4 and len(page) > 2:
            page = page[: int(rng.integers(1, len(page)))]     # truncated: exercises the 1000-value batching
        add(page, int(rng.choice([n, n + 50, n // 3])))
    add(bytes([0x02, 0x08, 0x02]), 4)
    add(bytes([10, 0x08, 0xE9, 0x03]), 4)                    # index 1001 > dict size 1000
    add(bytes([10, 0x08, 0xE8, 0x03]), 4)                    # index == dict size: accepted (ref off-by-one)
    add(bytes([33, 0x08, 0x03]), 4)
    add(bytes([0x00, 0x0A]), 100)
    add(bytes([0x00, 0x03]), 100)
    add(b"", 10)
    bad = pagegen.dict_page_values_encode(np.asarray([1, 2, 3, 5000, 7, 6000] + [1] * 3000, dtype=np.uint32), 13)
    add(bad, 3006)
    outs, status = run_fixed("p4s_decode_dictionary", pages, ns, es, [max(n, 1) + 2 for n in ns],
                             dict_args=(d_dict, dsize))
    d_dict.free()
    for i, ((idx, err), o) in enumerate(zip(expect, outs)):
        st = status[i]
        if err is not None:
            assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
            continue
        assert st["code"] == 0 and st["values_decoded"] == len(idx), (i, st, len(idx))
        assert np.array_equal(o.reshape(-1, es), padded[idx]), i
