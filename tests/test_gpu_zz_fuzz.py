"""GPU parity on damaged input: valid pages and streams with a few bytes overwritten, inserted or removed, so that
headers turn into other headers, lengths overrun the data and nested streams break.  Values, short counts and errors
must be the oracle's, bit for bit.  These tests are collected last."""
import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import pagegen
from tests.test_gpu_delta_bss_plain import _check_delta
from tests.test_gpu_rle import _random_stream, _run_rle
from tests.test_gpu_strings import _check, _rand_strings, _sorted_keys

pytestmark = pytest.mark.gpu


def _mutate(rng, page):
    b = bytearray(page)
    for _ in range(int(rng.integers(1, 4))):
        if len(b) < 6:
            break
        pos = int(rng.integers(4, len(b)))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            b[pos] = int(rng.integers(0, 256))
        elif kind == 1:
            b[pos] |= 0x80
        else:
            del b[pos:pos + int(rng.integers(1, 30))]
    return bytes(b)


def test_mutated_string_pages(seed=0):
    """Valid DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY pages with bytes overwritten or removed anywhere past the
    first header bytes: lengths that overrun the data, prefixes longer than the previous value, broken nested
    headers.  Values and errors must be the oracle's."""
    rng = np.random.default_rng(23 + 1000 * seed)
    dl, db, ns = [], [], []
    for it in range(60):
        n = int(rng.choice([5, 200, 1000, 5000]))
        strs = _sorted_keys(rng, n) if it % 2 else sorted(_rand_strings(rng, n, 24))
        dl.append(_mutate(rng, pagegen.dlba_encode(strs, 128, 4)))
        db.append(_mutate(rng, pagegen.dba_encode(strs)))
        # (other seeds, run by tools/fuzz_soak.py, also ask for fewer or more values than the page holds)
        ns.append(n if seed == 0 else int(rng.choice([n, n // 2, n + 9, 1])))
    _check("dlba", orc.dlba_decode, dl, ns)
    _check("dba", orc.dba_decode, db, ns)


@pytest.mark.parametrize("es", [2, 4])
def test_rle_mutated_streams(es, seed=0):
    """Valid streams with a few bytes overwritten, inserted or removed: run headers become other headers (long
    runs, zero-length runs, literal runs that outrun the stream, multi-byte ULEB128s).  Counts and values must be
    the reference's for one GetBatch of n."""
    rng = np.random.default_rng(90 + es + 1000 * seed)
    streams, widths, ns = [], [], []
    maxw = {2: 16, 4: 32}[es]
    for it in range(200):
        w = int(rng.integers(1, maxw + 1))
        s, total = _random_stream(rng, w, int(rng.integers(200, 9000)), big=(it % 10 == 0))
        b = bytearray(s)
        for _ in range(int(rng.integers(1, 4))):
            if not b:
                break
            pos = int(rng.integers(0, len(b)))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                b[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                b[pos] |= 0x80
            elif kind == 2:
                del b[pos:pos + int(rng.integers(1, 20))]
            else:
                b[pos:pos] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8))
        streams.append(bytes(b))
        widths.append(w)
        ns.append(int(rng.choice([total // 2, total, total + 100])))
    _run_rle(streams, widths, ns, es)
    _run_rle(streams, widths, ns, es, align=16, lead=5)


@pytest.mark.parametrize("es", [4, 8])
def test_delta_mutated_pages(es, seed=0):
    """Valid multi-block pages with a few bytes overwritten past the page header: block headers (min_delta, width
    bytes) and bodies change under the header walker.  Values, short counts and errors must match the oracle."""
    rng = np.random.default_rng(77 + es + 1000 * seed)
    dt = np.int32 if es == 4 else np.int64
    pages, ns = [], []
    for it in range(160):
        bs, nmb = ((256, 8), (128, 4), (512, 16), (64, 2))[it % 4]
        n = int(rng.choice([700, 3000, 20000]))
        vals = rng.integers(-100, 3996, size=n).cumsum().astype(dt)
        page = bytearray(pagegen.delta_encode(vals, bs, nmb))
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(8, len(page)))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                page[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                page[pos] |= 0x80                       # a ULEB128 that runs on, or a width above 64
            elif kind == 2:
                page[pos] = int(rng.integers(65, 256))
            else:
                del page[pos:pos + int(rng.integers(1, 40))]
        pages.append(bytes(page))
        ns.append(n + 3 if it % 3 else max(1, n // 2))
    _check_delta(pages, ns, es)


def test_plain_byte_array_mutated(seed=0):
    """PLAIN BYTE_ARRAY pages with damaged length fields: lengths that overrun the page, a page that ends inside a
    length."""
    rng = np.random.default_rng(55 + 1000 * seed)
    pages, ns = [], []
    for it in range(80):
        n = int(rng.choice([3, 64, 65, 1000, 8000]))
        strs = _rand_strings(rng, n, int(rng.choice([0, 3, 24, 300])) if n <= 1000 else 24)
        pages.append(_mutate(rng, pagegen.plain_byte_array_encode(strs)))
        ns.append(int(rng.choice([n, n + 7, n // 2])))
    _check("plain_ba", orc.plain_byte_array, pages, ns)


@pytest.mark.parametrize("es", [4, 8])
def test_dictionary_mutated(es, seed=0):
    """Dictionary-index pages (width byte + hybrid runs) with damaged bytes: other widths, indices past the
    dictionary, runs that outrun the page."""
    from parquet4seastar_amd import capi
    from tests.gpu_util import ctx, run_fixed
    rng = np.random.default_rng(66 + es + 1000 * seed)
    dsize = 1000
    dictionary = rng.integers(0, 256, size=(dsize + 1, es), dtype=np.uint8)      # [dsize] = the pad element
    d_dict = ctx().to_device(dictionary)
    w = pagegen.bit_width(dsize - 1)
    pages, ns, expect = [], [], []
    for it in range(120):
        n = int(rng.integers(1, 12000))
        idx = rng.integers(0, dsize, size=n) if it % 2 else np.minimum(rng.zipf(1.3, size=n) - 1, dsize - 1)
        page = bytearray(pagegen.dict_page_values_encode(idx.astype(np.uint32), w, max_groups=63 if it % 3 else 4000))
        for _ in range(int(rng.integers(1, 4))):
            if len(page) < 2:                 # (the deletions before left nothing to damage)
                break
            pos = int(rng.integers(0 if it % 8 == 0 else 1, len(page)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                page[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                page[pos] |= 0x80
            else:
                del page[pos:pos + int(rng.integers(1, 20))]
        pages.append(bytes(page))
        ns.append(int(rng.choice([n, n + 50, n // 3])))
        try:
            got = orc.dict_decode(pages[-1], np.arange(1, dsize + 1, dtype=np.int32), ns[-1])
            expect.append((np.where(got == 0, dsize, got - 1), None))     # 0: the element past the dictionary
        except orc.OracleError as e:
            expect.append((None, e))
    try:
        outs, status = run_fixed("p4s_decode_dictionary", pages, ns, es, [max(n, 1) + 2 for n in ns],
                                 dict_args=(d_dict, dsize))
    finally:
        d_dict.free()
    for i, ((idx, err), o) in enumerate(zip(expect, outs)):
        st = status[i]
        if err is not None:
            assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
            continue
        assert st["code"] == 0 and st["values_decoded"] == len(idx), (i, st, len(idx))
        assert np.array_equal(o.reshape(-1, es), dictionary[idx]), i


def test_levels_mutated(seed=0):
    """Level sections (v1 RLE with its length prefix, v1 bit-packed, v2) with damaged bytes."""
    from parquet4seastar_amd import capi
    from tests.gpu_util import run_fixed
    rng = np.random.default_rng(88 + 1000 * seed)
    pages, ns, p0, p1, flags, expect = [], [], [], [], [], []
    for it in range(120):
        max_level = int(rng.choice([1, 1, 2, 3, 7, 100, 65535]))
        nv = int(rng.integers(1, 12000))
        lv = rng.integers(0, max_level + 1, size=nv)
        if it % 2:
            lv[rng.random(nv) < 0.9] = max_level
        kind = it % 3
        w = pagegen.bit_width(max_level)
        if kind == capi.LEVELS_V1_RLE:
            page = pagegen.levels_v1_rle_encode(lv, max_level) + b"values follow"
        elif kind == capi.LEVELS_V1_BIT_PACKED:
            page = pagegen.bitpack(lv.astype(np.uint32), w) + b"values follow"
        else:
            page = pagegen.rle_encode(lv.astype(np.uint32), w)
        page = bytearray(page)
        for _ in range(int(rng.integers(1, 4))):
            if len(page) < 2:
                break
            pos = int(rng.integers(0, len(page)))
            k = int(rng.integers(0, 3))
            if k == 0:
                page[pos] = int(rng.integers(0, 256))
            elif k == 1:
                page[pos] |= 0x80
            else:
                del page[pos:pos + int(rng.integers(1, 20))]
        page = bytes(page)
        n = int(rng.choice([nv, nv + 10, nv // 2]))
        pages.append(page); ns.append(n); p0.append(max_level); p1.append(nv); flags.append(kind)
        try:
            if kind == capi.LEVELS_V2:
                expect.append((orc.levels_v2(page, max_level, nv, n), len(page), None))
            else:
                lv2, used = orc.levels_v1(page, 3 if kind == capi.LEVELS_V1_RLE else 4, max_level, nv, n)
                expect.append((lv2, used, None))
        except orc.OracleError as e:
            expect.append((None, None, e))
    dts = {2: np.int16, 4: np.int32}
    for es in (4, 2):
        outs, status = run_fixed("p4s_decode_levels", pages, ns, es, [p1[i] + 2 for i in range(len(pages))],
                                 param0=p0, param1=p1, flags=flags, extra=(es,))
        for i, ((lv, used, err), o) in enumerate(zip(expect, outs)):
            st = status[i]
            if err is not None:
                assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
                continue
            assert st["code"] == 0, (i, st)
            assert st["values_decoded"] == len(lv), (i, st, len(lv))
            assert np.array_equal(o.view(dts[es]), lv.astype(dts[es])), i
            assert st["aux0"] == int(np.count_nonzero(lv == p0[i])), i
            if flags[i] != capi.LEVELS_V2:
                assert st["bytes_consumed"] == used, (i, st, used)


def test_snappy_mutated(seed=0):
    """Snappy streams with damaged bytes: other tags, offsets before the start of the output, lengths past the
    announced size, streams cut inside an element.  Accepted or not, and the bytes, must be the oracle's."""
    from tests.test_gpu_snappy import _check as check_snappy, _synth
    rng = np.random.default_rng(99 + 1000 * seed)
    pages, sizes = [], []
    for it in range(100):
        pg, n = _synth(rng, int(rng.choice([20, 3000, 40000])), far=(it % 3 == 0))
        b = bytearray(pg)
        for _ in range(int(rng.integers(1, 4))):
            if len(b) < 3:
                break
            pos = int(rng.integers(0 if it % 10 == 0 else 1, len(b)))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                b[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                b[pos] ^= 1 << int(rng.integers(0, 8))
            else:
                del b[pos:pos + int(rng.integers(1, 20))]
        pages.append(bytes(b))
        sizes.append(n if it % 4 else n + 100)
    check_snappy(pages, sizes)
