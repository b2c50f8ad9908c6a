"""GPU DELTA_BINARY_PACKED encoder (p4s_encode_delta_binary_packed) against the oracle's restatement of
delta_binary_packed_encoder<T> (src/encoding.cc:862-996), byte for byte, and through the GPU decoder again.
The reference leaves the padding bits behind a page's last value undefined; oracle and kernel write zeros, so whole
pages compare.  One reference output is on record (alternating INT64_MIN / INT64_MAX, then steps of 2^32): its
header and first block head are checked literally."""
import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import capi, pagegen as g
from tests import gpu_util as gu

pytestmark = pytest.mark.gpu


def _cases(rng, dt):
    info = np.iinfo(dt)
    yield np.zeros(0, dtype=dt)
    yield np.array([info.min], dtype=dt)
    yield np.array([5, 5], dtype=dt)
    for n in (31, 32, 33, 256, 257, 258, 289, 512, 513, 1000, 70001):
        yield rng.integers(-100, 3996, size=n).cumsum().astype(dt)                      # narrow deltas
        yield rng.integers(info.min, info.max, size=n, endpoint=True).astype(dt)       # full range: wrap-around, widest fields
    yield np.arange(3000, dtype=dt) * 7                                                 # constant stride: width 0
    a = np.empty(600, dtype=dt)
    a[0::2], a[1::2] = info.min, info.max                                               # deltas of +-(2^k - 1): two-bit fields
    yield a
    v = rng.integers(-5, 5, size=4000).astype(dt)
    v[1234] = info.max                                                                  # one outlier in one miniblock
    yield v
    yield np.concatenate([np.zeros(300, dtype=dt), rng.integers(0, 2 ** 20, size=300).astype(dt)])


@pytest.mark.parametrize("elem_size", [4, 8])
def test_delta_encoder_matches_the_oracle_and_decodes_back(elem_size):
    dt = np.int32 if elem_size == 4 else np.int64
    rng = np.random.default_rng(900 + elem_size)
    cases = list(_cases(rng, dt))
    values = np.concatenate(cases)
    counts = [len(c) for c in cases]
    with capi.Context(0) as ctx:
        pages, st = capi.encode_delta_binary_packed(ctx, values, counts, elem_size)
    assert all(int(s["code"]) == 0 for s in st)
    for c, p, s in zip(cases, pages, st):
        want = orc.delta_encode(c, elem_size)
        assert p == want, (len(c), len(p), len(want))
        assert p == g.delta_encode(c, 256, 8)
        assert int(s["values_decoded"]) == len(c) and int(s["bytes_consumed"]) == len(c) * elem_size
        assert len(p) <= capi.delta_encode_bound(len(c), elem_size)
        assert np.array_equal(orc.delta_decode(p, len(c) + 3, elem_size), c)
    # and back through the GPU decoder
    outs, status = gu.run_fixed("p4s_decode_delta_binary_packed", pages, counts, elem_size, [n + 2 for n in counts],
                                extra=(elem_size,))
    for c, raw, s in zip(cases, outs, status):
        assert int(s["code"]) == 0 and np.array_equal(raw.view(dt), c)


def test_delta_encoder_reference_round_trip_inputs():
    """The inputs of the reference's encoder tests (tests/delta_binary_packed_test.cc:92-188: 1337 ascending values,
    MIN / MAX alternation, squares; the empty input; three repeats) through k_delta_encode and back through
    k_delta_binary_packed: decoded == input, and the pages equal the oracle's byte for byte."""
    from tests.test_oracle import reference_encoder_test_inputs
    for es, v in reference_encoder_test_inputs():
        dt = np.int32 if es == 4 else np.int64
        with capi.Context(0) as ctx:
            pages, st = capi.encode_delta_binary_packed(ctx, np.concatenate([v, v, v]), [len(v)] * 3, es)
        assert all(int(s["code"]) == 0 for s in st)
        assert pages[0] == pages[1] == pages[2] == orc.delta_encode(v, es)
        outs, status = gu.run_fixed("p4s_decode_delta_binary_packed", pages, [len(v)] * 3, es, [len(v) + 2] * 3, extra=(es,))
        for raw, s in zip(outs, status):
            assert int(s["code"]) == 0 and np.array_equal(raw.view(dt), v)


def test_delta_encoder_reference_output_on_record():
    """What the reference's encoder wrote for 300 values alternating INT64_MIN / INT64_MAX followed by 300 values in
    steps of 2^32: header 80 02 08 D8 04 FF x9 01, first block 01 | 02 x8 (wrap-around makes every delta +-(2^64 - 1))."""
    a = np.empty(300, dtype=np.int64)
    a[0::2], a[1::2] = np.iinfo(np.int64).min, np.iinfo(np.int64).max
    b = a[-1] + (np.arange(1, 301, dtype=np.uint64) * np.uint64(2 ** 32)).astype(np.int64)
    v = np.concatenate([a, b])
    with capi.Context(0) as ctx:
        pages, st = capi.encode_delta_binary_packed(ctx, v, [len(v)], 8)
    want_head = bytes([0x80, 0x02, 0x08, 0xD8, 0x04] + [0xFF] * 9 + [0x01] + [0x01] + [0x02] * 8)
    assert pages[0][:len(want_head)] == want_head
    assert pages[0] == orc.delta_encode(v, 8)


def test_delta_encoder_capacity():
    """A slot that is too small: P4S_E_OUT_CAPACITY with the size the page needs, nothing written behind the slot;
    its neighbours are unaffected."""
    rng = np.random.default_rng(7)
    c = [rng.integers(-1000, 1000, size=n).astype(np.int64) for n in (700, 700, 700)]
    sizes = [len(orc.delta_encode(x, 8)) for x in c]
    with capi.Context(0) as ctx:
        pages, st = capi.encode_delta_binary_packed(ctx, np.concatenate(c), [700] * 3, 8, caps=[sizes[0], sizes[1] - 1, sizes[2]])
    assert [int(s["code"]) for s in st] == [0, 23, 0]
    assert int(st[1]["aux0"]) == sizes[1]
    assert pages[0] == orc.delta_encode(c[0], 8) and pages[2] == orc.delta_encode(c[2], 8) and pages[1] is None


def test_delta_encoder_full_size_pages():
    """BASELINE config 2's pages the other way round: 64 pages x 131072 int64 values, encoded on the GPU, equal to
    the generator's pages and decoded back by the GPU."""
    from parquet4seastar_amd.pagegen import synth
    n = 131072
    enc, values = synth.delta_pages(8, n, 11)
    with capi.Context(0) as ctx:
        pages, st = capi.encode_delta_binary_packed(ctx, np.tile(values, 8), [n] * 64, 8)
    assert all(int(s["code"]) == 0 for s in st)
    for i, p in enumerate(pages):
        assert p == bytes(enc[i % 8])


def test_delta_encoder_ragged_batches_and_slots():
    """Many pages of random sizes and value ranges in one launch, some slots too small by a random amount: every page
    that fits equals the oracle's, every one that does not reports the bytes it needs, and nothing is written
    behind a page's bytes (or, for a page that did not fit, behind its slot's capacity)."""
    rng = np.random.default_rng(31)
    for elem_size, dt in ((4, np.int32), (8, np.int64)):
        info = np.iinfo(dt)
        cases, caps, want = [], [], []
        for i in range(150):
            n = int(rng.choice([0, 1, 2, 7, 255, 256, 257, 300, 1025, 5000, int(rng.integers(1, 20000))]))
            bits = int(rng.integers(1, 8 * elem_size))
            v = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=n).astype(dt) if i % 3 else \
                rng.integers(info.min, info.max, size=n, endpoint=True).astype(dt)
            page = orc.delta_encode(v, elem_size)
            cases.append(v)
            want.append(page)
            caps.append(len(page) if i % 4 else max(len(page) - int(rng.integers(1, 400)), 0))
        with capi.Context(0) as ctx:
            pages, st = capi.encode_delta_binary_packed(ctx, np.concatenate(cases), [len(c) for c in cases], elem_size,
                                                        caps=caps, guard=0xA5)
        for i, (p, s) in enumerate(zip(pages, st)):
            if caps[i] >= len(want[i]):
                assert int(s["code"]) == 0 and p == want[i], i
            else:
                assert int(s["code"]) == 23 and int(s["aux0"]) == len(want[i]) and p is None, i
