"""GPU parity: DELTA_BINARY_PACKED, BYTE_STREAM_SPLIT, PLAIN fixed/boolean through the C ABI
against the CPU oracle and the reference's golden vectors.  Bit-exact (integer/byte work)."""
import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import pagegen
from tests.gpu_util import run_fixed

pytestmark = pytest.mark.gpu


def _oracle_delta(page, n, es):
    try:
        return orc.delta_decode(page, n, es, return_bytes_left=True), None
    except orc.OracleError as e:
        return None, e


def _check_delta(pages, ns, es, **kw):
    dt = np.int32 if es == 4 else np.int64
    caps = []
    expect = []
    for pg, n in zip(pages, ns):
        res, err = _oracle_delta(pg, n, es)
        expect.append((res, err))
        # a page that ends in an error may have written the values before it: its slot holds num_values
        caps.append((len(res[0]) if res else min(n, 1 << 20)) + 3)
    outs, status = run_fixed("p4s_decode_delta_binary_packed", pages, ns, es, caps, extra=(es,), **kw)
    for i, ((res, err), out, st) in enumerate(zip(expect, outs, status)):
        if err is not None:
            assert st["code"] == err.code, (i, st, str(err))
            from parquet4seastar_amd import capi
            assert capi.format_error(st) == str(err), i
        else:
            vals, left = res
            assert st["code"] == 0, (i, st)
            assert st["values_decoded"] == len(vals), (i, st, len(vals))
            assert np.array_equal(out.view(dt), vals), "page %d differs" % i
            # size - bytes_left() after the read, partial reads included: a read of n < total values has consumed
            # n deltas (ref src/encoding.cc:212-237)
            assert st["bytes_consumed"] == len(pages[i]) - left, (i, st, left, ns[i])


def _total(page):
    # total value count from the page header (3rd ULEB128)
    pos = 0
    vals = []
    for _ in range(3):
        v, sh = 0, 0
        while True:
            b = page[pos]
            pos += 1
            v |= (b & 0x7F) << sh
            sh += 7
            if not b & 0x80:
                break
        vals.append(v)
    return vals[2]


def test_delta_reference_vectors(reference_vectors):
    pages, ns, ess = [], [], []
    for v in reference_vectors:
        if v["kind"] == "delta":
            _check_delta([bytes.fromhex(v["input"])], [v["n"]], v["elem_size"])
            out, st = run_fixed("p4s_decode_delta_binary_packed", [bytes.fromhex(v["input"])], [v["n"]],
                                v["elem_size"], [len(v["expect"]) + 2], extra=(v["elem_size"],))
            assert list(out[0].view(np.int32)) == v["expect"]
        elif v["kind"] == "delta_roundtrip":
            dt = np.int32 if v["elem_size"] == 4 else np.int64
            vals = np.asarray(v["values"], dtype=dt)
            for bs, nmb in ((256, 8), (128, 4)):
                page = pagegen.delta_encode(vals, bs, nmb)
                out, st = run_fixed("p4s_decode_delta_binary_packed", [page], [len(vals) + 5], v["elem_size"],
                                    [len(vals) + 2], extra=(v["elem_size"],))
                assert st[0]["code"] == 0 and np.array_equal(out[0].view(dt), vals), v["name"]


@pytest.mark.parametrize("es", [4, 8])
def test_delta_random_shapes(es):
    rng = np.random.default_rng(100 + es)
    dt = np.int32 if es == 4 else np.int64
    pages, ns = [], []
    shapes = [(256, 8), (128, 4), (128, 1), (12, 4), (8, 8), (1024, 2), (64, 64), (4096, 1), (24, 3), (520, 65)]
    for it in range(120):
        bs, nmb = shapes[it % len(shapes)]
        n = int(rng.choice([0, 1, 2, 3, bs, bs + 1, bs + 2, 2 * bs + 1, 1000, 5000, 20000]))
        mode = it % 5
        if mode == 0:
            vals = rng.integers(-100, 3996, size=n).cumsum()
        elif mode == 1:
            vals = np.arange(n) * 7 + 3                      # constant stride: width 0
        elif mode == 2:
            lim = 2**31 if es == 4 else 2**63
            vals = rng.integers(-lim, lim - 1, size=n)       # full range: widths up to 64 with wrap-around
        elif mode == 3:
            vals = rng.integers(0, 2, size=n) * (2**31 - 1 if es == 4 else 2**63 - 1)
        else:
            vals = (rng.integers(0, 1 << int(rng.integers(1, 30)), size=n)).cumsum()
        vals = vals.astype(dt)
        page = pagegen.delta_encode(vals, bs, nmb)
        if it % 7 == 3:
            page += b"trailing-bytes"
        pages.append(page)
        ns.append(n + 5 if it % 3 else max(0, n - int(rng.integers(0, n + 1))))
    _check_delta(pages, ns, es)
    _check_delta(pages, ns, es, align=16, lead=3)


def test_delta_large_pages():
    rng = np.random.default_rng(5)
    pages, ns = [], []
    for k in range(6):
        n = 131072 + (k % 2)
        vals = rng.integers(-100, 3996, size=n).astype(np.int64).cumsum()
        pages.append(pagegen.delta_encode(vals, 256 if k % 2 else 128, 8 if k % 2 else 4))
        ns.append(n)
    full = rng.integers(-2**63, 2**63 - 1, size=70000, dtype=np.int64)
    pages.append(pagegen.delta_encode(full, 256, 8))
    ns.append(70000)
    _check_delta(pages, ns, 8)
    _check_delta(pages, ns, 8, align=64)


@pytest.mark.parametrize("es", [8, 4])
def test_delta_pages_split_over_workgroups(es):
    """Small batches of long pages are cut into parts that several workgroups decode (the later parts relative to 0,
    their bases added afterwards); damaged pages fall back to the one-workgroup decode.  Same outcome as ever."""
    rng = np.random.default_rng(31 + es)
    dt = np.int64 if es == 8 else np.int32
    pages, ns = [], []
    for it in range(14):
        bs, nmb = [(256, 8), (128, 4), (512, 16), (256, 1), (1024, 32), (64, 2)][it % 6]
        n = int(rng.integers(16 * bs + 1, 60 * bs)) if it % 5 else 131072 + it
        if it % 4 == 0:
            vals = rng.integers(-2**62, 2**62, size=n).astype(dt)                 # wide, wrap-around sums
        elif it % 4 == 1:
            vals = (np.arange(n) * 3 + 11).astype(dt)                             # every width 0
        else:
            vals = rng.integers(-100, 3996, size=n).cumsum().astype(dt)
        page = pagegen.delta_encode(vals, bs, nmb)
        if it % 3 == 1:
            page += b"what follows the stream"
        pages.append(page)
        ns.append(n if it % 2 else n + 100)
    # damage inside later parts: a width byte above 64, a cut in the middle, a cut inside the last miniblock
    vals = rng.integers(-100, 3996, size=50000).cumsum().astype(dt)
    good = pagegen.delta_encode(vals, 256, 8)
    bad = bytearray(good)
    hdr = len(good) - orc.delta_decode(good, 0, es, return_bytes_left=True)[1]
    end_100 = len(good) - orc.delta_decode(good, 100 * 256, es, return_bytes_left=True)[1]
    bad[end_100 + 2] = 77                      # a width byte of block 100 (min_delta takes one or two bytes here)
    pages += [bytes(bad), good[:len(good) // 2], good[:len(good) - 3], good[:hdr + 40]]
    ns += [50000, 50000, 50000, 50000]
    # reads that stop before the end are not split
    pages += [good, good]
    ns += [49999, 25000]
    for k in (1, 3, len(pages)):
        _check_delta(pages[:k], ns[:k], es)
    _check_delta(pages, ns, es, align=16, lead=7)
    _check_delta(pages[::-1], ns[::-1], es)


def test_delta_split_pages_damaged():
    """Long pages with bytes overwritten anywhere (headers, width bytes, packed deltas) in batches small enough to be
    split: values, counts, stream positions and errors are the oracle's."""
    rng = np.random.default_rng(404)
    vals = rng.integers(-100, 3996, size=40000).cumsum().astype(np.int64)
    good = pagegen.delta_encode(vals, 256, 8)
    pages = []
    for _ in range(24):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        if rng.random() < 0.3:
            del b[int(rng.integers(len(b) // 2, len(b))):]
        pages.append(bytes(b))
    for lo in range(0, len(pages), 8):
        _check_delta(pages[lo:lo + 8], [40000] * 8, 8)


def test_delta_partial_reads_consume_the_next_delta():
    """read_batch(n) with n < total reads the delta after the n-th value (ref src/encoding.cc:212-237), so a page cut
    where that delta's miniblock or block header would start fails with the reference's error, although the n values
    asked for are all there; cut one miniblock later, it succeeds and reports the reference's stream position."""
    rng = np.random.default_rng(77)
    for es in (8, 4):
        dt = np.int64 if es == 8 else np.int32
        for bs, nmb in ((256, 8), (128, 4), (64, 1)):
            vpm = bs // nmb
            vals = rng.integers(-5000, 5000, size=3 * bs + 17).cumsum().astype(dt)
            page = pagegen.delta_encode(vals, bs, nmb)
            header = len(page) - orc.delta_decode(page, 0, es, return_bytes_left=True)[1]
            pages, ns = [page[:header], page[:header + 1], page[:header], page], [1, 1, 0, 1]
            for k in (1, 2, nmb - 1, nmb, nmb + 1, 2 * nmb, 3 * nmb):
                if k == 0 or k * vpm + 1 >= len(vals):
                    continue
                # a read of k*vpm values consumes k whole miniblocks: the stream position after it
                end_k = len(page) - orc.delta_decode(page, k * vpm, es, return_bytes_left=True)[1]
                cut = page[:end_k]
                pages += [cut, cut, cut, page[:end_k + 1], page]
                ns += [k * vpm, k * vpm + 1, k * vpm - 1, k * vpm + 1, k * vpm + 1]
            _check_delta(pages, ns, es)
            _check_delta(pages, ns, es, align=16, lead=5)


def test_delta_corrupt_pages():
    rng = np.random.default_rng(9)
    vals = rng.integers(-1000, 1000, size=3000).astype(np.int64).cumsum()
    good = pagegen.delta_encode(vals, 256, 8)
    pages = [good[:cut] for cut in (0, 1, 2, 3, 4, 5, 6, 7, 12, 40, 100, 1000, len(good) - 1)]
    pages.append(bytes([0x80, 0x01, 0x00, 0x05, 0x10]))                         # zero miniblocks
    pages.append(bytes([0x80, 0x01, 0x04, 0x05, 0x10, 0x00, 0x01]))             # short block header
    pages.append(bytes([0x80, 0x01, 0x04, 0x05, 0x10, 0x00, 65, 0, 0, 0]) + bytes(64))   # width 65
    pages.append(bytes([0x04, 0x08, 0x05, 0x10, 0x00] + [1] * 8 + [0xFF] * 8))  # block 4 < 8 miniblocks
    pages.append(bytes([0x80] * 10 + [0x01]))                                   # over-long ULEB128
    ns = [5000] * len(pages)
    _check_delta(pages, ns, 8)
    # truncated pages that are still fine when fewer values are requested
    _check_delta([good[: len(good) // 2]] * 3, [10, 300, 1500], 8)


def test_byte_stream_split(reference_vectors):
    for v in reference_vectors:
        if v["kind"] == "bss":
            es = v["elem_size"]
            out, st = run_fixed("p4s_decode_byte_stream_split", [bytes.fromhex(v["input"])], [v["n"]], es, [8],
                                extra=(es,))
            assert st[0]["code"] == 0 and out[0].tobytes() == bytes.fromhex(v["expect"]), v["name"]
    rng = np.random.default_rng(11)
    for es, dt in ((4, np.float32), (8, np.float64)):
        pages, ns, expect = [], [], []
        for n in (0, 1, 2, 3, 4, 5, 7, 63, 64, 65, 1000, 4097, 262144, 300001):
            bits = rng.integers(0, 256, size=n * es, dtype=np.uint8)   # raw bit patterns incl. NaN payloads
            vals = bits.view(dt)
            pages.append(pagegen.bss_encode(vals))
            ns.append(n + 3)
            expect.append(bits)
            pages.append(pagegen.bss_encode(vals))
            ns.append(n // 2)
            expect.append(bits[: (n // 2) * es])
        for kw in ({}, {"align": 16}, {"lead": 1}):
            outs, status = run_fixed("p4s_decode_byte_stream_split", pages, ns, es, [len(e) // es + 1 for e in expect],
                                     extra=(es,), **kw)
            for i, (o, e) in enumerate(zip(outs, expect)):
                assert status[i]["code"] == 0
                assert np.array_equal(o, e), (es, i, kw)
                assert np.array_equal(o, orc.bss_decode(pages[i], es, ns[i]))
        outs, status = run_fixed("p4s_decode_byte_stream_split", [bytes(12)], [10], es, [4], extra=(es,))
        if es == 8:
            assert status[0]["code"] == 9
            from parquet4seastar_amd import capi
            assert capi.format_error(status[0]) == "Data size in BYTE_STREAM_SPLIT is not divisible by size of data type"


def test_plain_fixed_and_boolean():
    rng = np.random.default_rng(13)
    for es in (4, 8, 12):
        pages, ns, expect = [], [], []
        for nbytes in (0, 3, 10, 16, 100, 4096, 100003):
            pg = bytes(rng.integers(0, 256, size=nbytes, dtype=np.uint8))
            for n in (0, 2, 10**6):
                pages.append(pg)
                ns.append(n)
                expect.append(orc.plain_fixed(pg, es, n))
        for kw in ({}, {"align": 16}, {"lead": 5}):
            outs, status = run_fixed("p4s_decode_plain_fixed", pages, ns, es, [len(e) // es + 1 for e in expect],
                                     extra=(es,), **kw)
            for i, (o, e) in enumerate(zip(outs, expect)):
                assert status[i]["code"] == 0 and np.array_equal(o, e), (es, i, kw)
    pages, ns, expect = [], [], []
    for nbytes in (0, 1, 2, 5, 64, 1000, 30011):
        pg = bytes(rng.integers(0, 256, size=nbytes, dtype=np.uint8))
        for n in (0, 1, 7, 9, nbytes * 8, nbytes * 8 + 5, max(0, nbytes * 8 - 3)):
            pages.append(pg)
            ns.append(n)
            expect.append(orc.plain_boolean(pg, n))
    pages.append(bytes([0xB2, 0x01]))
    ns.append(16)
    expect.append(np.asarray([0, 1, 0, 0, 1, 1, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0], dtype=np.uint8))
    for gap in (0, 1):
        outs, status = run_fixed("p4s_decode_plain_boolean", pages, ns, 1, [len(e) + 1 for e in expect], out_gap=gap)
        for i, (o, e) in enumerate(zip(outs, expect)):
            assert status[i]["code"] == 0 and np.array_equal(o, e), i
