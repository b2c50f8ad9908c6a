"""GPU parity: DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY, PLAIN BYTE_ARRAY and FIXED_LEN_BYTE_ARRAY
through the C ABI, as (offsets, bytes), against the oracle and the reference's unit-test vectors."""
import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import capi, pagegen
from tests.gpu_util import run_strings

pytestmark = pytest.mark.gpu


def _split(offsets, data):
    return [bytes(data[offsets[i]:offsets[i + 1]]) for i in range(len(offsets) - 1)]


def _check(kind, oracle_fn, pages, ns, **kw):
    expect = []
    for pg, n in zip(pages, ns):
        try:
            expect.append((oracle_fn(pg, n), None))
        except orc.OracleError as e:
            expect.append((None, e))
    # the contract: num_values + 1 offset slots per page (a page may write them before it finds an error)
    vcaps = [int(n) + 2 for n in ns]
    bcaps = [(len(e[0][1]) if e[0] else 0) + 5 for e in expect]
    outs, status = run_strings(kind, pages, ns, vcaps, bcaps, **kw)
    for i, ((res, err), (o, b)) in enumerate(zip(expect, outs)):
        st = status[i]
        if err is not None:
            assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
            continue
        assert st["code"] == 0, (i, st, capi.format_error(st))
        assert np.array_equal(o, res[0]), (i, o[:10], res[0][:10])
        assert np.array_equal(b, res[1]), i
    return outs, status


def _rand_strings(rng, n, maxlen):
    return [bytes(rng.integers(97, 123, size=int(rng.integers(0, maxlen + 1)), dtype=np.uint8)) for _ in range(n)]


def _sorted_keys(rng, n):
    hosts = [b"http://www.example.com/", b"http://www.example.org/a/", b"https://data.site/"]
    keys = set()
    while len(keys) < n:
        h = hosts[int(rng.integers(0, len(hosts)))]
        keys.add(h + bytes(rng.integers(97, 101, size=int(rng.integers(0, 12)), dtype=np.uint8)))
    return sorted(keys)


def _uleb(v):
    out = bytearray()
    while True:
        out.append((v & 0x7F) | (0x80 if v > 0x7F else 0))
        v >>= 7
        if not v:
            return bytes(out)


def _announce(page, total):
    """The page with the value count of its (first) DELTA header replaced."""
    pos, ends = 0, []
    for _ in range(3):
        while page[pos] & 0x80:
            pos += 1
        pos += 1
        ends.append(pos)
    return page[:ends[1]] + _uleb(total) + page[ends[2]:]


def test_reference_vectors(reference_vectors):
    for v in reference_vectors:
        if v["kind"] in ("dlba", "dba"):
            fn = orc.dlba_decode if v["kind"] == "dlba" else orc.dba_decode
            outs, _ = _check(v["kind"], fn, [bytes.fromhex(v["input"])], [v["n"]])
            assert _split(*outs[0]) == [bytes.fromhex(e) for e in v["expect"]], v["name"]


def test_dlba():
    rng = np.random.default_rng(21)
    pages, ns = [], []
    for it in range(40):
        n = int(rng.choice([0, 1, 2, 5, 127, 128, 129, 1000, 20000]))
        strs = _rand_strings(rng, n, int(rng.choice([0, 3, 24, 300])))
        pg = pagegen.dlba_encode(strs, *((128, 4) if it % 2 else (256, 8)))
        if it % 6 == 5 and len(pg) > 8:
            pg = pg[: len(pg) - int(rng.integers(1, 8))]
        pages.append(pg)
        ns.append(int(rng.choice([n, n + 7, n // 2])))
    pages.append(pagegen.delta_encode(np.asarray([10], dtype=np.int32), 128, 4) + b"ab")
    ns.append(5)
    # a length stream that announces more than 2^32 values: what the reference meets first while it reads them
    # (the string bytes taken for a block header) is the error, not the count
    pages.append(_announce(pagegen.dlba_encode(_rand_strings(rng, 300, 12), 128, 4), (1 << 33) + 300))
    ns.append(300)
    _check("dlba", orc.dlba_decode, pages, ns)
    _check("dlba", orc.dlba_decode, pages, ns, align=16, lead=5)


def test_dba():
    rng = np.random.default_rng(22)
    pages, ns = [], []
    for it in range(40):
        n = int(rng.choice([0, 1, 2, 5, 255, 256, 257, 1000, 6000]))
        if it % 2:
            strs = _sorted_keys(rng, n)
        else:
            strs = _rand_strings(rng, n, int(rng.choice([0, 3, 24])))
            if it % 4 == 0:
                strs = sorted(strs)
        pg = pagegen.dba_encode(strs)
        if it % 7 == 6 and len(pg) > 8:
            pg = pg[: len(pg) - int(rng.integers(1, 8))]
        pages.append(pg)
        ns.append(int(rng.choice([n, n + 7])))
    # long shared prefixes that cross the 256-string tiles, and strings longer than a tile of bytes
    base = b"x" * 700
    pages.append(pagegen.dba_encode([base + b"%05d" % i for i in range(1500)]))
    ns.append(1500)
    pages.append(pagegen.dba_encode([b"k" * (i % 40) for i in range(3000)]))
    ns.append(3000)
    # prefix longer than the previous string, and more suffixes than prefix lengths
    pages.append(pagegen.delta_encode(np.asarray([3], dtype=np.int32), 128, 4) + pagegen.dlba_encode([b"abc"]))
    ns.append(10)
    pages.append(pagegen.delta_encode(np.asarray([0, 2, 9], dtype=np.int32), 128, 4)
                 + pagegen.dlba_encode([b"abc", b"de", b"f"]))
    ns.append(10)
    pages.append(pagegen.delta_encode(np.asarray([0], dtype=np.int32), 128, 4) + pagegen.dlba_encode([b"abc", b"de"]))
    ns.append(10)
    # the same in the middle of larger pages: the string without a prefix length is met by one thread of many
    for n_pre, n_suf in ((100, 300), (257, 2000), (1, 1200)):
        pages.append(pagegen.delta_encode(np.zeros(n_pre, dtype=np.int32), 128, 4) + pagegen.dlba_encode([b"ab"] * n_suf))
        ns.append(n_suf)
    pages.append(_announce(pagegen.dba_encode(_sorted_keys(rng, 700)), (1 << 34) + 5))
    ns.append(700)
    # fewer values asked for than the page holds: the page may only use the slots of the values it returns
    for n, ask in ((600, 300), (3000, 1), (3000, 1501), (5000, 4999)):
        pages.append(pagegen.dba_encode(_sorted_keys(rng, n)))
        ns.append(ask)
    _check("dba", orc.dba_decode, pages, ns)
    outs, status = _check("dba", orc.dba_decode, pages, ns, align=16, lead=9)
    # the sizes pass reports the same counts and byte totals (slots: num_values + 1 per page, the contract of the
    # scratch buffer, whatever the page turns out to hold)
    vcaps = [int(n) + 2 for n in ns]
    bcaps = [int(st["out_bytes"]) + 5 for st in status]
    _, st2 = run_strings("dba_sizes", pages, ns, vcaps, bcaps)
    for a, b in zip(status, st2):
        assert a["code"] == b["code"]
        if a["code"] == 0:
            assert a["values_decoded"] == b["values_decoded"] and a["out_bytes"] == b["out_bytes"]
    # a slot that is too small is reported, never overrun
    _, st3 = run_strings("dba", pages[:3], ns[:3], vcaps[:3], [max(0, int(status[i]["out_bytes"]) - 1) for i in range(3)])
    for i in range(3):
        if status[i]["code"] == 0 and status[i]["out_bytes"] > 0:
            assert st3[i]["code"] == 23 and st3[i]["out_bytes"] == status[i]["out_bytes"]


def _chain(rng, n, lens, keep):
    """n strings; string i keeps keep(i, len(previous)) leading bytes of its predecessor and has length lens(i)."""
    out, prev = [], b""
    for i in range(n):
        L = int(lens(i))
        k = min(int(keep(i, len(prev))), len(prev), L)
        cur = prev[:k] + bytes(rng.integers(97, 123, size=L - k, dtype=np.uint8))
        out.append(cur)
        prev = cur
    return out


def test_dba_pages_split_over_many_wavefronts():
    """Pages of tens of thousands of strings are assembled by many wavefronts, each of which first rebuilds what it
    inherits from the strings before its run: shared prefixes that reach back to the page's first string, that
    are longer than the on-chip copy of a string, that shrink and grow in steps, long strings among short ones,
    empty strings."""
    rng = np.random.default_rng(2222)
    pages = []
    # the whole page shares 700 bytes (every run reaches back to string 0 through the table of tile minima)
    pages.append([b"x" * 700 + b"%06d" % i for i in range(9000)])
    # ... 1500 bytes: longer than what a wavefront keeps on chip
    pages.append([b"y" * 1500 + b"%05d" % i for i in range(5000)])
    # two leading bytes shared by everything, the rest like sorted keys
    pages.append(_chain(rng, 40000, lambda i: rng.integers(8, 25), lambda i, pl: max(2, int(pl * 0.6) + int(rng.integers(-2, 3)))))
    # prefix lengths that fall in steps and jump back up: inherited bytes come from several distant strings
    pages.append(_chain(rng, 30000, lambda i: 40 + (i % 7), lambda i, pl: 38 - (i % 4099) // 120 if i % 4099 else 0))
    # medium strings (several column batches), and a very long string every few thousand
    pages.append(_chain(rng, 12000, lambda i: 3000 if i % 2500 == 1234 else rng.integers(60, 300),
                        lambda i, pl: int(pl * 0.8)))
    # empty strings and single bytes
    pages.append(_chain(rng, 20000, lambda i: rng.integers(0, 3), lambda i, pl: rng.integers(0, 3)))
    pages.append([b""] * 5000)
    # no shared bytes at all
    pages.append(_rand_strings(rng, 20000, 12))
    enc = [pagegen.dba_encode(p) for p in pages]
    ns = [len(p) for p in pages]
    outs, _ = _check("dba", orc.dba_decode, enc, ns)
    for (o, b), p in zip(outs, pages):
        assert _split(o, b) == p
    _check("dba", orc.dba_decode, enc, ns, align=16, lead=3)
    # fewer values than the page holds: the last run ends in the middle of the page
    _check("dba", orc.dba_decode, enc, [n // 2 + 17 for n in ns])
    # a prefix longer than the string before it, deep inside a page
    bad = list(pages[2])
    pre = [0] + [min(len(a), len(b), next((k for k in range(min(len(a), len(b))) if a[k] != b[k]), min(len(a), len(b))))
                 for a, b in zip(bad, bad[1:])]
    pre = np.asarray(pre, dtype=np.int32)
    suf = np.asarray([len(x) for x in bad], dtype=np.int32) - pre
    data = np.frombuffer(b"".join(x[p:] for x, p in zip(bad, pre)), dtype=np.uint8)
    pre[31337] = len(bad[31336]) + 1
    _check("dba", orc.dba_decode, [pagegen.dba_encode_arrays(pre, suf, data)], [len(bad)])


def test_plain_byte_array_and_flba():
    rng = np.random.default_rng(23)
    pages, ns = [], []
    for it in range(40):
        n = int(rng.choice([0, 1, 2, 5, 63, 64, 65, 1000, 20000]))
        strs = _rand_strings(rng, n, int(rng.choice([0, 3, 24, 300, 5000])) if n <= 1000 else 24)
        pg = pagegen.plain_byte_array_encode(strs)
        if it % 5 == 4 and len(pg) > 8:
            pg = pg[: len(pg) - int(rng.integers(1, 8))]
        pages.append(pg)
        ns.append(int(rng.choice([n, n + 7, n // 2])))
    pages += [bytes([5, 0, 0, 0]) + b"ab", bytes([1, 0, 0]), b""]
    ns += [10, 10, 10]
    _check("plain_ba", orc.plain_byte_array, pages, ns)
    _check("plain_ba", orc.plain_byte_array, pages, ns, align=16, lead=2)
    # FIXED_LEN_BYTE_ARRAY
    pages, ns, lens = [], [], []
    for it in range(30):
        L = int(rng.choice([0, 1, 2, 7, 16, 33]))
        nbytes = int(rng.choice([0, 1, L * 5, L * 100 + 3, 4096]))
        pages.append(bytes(rng.integers(0, 256, size=nbytes, dtype=np.uint8)))
        ns.append(int(rng.choice([0, 3, 5, 100, 10**6])))
        lens.append(L)
    expect = []
    for pg, n, L in zip(pages, ns, lens):
        try:
            expect.append((orc.plain_flba(pg, L, n), None))
        except orc.OracleError as e:
            expect.append((None, e))
    outs, status = run_strings("flba", pages, ns, list(ns), [len(p) + 1 for p in pages], flba_len=lens)
    for i, ((res, err), (o, b)) in enumerate(zip(expect, outs)):
        st = status[i]
        if err is not None:
            assert st["code"] == err.code and capi.format_error(st) == str(err), (i, st, str(err))
        else:
            assert st["code"] == 0 and st["values_decoded"] == res[0], (i, st, res[0])
            assert np.array_equal(b, res[1]), i
