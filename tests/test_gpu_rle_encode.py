"""GPU parity: the RLE / bit-packed hybrid ENCODER through the C ABI against the oracle's restatement of the reference's
RleEncoder (oracle/p4s_oracle.c orc_rle_encode, itself checked against the reference's own encoder in
tests/test_oracle.py), byte for byte, and decoded back on the GPU."""
import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import capi
from tests.gpu_util import run_fixed

pytestmark = pytest.mark.gpu


def _shapes(rng, w, n):
    hi = (1 << w) if w < 32 else (1 << 32)
    lo3 = min(hi, 3)
    yield rng.integers(0, hi, size=n, dtype=np.uint64)                                             # literals only
    yield np.repeat(rng.integers(0, hi, size=n // 5 + 1, dtype=np.uint64), rng.integers(1, 20, size=n // 5 + 1))[:n]
    yield np.repeat(rng.integers(0, lo3, size=n // 3 + 1, dtype=np.uint64), rng.integers(1, 12, size=n // 3 + 1))[:n]
    yield np.full(n, rng.integers(0, hi), dtype=np.uint64)                                          # one run
    # repeated runs of exactly 8, 9, 7 values between literals; a run that starts in the middle of a group
    parts = [np.zeros(0, dtype=np.uint64)]
    while sum(len(p) for p in parts) < n:
        parts.append(np.full(int(rng.choice([7, 8, 9, 15, 16, 17, 64, 513])), rng.integers(0, hi), dtype=np.uint64))
        parts.append(rng.integers(0, hi, size=int(rng.integers(0, 20)), dtype=np.uint64))
    yield np.concatenate(parts)[:n]


def _check(values_list, widths, **kw):
    want = [orc.rle_encode(v, w) for v, w in zip(values_list, widths)]
    flat = np.concatenate([np.asarray(v, dtype=np.uint32) for v in values_list]) if values_list else np.zeros(0, np.uint32)
    with capi.Context(0) as ctx:
        got, st = capi.encode_rle(ctx, flat, [len(v) for v in values_list], widths, guard=0xA5, **kw)
    for i, (g, x) in enumerate(zip(got, want)):
        assert st[i]["code"] == 0, (i, st[i])
        assert st[i]["values_decoded"] == len(values_list[i]) and st[i]["out_bytes"] == len(x), (i, st[i], len(x))
        assert g == x, "page %d (width %d, %d values) differs at byte %d" % (
            i, widths[i], len(values_list[i]), next((k for k in range(min(len(g), len(x))) if g[k] != x[k]), -1))
    return got


@pytest.mark.parametrize("w", [0, 1, 2, 3, 5, 7, 8, 9, 12, 16, 17, 20, 24, 31, 32])
def test_rle_encoder_matches_oracle(w):
    rng = np.random.default_rng(1000 + w)
    vals, widths = [], []
    for n in [0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 503, 504, 505, 512, 2040, 2047, 2048, 2049, 2055, 2056, 4096, 5000, 20000]:
        for v in _shapes(rng, w, n):
            vals.append(v.astype(np.uint32))
            widths.append(w)
    _check(vals, widths)


def test_rle_encoder_long_runs_and_chunk_edges():
    """Runs that cross many chunks of 2048 values, literal runs of more than 63 groups, groups that straddle a chunk
    boundary (a repeated run of odd length in front of them), pages that end inside a run / a group."""
    rng = np.random.default_rng(7)
    vals, widths = [], []
    for w in (1, 4, 12, 20):
        hi = 1 << w
        for lead in range(0, 9):
            # `lead` equal values (a short run or part of a literal group), then literals: every alignment of the groups
            v = np.concatenate([np.full(8 + lead, 1, dtype=np.uint64), rng.integers(0, hi, size=6000, dtype=np.uint64)])
            vals.append(v.astype(np.uint32)); widths.append(w)
        v = np.concatenate([rng.integers(0, hi, size=100, dtype=np.uint64), np.full(100000, hi - 1, dtype=np.uint64),
                            rng.integers(0, hi, size=3, dtype=np.uint64), np.full(5, 1, dtype=np.uint64)])
        vals.append(v.astype(np.uint32)); widths.append(w)
        vals.append(np.full(2048 * 3, 1, dtype=np.uint32)); widths.append(w)
        vals.append(rng.integers(0, hi, size=2048 * 3 + 5, dtype=np.uint64).astype(np.uint32)); widths.append(w)
    got = _check(vals, widths)
    # and back through the GPU decoder
    ns = [len(v) for v in vals]
    outs, status = run_fixed("p4s_decode_rle", got, ns, 4, [n + 2 for n in ns], param0=widths, extra=(4,))
    for o, v, s in zip(outs, vals, status):
        assert s["code"] == 0 and np.array_equal(o.view(np.uint32), v)


def test_rle_encoder_level_like_and_benchmark_shapes():
    rng = np.random.default_rng(8)
    n = 1 << 18
    vals = [(rng.random(n) > 0.1).astype(np.uint32),                                    # definition levels, 10 % nulls
            np.ones(n, dtype=np.uint32),                                                # no nulls: one run
            np.repeat(rng.integers(0, 1 << 12, size=n // 40 + 1, dtype=np.uint64), rng.integers(8, 108, size=n // 40 + 1))[:n].astype(np.uint32)]
    _check(vals, [1, 1, 12])


def test_rle_encoder_slot_too_small_and_bad_values():
    rng = np.random.default_rng(9)
    v = rng.integers(0, 1 << 12, size=5000, dtype=np.uint64).astype(np.uint32)
    need = len(orc.rle_encode(v, 12))
    with capi.Context(0) as ctx:
        got, st = capi.encode_rle(ctx, np.concatenate([v, v, v]), [5000, 5000, 5000], [12, 12, 12], caps=[need, need - 1, 100], guard=0x5A)
        assert st[0]["code"] == 0 and got[0] == orc.rle_encode(v, 12)
        for i in (1, 2):
            assert st[i]["code"] == 23 and st[i]["aux0"] == need and got[i] is None, st[i]
        bad = v.copy(); bad[4000] = 1 << 12
        got, st = capi.encode_rle(ctx, bad, [5000], [12], guard=0x5A)
        assert st[0]["code"] == 35 and "bit width 12" in capi.format_error(st[0])
        got, st = capi.encode_rle(ctx, v, [5000], [33], guard=0x5A)
        assert st[0]["code"] == 35
