"""GPU check of the level-driven scatter (dense values + definition levels -> positional column + validity
bitmap) against a numpy statement of the reference consumer's per-value rule (a value is present where
def == max_def, include/parquet4seastar/record_reader.hh:449-466).  The reference has no bulk form of this
step and none of its tests holds a vector for it: parity unpinned beyond that rule.  Bit-exact."""
import numpy as np
import pytest

from parquet4seastar_amd import capi
from tests.gpu_util import ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("elem,dtype", [(4, np.int32), (8, np.int64), (1, np.uint8), (12, None)])
def test_scatter_by_levels(elem, dtype):
    rng = np.random.default_rng(elem)
    c = ctx()
    L = capi.lib()
    n_levels = [0, 1, 31, 32, 33, 63, 64, 65, 255, 256, 1023, 1024, 1025, 5000, 70001]
    max_defs = [1, 0, 3, 1, 2, 1, 1, 7, 1, 1, 1, 2, 1, 1, 3]
    short = {13}                     # this page gets fewer dense values than its levels announce
    slots, words = 0, 0
    descs = np.zeros(len(n_levels), dtype=capi.PAGE_DESC_DTYPE)
    defs, dense, expect_out, expect_bits, expect_nn = [], [], [], [], []
    for i, (n, md) in enumerate(zip(n_levels, max_defs)):
        d = rng.integers(0, md + 1, size=n).astype(np.int32) if md else np.zeros(n, dtype=np.int32)
        if i % 2:
            d[rng.random(n) < 0.8] = md
        valid = d == md
        nn = int(valid.sum())
        v = rng.integers(1, 255, size=(nn, elem), dtype=np.uint8)
        have = nn - 5 if i in short else nn
        out = np.zeros((n, elem), dtype=np.uint8)
        idx = np.flatnonzero(valid)
        out[idx[:have]] = v[:have]
        bits = np.zeros((n + 31) // 32 * 32, dtype=np.uint8)
        bits[:n] = valid
        descs[i] = (0, 0, slots, words, n, md, have, 0)
        full = np.zeros((n, elem), dtype=np.uint8)
        full[:nn] = v
        defs.append(d); dense.append(full); expect_out.append(out)
        expect_bits.append(np.packbits(bits, bitorder="little").view(np.uint32)); expect_nn.append(nn)
        slots += n
        words += (n + 31) // 32
    d_def = c.to_device(np.concatenate(defs))
    d_vals = c.to_device(np.concatenate(dense))
    d_pages = c.to_device(descs.view(np.uint8))
    d_out = c.alloc(slots * elem + 16); d_out.fill(0xCD)
    d_bits = c.alloc(words * 4 + 16); d_bits.fill(0xCD)
    d_status = c.alloc(48 * len(n_levels))
    c._check(L.p4s_scatter_by_levels(c.handle, d_def.ptr, d_vals.ptr, d_pages.ptr, len(n_levels), elem, d_out.ptr,
                                     d_bits.ptr, d_status.ptr))
    c.sync()
    status = d_status.download(capi.PAGE_STATUS_DTYPE, len(n_levels))
    got = d_out.download(np.uint8, slots * elem + 16)
    bits = d_bits.download(np.uint32, words + 4)
    assert np.all(got[slots * elem:] == 0xCD) and np.all(bits[words:] == 0xCDCDCDCD)
    so, wo = 0, 0
    for i, n in enumerate(n_levels):
        st = status[i]
        assert st["code"] == 0 and st["values_decoded"] == n and st["aux0"] == expect_nn[i], (i, st)
        assert st["aux1"] == (1 if i in short and expect_nn[i] > 0 else 0), (i, st)
        assert np.array_equal(got[so * elem:(so + n) * elem].reshape(n, elem), expect_out[i]), i
        assert np.array_equal(bits[wo:wo + (n + 31) // 32], expect_bits[i]), i
        so += n
        wo += (n + 31) // 32
    for b in (d_def, d_vals, d_pages, d_out, d_bits, d_status):
        b.free()
