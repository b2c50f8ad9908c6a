"""GPU parity at the sizes BASELINE.json names, through properties that need no CPU decode of the whole
input: encode -> GPU decode round trips against the generator's own column, plus the oracle on a sample
of pages.  Bit-exact."""
import numpy as np
import pytest

from oracle import oracle as orc
from parquet4seastar_amd import capi, pagegen
from tests.gpu_util import ctx

pytestmark = pytest.mark.gpu


def _decode_fixed(fn, pages, n_per_page, elem, extra, param0=0):
    c = ctx()
    buf, offs, sizes = capi.pack_pages(pages, align=16)
    descs = capi.make_descs(offs, sizes, [i * n_per_page for i in range(len(pages))], n_per_page, param0)
    d_in, d_pages = c.to_device(buf), c.to_device(descs.view(np.uint8))
    d_status, d_out = c.alloc(48 * len(pages)), c.alloc(len(pages) * n_per_page * elem)
    c._check(getattr(capi.lib(), fn)(c.handle, d_in.ptr, buf.size, d_pages.ptr, len(pages), d_out.ptr, d_status.ptr, *extra))
    c.sync()
    status = d_status.download(capi.PAGE_STATUS_DTYPE, len(pages))
    capi.raise_on_page_errors(status)
    assert int(status["values_decoded"].sum()) == len(pages) * n_per_page
    out = d_out.download(np.uint8, len(pages) * n_per_page * elem)
    for b in (d_in, d_pages, d_status, d_out):
        b.free()
    return out


def test_config1_rle_one_million_values_every_width():
    """1Mi int32 values per page, bit widths 1-20, runs of 8-107 values (the reference's rle_encoding_test path)."""
    rng = np.random.default_rng(1)
    n = 1 << 20
    for w in range(1, 21):
        lens = rng.integers(8, 108, size=n // 8)
        run_id = np.repeat(np.arange(len(lens)), lens)[:n]
        rep_vals = rng.integers(0, 1 << w, size=len(lens), dtype=np.uint64).astype(np.uint32)
        lit = rng.integers(0, 1 << w, size=n, dtype=np.uint64).astype(np.uint32)
        vals = np.where(run_id % 2 == 0, rep_vals[run_id], lit)
        page = pagegen.rle_encode(vals, w)
        got = _decode_fixed("p4s_decode_rle", [page], n, 4, (4,), param0=w).view(np.uint32)
        assert np.array_equal(got, vals), w
        assert np.array_equal(orc.rle_decode(page, w, n, 4).view(np.uint32), vals), w


def test_config2_delta_int64_256m_values():
    """256Mi int64 values in 2048 pages of 131072 (1 MiB decoded each): a running sum checked against numpy."""
    rng = np.random.default_rng(2)
    n_pages, n = 2048, 131072
    distinct = 64                      # distinct pages, each decoded 32 times at different addresses
    cols = []
    pages = []
    for p in range(distinct):
        steps = rng.integers(-100, 3996, size=n, dtype=np.int64)
        steps[0] = rng.integers(-2**40, 2**40)
        v = np.cumsum(steps)
        cols.append(v)
        pages.append(pagegen.delta_encode(v, 256, 8))
    assert np.array_equal(orc.delta_decode(pages[3], n, 8), cols[3])
    out = _decode_fixed("p4s_decode_delta_binary_packed", pages * (n_pages // distinct), n, 8, (8,)).view(np.int64)
    truth = np.concatenate(cols)
    out = out.reshape(n_pages // distinct, -1)
    for r in range(out.shape[0]):
        assert np.array_equal(out[r], truth), r


@pytest.mark.parametrize("dtype,elem", [(np.uint32, 4), (np.uint64, 8)])
def test_config3_byte_stream_split_512m_values_per_8_gpus(dtype, elem):
    """One GPU's share of the 512Mi-value column sharded over 8 GPUs: 64Mi values, 1 MiB encoded pages."""
    rng = np.random.default_rng(3)
    n = (1 << 20) // elem
    n_pages = (64 << 20) // n
    distinct = 32
    vals = [rng.integers(0, np.iinfo(dtype).max, size=n, dtype=dtype) for _ in range(distinct)]   # raw bit patterns
    pages = [pagegen.bss_encode(v) for v in vals]
    assert np.array_equal(orc.bss_decode(pages[0], elem, n).view(dtype), vals[0])
    out = _decode_fixed("p4s_decode_byte_stream_split", pages * (n_pages // distinct), n, elem, (elem,)).view(dtype)
    truth = np.concatenate(vals)
    out = out.reshape(n_pages // distinct, -1)
    for r in range(out.shape[0]):
        assert np.array_equal(out[r], truth), r
