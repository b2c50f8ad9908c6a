"""Seeded synthetic workloads of the benchmark configurations, shared by bench.py and the parity tests so that
both run the same inputs.  Generators only: every function returns encoded pages AND the values they were made
from (the ground truth of an encode -> decode round trip); nothing here decodes.
"""
import numpy as np

from . import (ENC_RLE_DICTIONARY, bit_width, data_page_v1, dba_encode_arrays, delta_encode, dict_page_values_encode,
               dictionary_page)


def delta_pages(n_pages, page_values, seed, block=256, nmb=8):
    """INT64 column, per page a random walk with steps uniform in [-100, 3996): (pages as uint8 arrays, values)."""
    rng = np.random.default_rng(seed)
    pages = []
    values = np.empty(n_pages * page_values, dtype=np.int64)
    for p in range(n_pages):
        steps = rng.integers(-100, 3996, size=page_values, dtype=np.int64)
        steps[0] = rng.integers(-2**40, 2**40)
        v = np.cumsum(steps)
        values[p * page_values:(p + 1) * page_values] = v
        pages.append(delta_encode(v, block, nmb, as_array=True))
    return pages, values


def run_values(rng, n, w, lo=8, hi=107):
    """n values of width w in alternating repeated / literal runs with lengths uniform in [lo, hi]."""
    mean = (lo + hi) / 2
    k = int(n / mean * 1.2) + 16
    lens = rng.integers(lo, hi + 1, size=k)
    while lens.sum() < n:
        lens = np.concatenate([lens, rng.integers(lo, hi + 1, size=k)])
    run_id = np.repeat(np.arange(len(lens)), lens)[:n]
    mx = (1 << w) - 1
    rep_vals = rng.integers(0, mx + 1, size=len(lens), dtype=np.uint64).astype(np.uint32)
    lit = rng.integers(0, mx + 1, size=n, dtype=np.uint64).astype(np.uint32)
    return np.where(run_id % 2 == 0, rep_vals[run_id], lit)


def dict_index_width(dsize):
    """Index width the reference's dictionary encoder writes: bit_width(cardinality), so 17 bits for exactly
    65536 entries (src/encoding.cc:769-771); bit_width(cardinality - 1) otherwise, as other writers do."""
    return 17 if dsize == 65536 else bit_width(dsize - 1)


def dict_chunk_pages(rng, dsize, dist, max_def, n, distinct):
    """`distinct` v1 data pages of n levels each over a dsize-entry INT32 dictionary.
    Returns (dictionary page bytes, [data page bytes], dictionary, [values per page], [def levels per page])."""
    w = dict_index_width(dsize)
    dictionary = rng.integers(-2**31, 2**31 - 1, size=dsize).astype(np.int32)
    pages, truth_vals, truth_def = [], [], []
    for _ in range(distinct):
        dl = (rng.random(n) >= 0.1).astype(np.uint32) if max_def else np.ones(n, dtype=np.uint32)
        nn = int(dl.sum())
        if dist == "uniform":
            idx = rng.integers(0, dsize, size=nn).astype(np.uint32)
        else:
            idx = np.minimum(rng.zipf(1.1, size=nn) - 1, dsize - 1).astype(np.uint32)
        pages.append(data_page_v1(dict_page_values_encode(idx, w), n, ENC_RLE_DICTIONARY, def_levels=dl, max_def=max_def))
        truth_vals.append(dictionary[idx])
        truth_def.append(dl.astype(np.int32) if max_def else np.zeros(n, dtype=np.int32))
    return dictionary_page(dictionary.tobytes(), dsize), pages, dictionary, truth_vals, truth_def


def string_page(rng, n):
    """n sorted-key-like strings: lengths 8-24, each sharing about 60% of its bytes with its predecessor.
    Returns (lengths, prefix lengths, suffix lengths, suffix bytes)."""
    lens = rng.integers(8, 25, size=n).astype(np.int64)
    want = np.floor(lens * 0.6 + rng.integers(-2, 3, size=n)).astype(np.int64)
    prev = np.concatenate([[0], lens[:-1]])
    prefix = np.clip(want, 0, np.minimum(prev, lens))
    prefix[0] = 0
    suffix = lens - prefix
    data = rng.integers(97, 123, size=int(suffix.sum()), dtype=np.uint8)
    return lens, prefix.astype(np.int32), suffix.astype(np.int32), data


def assemble_strings(prefix, suffix, data):
    """The strings a (prefix lengths, suffix lengths, suffix bytes) triple spells: (int64 offsets[n + 1], bytes).
    value_i = value_{i-1}[0:prefix_i] ++ suffix_i."""
    n = len(prefix)
    lens = prefix.astype(np.int64) + suffix.astype(np.int64)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    out = bytearray(int(offs[-1]))
    raw = data.tobytes()
    spos = 0
    for i in range(n):
        o, p, s = int(offs[i]), int(prefix[i]), int(suffix[i])
        if p:
            prev = int(offs[i - 1])
            out[o:o + p] = out[prev:prev + p]
        out[o + p:o + p + s] = raw[spos:spos + s]
        spos += s
    return offs, np.frombuffer(bytes(out), dtype=np.uint8)


def string_pages(rng, n, distinct):
    """`distinct` pages of n strings, each both as DELTA_BYTE_ARRAY and as DELTA_LENGTH_BYTE_ARRAY.
    Returns a list of dicts {dba, dlba, offsets, bytes}."""
    out = []
    for _ in range(distinct):
        lens, prefix, suffix, data = string_page(rng, n)
        offs, full = assemble_strings(prefix, suffix, data)
        out.append({"dba": dba_encode_arrays(prefix, suffix, data),
                    "dlba": delta_encode(lens.astype(np.int32), 128, 4) + full.tobytes(),
                    "offsets": offs, "bytes": full})
    return out
