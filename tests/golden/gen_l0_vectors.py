#!/usr/bin/env python3
"""Writes tests/golden/l0_ref_vectors.json by RUNNING the reference's own
header-only bit-level layer (oracle/_ref/libp4s_ref_l0.so, built from
/root/reference by oracle/Makefile) on seeded inputs: RleDecoder::GetBatch on
valid, hand-shaped and corrupted streams, BitReader::GetBatch, and scripted
BitReader sessions.  The JSON holds inputs and the reference's outputs only.
Only runs where /root/reference exists; the JSON is committed.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from parquet4seastar_amd import pagegen  # noqa: E402

assert orc.ref() is not None, "oracle/_ref is not built (needs /root/reference)"
rng = np.random.default_rng(20261111)
V = []


def ref_rle(data, w, batches, es=4):
    vals, counts = orc.rle_decode_batches(data, w, batches, elem_size=es, use_ref=True)
    return [int(x) for x in vals], [int(c) for c in counts]


def add_rle(name, data, w, batches, es=4):
    vals, counts = ref_rle(data, w, batches, es)
    V.append(dict(kind="rle", name=name, bit_width=w, elem_size=es, input=bytes(data).hex(),
                  batches=list(batches), counts=counts, values=vals))


def mixed_values(n, w, rng):
    out = []
    while len(out) < n:
        ln = int(rng.integers(1, 60))
        if rng.random() < 0.5:
            out.extend([int(rng.integers(0, 1 << w)) if w else 0] * ln)
        else:
            out.extend(int(x) for x in (rng.integers(0, 1 << w, size=ln) if w else np.zeros(ln, dtype=np.int64)))
    return out[:n]


# valid streams, every width, encoded by the REFERENCE's RleEncoder, decoded in one batch
for w in range(0, 33):
    n = int(rng.integers(50, 400))
    vals = np.asarray(mixed_values(n, w, rng), dtype=np.uint64)
    cap = 8192 + n * 8   # RleEncoder wants room for a maximal run beyond what it wrote
    buf = np.zeros(cap, dtype=np.uint8)
    if w == 0:
        data = pagegen.rle_encode(vals.astype(np.uint32), 0)
    else:
        ln = orc.ref().ref_rle_encode(vals.ctypes.data, n, w, buf.ctypes.data, cap)
        assert ln > 0
        data = buf[:ln].tobytes()
    add_rle("rle_valid_w%d" % w, data, w, [n + 7])
    add_rle("rle_valid_w%d_cap" % w, data, w, [max(1, n // 2)])

# hand-shaped streams: zero-length runs, multi-byte headers, unmasked repeat value, int16/uint8 out
add_rle("rle_unmasked_repeat", bytes([0x08, 0xFF]), 3, [100])
add_rle("rle_zero_len_runs", bytes([0x00, 0x09, 0x01, 0x06, 0x07]), 4, [100])
add_rle("rle_w0_rle5", bytes([0x0A]), 0, [100])
add_rle("rle_w0_lit1", bytes([0x03]), 0, [100])
add_rle("rle_multibyte_hdr", pagegen.rle_from_runs(
    [("rle", 300, 5), ("lit", list(range(200)) * 4), ("rle", 70000, 9)], 8), 8, [100000])
add_rle("rle_int16_out", pagegen.rle_from_runs([("rle", 20, 513), ("lit", list(range(1000, 1016)))], 11),
        11, [64], es=2)
add_rle("rle_uint8_bool", pagegen.rle_from_runs([("rle", 9, 1), ("lit", [1, 0, 1, 1, 0, 0, 1, 0])], 1),
        1, [64], es=1)
add_rle("rle_value_too_wide_for_type", pagegen.rle_from_runs([("rle", 9, 300)], 9), 9, [64], es=1)
add_rle("rle_hdr_count_overflow_literal", bytes([0xFF, 0xFF, 0xFF, 0xFF, 0x0F, 0x00]), 1, [64])
add_rle("rle_5byte_header", bytes([0x90, 0x80, 0x80, 0x80, 0x00, 0x07]), 8, [64])
add_rle("rle_6byte_header_fails", bytes([0x90, 0x80, 0x80, 0x80, 0x80, 0x00, 0x07]), 8, [64])

# corrupted: every truncation of a small mixed stream, single batch and split batches
base = pagegen.rle_from_runs([("lit", [1, 2, 3, 4, 5, 6, 7, 0] * 2), ("rle", 11, 6), ("lit", [7] * 8),
                              ("rle", 5, 2)], 3)
for cut in range(len(base) + 1):
    add_rle("rle_trunc_%d" % cut, base[:cut], 3, [100])
    add_rle("rle_trunc_%d_split" % cut, base[:cut], 3, [5, 3, 9, 100])
base20 = pagegen.rle_from_runs([("lit", [int(x) for x in rng.integers(0, 1 << 20, size=24)]), ("rle", 9, 77777)], 20)
for cut in range(0, len(base20) + 1, 3):
    add_rle("rle20_trunc_%d" % cut, base20[:cut], 20, [100])

# random garbage streams
for k in range(40):
    w = int(rng.integers(0, 33))
    data = bytes(int(x) for x in rng.integers(0, 256, size=int(rng.integers(0, 40))))
    # keep run counts small so that outputs stay small: clear the high bits of every byte
    data = bytes(b & 0x3F for b in data)
    add_rle("rle_garbage_%d" % k, data, w, [4096])

# BitReader::GetBatch (BIT_PACKED levels path), incl. clamping at the end of the buffer
for w in range(0, 33):
    n = int(rng.integers(1, 100))
    vals = rng.integers(0, 1 << w, size=n) if w else np.zeros(n, dtype=np.int64)
    data = pagegen.bitpack(vals.astype(np.uint32), w)
    got = orc.bitpacked_decode(data, w, n + 50, use_ref=True)
    V.append(dict(kind="bitpacked", name="bitpacked_w%d" % w, bit_width=w, elem_size=4, input=data.hex(),
                  n=n + 50, values=[int(x) for x in got]))

# scripted BitReader sessions on random bytes
for k in range(60):
    data = bytes(int(x) for x in rng.integers(0, 256, size=int(rng.integers(0, 48))))
    ops = []
    for _ in range(int(rng.integers(1, 12))):
        op = int(rng.integers(0, 7))
        arg = int(rng.integers(0, 65)) if op == 0 else (int(rng.integers(0, 10)) if op == 1 else 0)
        ops.append([op, arg])
    res = orc.br_script(data, ops, use_ref=True)
    V.append(dict(kind="br_script", name="br_script_%d" % k, input=data.hex(), ops=ops,
                  results=[[int(a), int(b)] for a, b in res]))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "l0_ref_vectors.json")
with open(out, "w") as f:
    json.dump(V, f)
print("wrote", out, len(V), "vectors")
