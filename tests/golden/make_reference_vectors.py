#!/usr/bin/env python3
"""Writes tests/golden/reference_vectors.json: the known-answer vectors that the
reference's own unit tests hold for the page-decoding path, transcribed as data
(input bytes, parameters, expected outputs) with the test file:line each comes
from.  Paths are relative to the reference tree.  Run: python make_reference_vectors.py
"""
import json
import os
import struct

V = []


def hx(b):
    return bytes(b).hex()


# ---- tests/rle_encoding_test.cc -------------------------------------------------
V.append(dict(
    name="BitReader_happy", source="tests/rle_encoding_test.cc:37-76", kind="bitreader_script",
    input=hx([0b10001000, 0b01000110, 0b10000000, 0b00000001, 0b11111111, 0b00000001]),
    # (op, arg, n): GetBatch(w=3) of 2 then 3 values, GetVlqInt(u32), GetZigZagVlqInt(i32), GetBatch
    steps=[
        dict(op="get_batch", bits=3, n=2, expect=[0, 1]),
        dict(op="get_batch", bits=3, n=3, expect=[2, 3, 4]),
        dict(op="vlq32", ok=True, expect=128),
        dict(op="zigzag32", ok=True, expect=-128),
        dict(op="get_batch", bits=3, n=999999, expect=[]),
    ]))
V.append(dict(name="BitReader_ULEB128_corrupted", source="tests/rle_encoding_test.cc:78-87",
              kind="bitreader_script", input=hx([0x80]), steps=[dict(op="vlq32", ok=False)]))
V.append(dict(name="BitReader_ULEB128_overflow", source="tests/rle_encoding_test.cc:89-98",
              kind="bitreader_script", input=hx([0x80] * 6 + [0x00]), steps=[dict(op="vlq32", ok=False)]))
V.append(dict(name="BitReader_zigzag_corrupted", source="tests/rle_encoding_test.cc:100-109",
              kind="bitreader_script", input=hx([0x80]), steps=[dict(op="zigzag32", ok=False)]))
V.append(dict(name="BitReader_zigzag_overflow", source="tests/rle_encoding_test.cc:111-120",
              kind="bitreader_script", input=hx([0x80] * 6 + [0x00]), steps=[dict(op="zigzag32", ok=False)]))

V.append(dict(
    name="RleDecoder_happy", source="tests/rle_encoding_test.cc:125-155", kind="rle",
    bit_width=3, elem_size=4,
    input=hx([0b00000011, 0b10001000, 0b11000110, 0b11111010, 0b00001000, 0b00000101]),
    batches=[6, 4, 9999999, 9999999],
    expect_batches=[[0, 1, 2, 3, 4, 5], [6, 7, 5, 5], [5, 5], []]))
_lit512 = bytes([0b10000001, 0b00000001]) + b"".join(struct.pack("<H", i) for i in range(512))
V.append(dict(
    name="RleDecoder_bit_packed_ULEB128", source="tests/rle_encoding_test.cc:157-176", kind="rle",
    bit_width=16, elem_size=4, input=hx(_lit512), batches=[9999999], expect_batches=[list(range(512))]))
V.append(dict(
    name="RleDecoder_rle_ULEB128", source="tests/rle_encoding_test.cc:178-195", kind="rle",
    bit_width=8, elem_size=4, input=hx([0b10000000, 0b00000001, 0b00000101]),
    batches=[9999999], expect_batches=[[5] * 64]))
V.append(dict(
    name="RleDecoder_bit_packed_too_short", source="tests/rle_encoding_test.cc:197-207", kind="rle",
    bit_width=3, elem_size=4, input=hx([0b00000011, 0b10001000, 0b11000110]),
    batches=[8], expect_batches=[[]]))
V.append(dict(
    name="RleDecoder_rle_too_short", source="tests/rle_encoding_test.cc:209-219", kind="rle",
    bit_width=3, elem_size=4, input=hx([0b00001000]), batches=[4], expect_batches=[[]]))
V.append(dict(
    name="RleDecoder_bit_packed_ULEB128_too_short", source="tests/rle_encoding_test.cc:221-231", kind="rle",
    bit_width=3, elem_size=4, input=hx([0b10000001]), batches=[512], expect_batches=[[]]))
V.append(dict(
    name="RleDecoder_rle_ULEB128_too_short", source="tests/rle_encoding_test.cc:233-243", kind="rle",
    bit_width=3, elem_size=4, input=hx([0b10000000]), batches=[512], expect_batches=[[]]))

# ---- tests/delta_binary_packed_test.cc:30-90 -------------------------------------
_delta_hdr = bytes([0x80, 0x01, 0x04, 0x42, 0x10])
_delta_blk = bytes([0x01, 0x04, 0x03, 0x02, 0x01]) + bytes([
    0b00010001, 0b00010001, 0b00010001, 0b00010001,
    0, 0, 0, 0,
    0, 0, 0, 0,
    0b00011001, 0b00010001, 0b00010001, 0b00010001,

    0b01001001, 0b10010010, 0b00100100, 0b01001001,
    0b10010010, 0b00100100, 0b01001001, 0b10010010,
    0b00100100, 0b01001001, 0b10010010, 0b00100100,
    0b01001001, 0b10010010, 0b00100100, 0b01001001,

    0b11111101, 0xFF, 0xFF, 0xFF,
    0xFF, 0xFF, 0xFF, 0xFF,

    0xFF, 0xFF, 0xFF, 0xFF,
])
_delta_expected = ([8] + [8] * 8 + [7, 6, 5, 4, 3, 2, 1, 0] + [-1, -2, -3, -4, -5, -6, -7, -8] + [0] * 8
                   + [0] * 32 + [0])
V.append(dict(
    name="delta_binary_packed_decoding", source="tests/delta_binary_packed_test.cc:30-90", kind="delta",
    elem_size=4, n=10000, input=hx(_delta_hdr + _delta_blk), expect=_delta_expected))

# round-trip inputs of tests/delta_binary_packed_test.cc:92-188 (the values; the encoded bytes
# come from this repository's own encoder because the reference encoder's padding bits are
# uninitialised memory)
V.append(dict(
    name="delta_roundtrip_int32", source="tests/delta_binary_packed_test.cc:92-126", kind="delta_roundtrip",
    elem_size=4,
    values=list(range(1337)) + [-2**31, 2**31 - 1, -2**31, 2**31 - 1] + [i * i for i in range(420)]))
V.append(dict(
    name="delta_roundtrip_int64", source="tests/delta_binary_packed_test.cc:128-162", kind="delta_roundtrip",
    elem_size=8,
    values=list(range(1337)) + [-2**63, 2**63 - 1, -2**63, 2**63 - 1] + [i * i for i in range(-420, 420)]))
V.append(dict(
    name="delta_roundtrip_int64_empty", source="tests/delta_binary_packed_test.cc:164-188",
    kind="delta_roundtrip", elem_size=8, values=[]))

# ---- tests/delta_length_byte_array_test.cc:33-90 ---------------------------------
_ones16 = bytes([0xFF] * 16)
_dlba = bytes([0x80, 0x01, 0x01, 0x04, 0x0a]) + bytes([0x00, 0x01]) + _ones16 \
    + b"aaaaa" + b"bbbbbb" + b"ccccccc" + b"dddddddd"
V.append(dict(
    name="delta_length_byte_array_happy", source="tests/delta_length_byte_array_test.cc:33-90", kind="dlba",
    n=10000, input=hx(_dlba), expect=[hx(b"aaaaa"), hx(b"bbbbbb"), hx(b"ccccccc"), hx(b"dddddddd")]))

# ---- tests/delta_byte_array_test.cc:33-126 ---------------------------------------
_dba_lengths = bytes([0x80, 0x01, 0x01, 0x04, 0x00]) + bytes([0x02, 0x01]) + _ones16
V.append(dict(
    name="delta_byte_array_happy", source="tests/delta_byte_array_test.cc:33-126", kind="dba",
    n=10000, input=hx(_dba_lengths + _dlba),
    expect=[hx(b"aaaaa"), hx(b"aabbbbbb"), hx(b"aabbccccccc"), hx(b"aabbccdddddddd")]))

# ---- tests/byte_stream_split_test.cc:29-102 --------------------------------------
V.append(dict(
    name="byte_stream_split_float", source="tests/byte_stream_split_test.cc:29-60", kind="bss", elem_size=4,
    n=10000, input=hx([0xa1, 0xb1, 0xc1, 0xa2, 0xb2, 0xc2, 0xa3, 0xb3, 0xc3, 0xa4, 0xb4, 0xc4]),
    expect=hx([0xa1, 0xa2, 0xa3, 0xa4, 0xb1, 0xb2, 0xb3, 0xb4, 0xc1, 0xc2, 0xc3, 0xc4])))
V.append(dict(
    name="byte_stream_split_double", source="tests/byte_stream_split_test.cc:62-97", kind="bss", elem_size=8,
    n=10000,
    input=hx([0xa1, 0xb1, 0xc1, 0xa2, 0xb2, 0xc2, 0xa3, 0xb3, 0xc3, 0xa4, 0xb4, 0xc4,
              0xa5, 0xb5, 0xc5, 0xa6, 0xb6, 0xc6, 0xa7, 0xb7, 0xc7, 0xa8, 0xb8, 0xc8]),
    expect=hx([0xa1, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
               0xb1, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8,
               0xc1, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8])))

# ---- tests/dictionary_encoder_test.cc:29-88 (decode side of what it pins) --------
# indices {0,1,0,2} at width 2 against the PLAIN dictionary {2,1,3} give back the input {2,1,2,3};
# indices {1,3,4} at width 3 against {2,1,3,4,5} give {1,4,5}.  The index bytes are produced by
# this repository's encoder (one literal group, zero padded).
V.append(dict(
    name="dictionary_int32_first_page", source="tests/dictionary_encoder_test.cc:29-59", kind="dict_roundtrip",
    bit_width=2, indices=[0, 1, 0, 2],
    dict_plain=hx(struct.pack("<3i", 2, 1, 3)), expect=[2, 1, 2, 3]))
V.append(dict(
    name="dictionary_int32_second_page", source="tests/dictionary_encoder_test.cc:60-87", kind="dict_roundtrip",
    bit_width=3, indices=[1, 3, 4],
    dict_plain=hx(struct.pack("<5i", 2, 1, 3, 4, 5)), expect=[1, 4, 5]))

# ---- tests/column_chunk_writer_test.cc:41-109 (expected triplets) ----------------
V.append(dict(
    name="column_chunk_triplets", source="tests/column_chunk_writer_test.cc:58-107", kind="chunk_triplets",
    def_levels=[1, 0, 1, 1, 0, 1], rep_levels=[1, 1, 1, 1, 1, 1], max_def=1, max_rep=1,
    type_length=1, values=[hx(b"a"), hx(b"c"), hx(b"a"), hx(b"e")],
    page_split=[3, 3]))

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")
with open(out, "w") as f:
    json.dump(V, f, indent=1)
print("wrote", out, len(V), "vectors")
