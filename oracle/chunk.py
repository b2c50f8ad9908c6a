"""CPU restatement of the reference's column-chunk page walk, on top of the C oracle's decoders.

TEST INFRASTRUCTURE ONLY (imported by tests/ and by bench.py's verification legs): the product path never
comes here.  Follows, as text, page_reader::next_page (src/column_chunk_reader.cc:27-50),
column_chunk_reader<T>::load_data_page / load_data_page_v2 / load_dictionary_page
(src/column_chunk_reader.cc:52-138) and read_batch_internal (include/parquet4seastar/column_chunk_reader.hh:101-146).
The Thrift compact-protocol reader covers what a PageHeader holds (doc/parquet/parquet.thrift:507-659).
"""
import numpy as np

from . import oracle as orc

# format::Type, format::Encoding, format::PageType (include/parquet4seastar/parquet_types.h:24-35,90-102,128-135)
BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY, FLBA = range(8)
PLAIN, PLAIN_DICTIONARY, RLE, BIT_PACKED, DELTA_BINARY_PACKED, DLBA, DBA, RLE_DICTIONARY, BSS = 0, 2, 3, 4, 5, 6, 7, 8, 9
DATA_PAGE, INDEX_PAGE, DICTIONARY_PAGE, DATA_PAGE_V2 = 0, 1, 2, 3

_FIXED = {INT32: np.dtype("<i4"), INT64: np.dtype("<i8"), INT96: np.dtype(("<i4", 3)), FLOAT: np.dtype("<f4"),
          DOUBLE: np.dtype("<f8")}


class ChunkError(Exception):
    pass


def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise ChunkError("truncated thrift varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return v, pos


def _zigzag(v):
    return (v >> 1) ^ -(v & 1)


def thrift_struct(buf, pos):
    """One compact-protocol struct -> ({field id: value}, position after it).  Nested structs become dicts,
    lists become Python lists, binary stays bytes."""
    out, fid = {}, 0
    while True:
        if pos >= len(buf):
            raise ChunkError("truncated thrift struct")
        h = buf[pos]
        pos += 1
        if h == 0:
            return out, pos
        t, delta = h & 0x0F, h >> 4
        if delta:
            fid += delta
        else:
            z, pos = _varint(buf, pos)
            fid = _zigzag(z)
        out[fid], pos = _thrift_value(buf, pos, t)


def _thrift_value(buf, pos, t):
    if t == 1:
        return True, pos
    if t == 2:
        return False, pos
    if t == 3:
        return buf[pos], pos + 1
    if t in (4, 5, 6):
        z, pos = _varint(buf, pos)
        return _zigzag(z), pos
    if t == 7:
        return bytes(buf[pos:pos + 8]), pos + 8
    if t == 8:
        n, pos = _varint(buf, pos)
        return bytes(buf[pos:pos + n]), pos + n
    if t in (9, 10):
        h = buf[pos]
        pos += 1
        n, et = h >> 4, h & 0x0F
        if n == 15:
            n, pos = _varint(buf, pos)
        items = []
        for _ in range(n):
            if et in (1, 2):            # a bool element is one byte
                items.append(buf[pos] == 1)
                pos += 1
            else:
                v, pos = _thrift_value(buf, pos, et)
                items.append(v)
        return items, pos
    if t == 12:
        return thrift_struct(buf, pos)
    raise ChunkError("unsupported thrift type %d" % t)


def read_page_header(buf, pos):
    """PageHeader at buf[pos:] -> (dict with the names of parquet.thrift, position of the page body)."""
    raw, pos = thrift_struct(buf, pos)
    h = {"type": raw.get(1), "uncompressed_page_size": raw.get(2), "compressed_page_size": raw.get(3)}
    if 5 in raw:
        d = raw[5]
        h["data_page_header"] = {"num_values": d.get(1), "encoding": d.get(2), "definition_level_encoding": d.get(3),
                                 "repetition_level_encoding": d.get(4)}
    if 7 in raw:
        d = raw[7]
        h["dictionary_page_header"] = {"num_values": d.get(1), "encoding": d.get(2)}
    if 8 in raw:
        d = raw[8]
        h["data_page_header_v2"] = {"num_values": d.get(1), "num_nulls": d.get(2), "num_rows": d.get(3), "encoding": d.get(4),
                                    "definition_levels_byte_length": d.get(5), "repetition_levels_byte_length": d.get(6),
                                    "is_compressed": d.get(7, True)}
    return h, pos


def _decompress(codec, data, size):
    if codec == 0:                         # compression.cc:29-37
        if size < len(data):
            raise ChunkError("Invalid or corrupted parquet file: Uncompression buffer size too small")
        return bytes(data)
    if codec == 1:
        return orc.snappy_decompress(data, size).tobytes()
    raise ChunkError("codec %d not restated" % codec)


def _plain(ptype, data, n, type_length):
    """value_decoder<T>{PLAIN}: (count, values)."""
    if ptype == BOOLEAN:
        v = orc.plain_boolean(data, n)
        return len(v), v
    if ptype in _FIXED:
        dt = _FIXED[ptype]
        raw = orc.plain_fixed(data, dt.itemsize, n)
        v = raw.view(dt.base).reshape((-1,) + dt.shape) if dt.shape else raw.view(dt)
        return len(v), v
    if ptype == BYTE_ARRAY:
        offs, b = orc.plain_byte_array(data, n)
        raw = b.tobytes()
        return len(offs) - 1, [raw[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    got, b = orc.plain_flba(data, type_length, n)
    raw = b.tobytes()
    return got, [raw[i * type_length:(i + 1) * type_length] for i in range(got)]


def decode_values(ptype, encoding, data, n, type_length=None, dictionary=None):
    """value_decoder<T>::reset(data, encoding) + read_batch(n) (src/encoding.cc:526-591): (count, values)."""
    if encoding == PLAIN:
        return _plain(ptype, data, n, type_length)
    if encoding in (PLAIN_DICTIONARY, RLE_DICTIONARY):
        if dictionary is None:
            raise ChunkError("No dictionary page found before a dictionary-encoded page")
        if isinstance(dictionary, np.ndarray) and not dictionary.dtype.shape and dictionary.ndim == 1:
            v = orc.dict_decode(data, dictionary, n)
            return len(v), v
        idx = orc.dict_decode(data, np.arange(len(dictionary), dtype=np.int32), n)
        return len(idx), ([dictionary[i] for i in idx] if isinstance(dictionary, list) else dictionary[idx])
    if encoding == RLE and ptype == BOOLEAN:
        v = orc.rle_boolean(data, n)
        return len(v), v
    if encoding == DELTA_BINARY_PACKED and ptype in (INT32, INT64):
        v = orc.delta_decode(data, n, _FIXED[ptype].itemsize)
        return len(v), v
    if encoding == BSS and ptype in (FLOAT, DOUBLE):
        dt = _FIXED[ptype]
        v = orc.bss_decode(data, dt.itemsize, n).view(dt)
        return len(v), v
    if encoding in (DLBA, DBA) and ptype == BYTE_ARRAY:
        offs, b = (orc.dlba_decode if encoding == DLBA else orc.dba_decode)(data, n)
        raw = b.tobytes()
        return len(offs) - 1, [raw[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    raise ChunkError("encoding %d not valid for type %d" % (encoding, ptype))


def decode_chunk(ptype, chunk, max_def, max_rep, type_length=None, codec=0):
    """Every data page of a column chunk read to its end with ONE read_batch of the page's num_values:
    list of dicts {def, rep, values, num_values, encoding}.  Errors raise ChunkError / OracleError."""
    buf = bytes(chunk) if not isinstance(chunk, (bytes, bytearray)) else chunk
    pos, pages, dictionary = 0, [], None
    while pos < len(buf):
        h, body = read_page_header(buf, pos)
        size = h["compressed_page_size"]
        if size is None or size < 0:
            raise ChunkError("Negative compressed_page_size in header")
        if body + size > len(buf):
            raise ChunkError("Unexpected end of column chunk while reading compressed page contents")
        contents = buf[body:body + size]
        pos = body + size
        if h["type"] == DICTIONARY_PAGE:                                         # column_chunk_reader.cc:113-138
            dh = h.get("dictionary_page_header")
            if dh is None:
                raise ChunkError("DictionaryPageHeader not set for DICTIONARY_PAGE header")
            data = _decompress(codec, contents, h["uncompressed_page_size"])
            got, vals = _plain(ptype, data, dh["num_values"], type_length)
            if got < dh["num_values"]:
                raise ChunkError("Unexpected end of dictionary page (expected %d values, got %d)" % (dh["num_values"], got))
            dictionary = vals
            continue
        if h["type"] == DATA_PAGE:                                               # column_chunk_reader.cc:52-78
            dh = h.get("data_page_header")
            if dh is None:
                raise ChunkError("DataPageHeader not set for DATA_PAGE header")
            nv = dh["num_values"]
            data = _decompress(codec, contents, h["uncompressed_page_size"])
            rep = np.zeros(nv, dtype=np.int32)
            dl = np.zeros(nv, dtype=np.int32)
            if max_rep > 0:
                rep, used = orc.levels_v1(data, dh["repetition_level_encoding"], max_rep, nv, nv)
                data = data[used:]
            if max_def > 0:
                dl, used = orc.levels_v1(data, dh["definition_level_encoding"], max_def, nv, nv)
                data = data[used:]
            enc = dh["encoding"]
        elif h["type"] == DATA_PAGE_V2:                                          # column_chunk_reader.cc:80-111
            dh = h.get("data_page_header_v2")
            if dh is None:
                raise ChunkError("DataPageHeaderV2 not set for DATA_PAGE_V2 header")
            nv = dh["num_values"]
            rl, dlen = dh["repetition_levels_byte_length"], dh["definition_levels_byte_length"]
            rep = np.zeros(nv, dtype=np.int32)
            dl = np.zeros(nv, dtype=np.int32)
            if max_rep > 0:
                rep = orc.levels_v2(contents[:rl], max_rep, nv, nv)
            if max_def > 0:
                dl = orc.levels_v2(contents[rl:rl + dlen], max_def, nv, nv)
            data = contents[rl + dlen:]
            if dh["is_compressed"]:
                data = _decompress(codec, data, h["uncompressed_page_size"] - rl - dlen)
            enc = dh["encoding"]
        else:
            continue                                                             # column_chunk_reader.cc:158-160: other pages are skipped
        rep, dl = np.asarray(rep, dtype=np.int32), np.asarray(dl, dtype=np.int32)
        # column_chunk_reader.hh:115-144
        if len(dl) != len(rep):
            raise ChunkError("Number of definition levels %d does not equal the number of repetition levels %d in batch"
                             % (len(dl), len(rep)))
        for name, lv, mx in (("Definition", dl, max_def), ("Repetition", rep, max_rep)):
            bad = np.flatnonzero((lv < 0) | (lv > mx))
            if bad.size:
                raise ChunkError("%s level (%d) out of range (0 to %d)" % (name, lv[bad[0]], mx))
        want = int(np.count_nonzero(dl == max_def))
        got, vals = decode_values(ptype, enc, data, want, type_length, dictionary)
        if got != want:
            raise ChunkError("Number of values in batch %d is less than indicated by def levels %d" % (got, want))
        pages.append({"def": dl, "rep": rep, "values": vals, "num_values": nv, "encoding": enc})
    return pages
