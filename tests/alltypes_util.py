"""The reference's value formatter (cql_consumer::append_value, src/cql_reader.cc:150-222) restated for the
alltypes fixtures: decoded physical values -> the tokens of the golden CQL text that
tests/cql_reader_alltypes_test.cc:32-72 expects."""
import base64
import json
import os
import struct

import numpy as np

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alltypes_chunks.json")


def load():
    with open(FIXTURE) as f:
        d = json.load(f)
    cols = []
    for fl in d["files"]:
        for c in fl["columns"]:
            c = dict(c)
            c["file"] = fl["file"]
            c["chunk"] = base64.b64decode(c["chunk_b64"])
            cols.append(c)
    return cols


def _time(v, units, digits):                       # cql_reader.cc:76-91
    v = int(v) & 0xFFFFFFFFFFFFFFFF
    frac, v = v % units, v // units
    sec, v = v % 60, v // 60
    mins, hours = v % 60, v // 60
    return "'%02d:%02d:%02d.%0*d'" % (hours, mins, sec, digits, frac)


def _signed_bytes(b):                              # cql_reader.cc:171-186
    x = int.from_bytes(b, "big")
    if len(b) and b[0] & 0x80:
        x -= 1 << (8 * len(b))
    return x


def _quoted(b):
    return "'" + b.decode("latin-1").replace("'", "''") + "'"


def _blob(b):
    return "0x" + b.hex().upper()


def render(kind, v):
    """One decoded value -> its CQL token."""
    k, _, arg = kind.partition(":")
    if k == "BOOLEAN":
        return "true" if v else "false"
    if k == "INT8":
        return str(int(np.array(v).astype(np.int32).astype(np.int8)))
    if k == "INT16":
        return str(int(np.array(v).astype(np.int32).astype(np.int16)))
    if k in ("INT32", "INT64", "TIMESTAMP"):
        return str(int(v))
    if k == "UINT8":
        return str(int(v) & 0xFF)
    if k == "UINT16":
        return str(int(v) & 0xFFFF)
    if k == "UINT32":
        return str(int(v) & 0xFFFFFFFF)
    if k == "UINT64":
        return str(int(v) & 0xFFFFFFFFFFFFFFFF)
    if k in ("DECIMAL_INT32", "DECIMAL_INT64"):
        return "%de-%s" % (int(v), arg)
    if k == "DECIMAL_BYTES":
        return "%de-%s" % (_signed_bytes(v), arg)
    if k == "TIME_INT32":
        return _time(np.int64(v), 1000, 3)          # static_cast<uint64_t>(int32): sign extended
    if k == "TIME_INT64":
        return _time(v, 1000000, 6) if arg == "MICROS" else _time(v, 1000000000, 9)
    if k == "INTERVAL":
        a, b, c = struct.unpack("<III", v)
        return "%dmo%dd%dms" % (a, b, c)
    if k == "INT96":                                  # cql_reader.cc:210-217
        x = (int(v[0]) << 64) + ((int(v[1]) & 0xFFFFFFFF) << 32) + (int(v[2]) & 0xFFFFFFFF)
        return str(x)
    if k in ("FLOAT", "DOUBLE"):
        return "%e" % float(v)                        # std::scientific, default precision 6
    if k in ("STRING", "ENUM", "JSON"):
        return _quoted(v)
    if k == "UUID":
        h = v.hex().upper()
        return "-".join((h[0:8], h[8:12], h[12:16], h[16:20], h[20:32]))
    if k in ("BSON", "BYTE_ARRAY", "FIXED_LEN_BYTE_ARRAY"):
        return _blob(v)
    raise ValueError(kind)
