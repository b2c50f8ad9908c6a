#!/usr/bin/env python3
"""Cuts the column chunks out of the reference's six vendored Parquet files
(/root/reference/tests/test_data/alltypes/*.uncompressed.plain.parquet) and pairs each with the values the
reference's own test expects for it (the golden CQL text of tests/cql_reader_alltypes_test.cc:32-72, read as
text), into tests/golden/alltypes_chunks.json.

What is taken from where:
  * chunk bytes: the file, [first page offset, + total_compressed_size).  pyarrow reads the FOOTER only (chunk
    offsets, physical type, type_length, max definition / repetition level, logical / converted type); no page is
    decoded by pyarrow, so the expected values do not come from a second implementation;
  * expected values: the VALUES(...) tuples of the golden text, split into one token per column and row.  The
    test renders what column_chunk_reader<T> decodes with the reference's formatter (src/cql_reader.cc:150-222)
    and compares tokens, so the pin is the reference's own file + the reference's own expected output;
  * `render`: which formatter the reference picks for the column, following read_logical_type
    (src/logical_type.cc:27-155): logicalType is honoured for TIME / TIMESTAMP / UUID only, then converted_type
    (DATE falls through to plain INT32, logical_type.cc:116-117), then the physical type;
  * nested columns (collections file): the leaf tokens in record order plus the definition / repetition levels
    they imply.  Every list / map of the golden text has two present elements per row, so rep = 0,1 per row and
    def = max_def throughout; `optional_uint32` is null in row 1 (def 1,0,1) and `optional_undefined_null` is null
    in every row (def 0,0,0, no values).

Run in the build container (needs /root/reference and pyarrow); the JSON it writes is committed.
"""
import base64
import json
import os
import re
import sys

REF = "/root/reference"
DATA = os.path.join(REF, "tests", "test_data", "alltypes")
TEST_CC = os.path.join(REF, "tests", "cql_reader_alltypes_test.cc")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "alltypes_chunks.json")

# tests/cql_reader_alltypes_test.cc:76-83
CASES = [("basic", "basic_cql"), ("collections", "collections_cql"), ("decimal", "decimal_cql"),
         ("other", "byte_array_cql"), ("time", "date_time_cql"), ("timestamp", "timestamp_interval_cql")]

PHYS = {"BOOLEAN": 0, "INT32": 1, "INT64": 2, "INT96": 3, "FLOAT": 4, "DOUBLE": 5, "BYTE_ARRAY": 6,
        "FIXED_LEN_BYTE_ARRAY": 7}


def split_top(s, sep=", ", angle=False):
    """Splits on `sep` outside quotes, brackets and braces (and, for type lists, angle brackets)."""
    out, depth, quote, cur, i = [], 0, False, "", 0
    while i < len(s):
        c = s[i]
        if quote:
            cur += c
            if c == "'":
                quote = False
        elif c == "'":
            quote = True
            cur += c
        elif c in "[{" or (angle and c == "<"):
            depth += 1
            cur += c
        elif c in "]}" or (angle and c == ">"):
            depth -= 1
            cur += c
        elif depth == 0 and s.startswith(sep, i):
            out.append(cur)
            cur = ""
            i += len(sep)
            continue
        else:
            cur += c
        i += 1
    out.append(cur)
    return out


def golden_rows(text):
    """(column names, cql types, rows of tokens) of one golden CQL block."""
    create = re.search(r'CREATE TABLE "parquet"\((.*)\);', text).group(1)
    cols = split_top(create, angle=True)
    names, types = [], []
    for c in cols:
        m = re.match(r'"((?:[^"]|"")*)" (.*?)( PRIMARY KEY)?$', c)
        names.append(m.group(1))
        types.append(m.group(2))
    rows = []
    for m in re.finditer(r"VALUES\((.*)\);", text):
        rows.append(split_top(m.group(1)))
    assert names[0] == "row_number" and [r[0] for r in rows] == ["0", "1", "2"]
    return names[1:], types[1:], [r[1:] for r in rows]


def render_kind(col_schema):
    """The reference's logical type for a leaf, as read_logical_type decides it (src/logical_type.cc:27-155)."""
    lt = str(col_schema.logical_type)
    ct = str(col_schema.converted_type)
    pt = col_schema.physical_type
    if lt.startswith("Time("):
        unit = re.search(r"timeUnit=(\w+)", lt).group(1)
        return {"milliseconds": "TIME_INT32", "microseconds": "TIME_INT64:MICROS", "nanoseconds": "TIME_INT64:NANOS"}[unit]
    if lt.startswith("Timestamp("):
        return "TIMESTAMP"
    if lt == "UUID":
        return "UUID"
    table = {"UTF8": "STRING", "ENUM": "ENUM", "INT_8": "INT8", "INT_16": "INT16", "INT_32": "INT32", "INT_64": "INT64",
             "UINT_8": "UINT8", "UINT_16": "UINT16", "UINT_32": "UINT32", "UINT_64": "UINT64", "JSON": "JSON", "BSON": "BSON",
             "INTERVAL": "INTERVAL", "TIME_MILLIS": "TIME_INT32", "TIME_MICROS": "TIME_INT64:MICROS",
             "TIMESTAMP_MILLIS": "TIMESTAMP", "TIMESTAMP_MICROS": "TIMESTAMP"}
    if ct in table:
        return table[ct]
    if ct == "DECIMAL":
        scale = int(re.search(r"scale=(\d+)", lt).group(1))
        return "DECIMAL_%s:%d" % ({"INT32": "INT32", "INT64": "INT64", "BYTE_ARRAY": "BYTES",
                                   "FIXED_LEN_BYTE_ARRAY": "BYTES"}[pt], scale)
    return pt      # DATE and every unannotated column: the physical type's own formatter


def leaf_tokens(tok):
    """Leaf tokens of a list `[a, b]` or a map `{k: v, k: v}` -> (list of element tokens) or (keys, values)."""
    if tok.startswith("["):
        return split_top(tok[1:-1])
    assert tok.startswith("{")
    pairs = [split_top(p, ": ") for p in split_top(tok[1:-1])]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def main():
    import pyarrow.parquet as pq
    src = open(TEST_CC).read()
    blocks = dict(re.findall(r'const char \* (\w+) = R"###\((.*?)\)###";', src, flags=re.S))
    out = {"_source": "tests/test_data/alltypes/*.uncompressed.plain.parquet + tests/cql_reader_alltypes_test.cc:32-72 of the reference",
           "files": []}
    for stem, var in CASES:
        path = os.path.join(DATA, stem + ".uncompressed.plain.parquet")
        raw = open(path, "rb").read()
        pf = pq.ParquetFile(path)
        md = pf.metadata
        assert md.num_row_groups == 1 and md.num_rows == 3
        names, types, rows = golden_rows(blocks[var])
        rg = md.row_group(0)
        # leaves of the file in schema order -> the golden column (top-level field) each belongs to
        columns, leaf = [], 0
        for ci, (name, cql_type) in enumerate(zip(names, types)):
            toks = [r[ci] for r in rows]
            n_leaves = 2 if cql_type.startswith("frozen<map") else 1
            for k in range(n_leaves):
                col, sc = rg.column(leaf), pf.schema.column(leaf)
                assert col.path_in_schema.split(".")[0] in (name, "map_logical") or stem != "collections"
                first = col.dictionary_page_offset if col.dictionary_page_offset else col.data_page_offset
                chunk = raw[first:first + col.total_compressed_size]
                assert col.compression == "UNCOMPRESSED"
                entry = {"name": name, "leaf": col.path_in_schema, "cql_type": cql_type,
                         "physical_type": PHYS[col.physical_type],
                         "type_length": sc.length if col.physical_type == "FIXED_LEN_BYTE_ARRAY" else None,
                         "max_def": sc.max_definition_level, "max_rep": sc.max_repetition_level,
                         "render": render_kind(sc), "num_values": col.num_values,
                         "chunk_b64": base64.b64encode(chunk).decode()}
                if cql_type.startswith("frozen<list"):
                    per_row = [leaf_tokens(t) for t in toks]
                    entry["tokens"] = [x for r in per_row for x in r]
                    entry["rep"] = [j and 1 for r in per_row for j in range(len(r))]
                    entry["def"] = [sc.max_definition_level] * len(entry["tokens"])
                elif cql_type.startswith("frozen<map"):
                    per_row = [leaf_tokens(t)[k] for t in toks]
                    entry["tokens"] = [x for r in per_row for x in r]
                    entry["rep"] = [j and 1 for r in per_row for j in range(len(r))]
                    entry["def"] = [sc.max_definition_level] * len(entry["tokens"])
                else:
                    entry["tokens"] = [t for t in toks if t != "null"]
                    entry["rep"] = [0] * 3
                    entry["def"] = [0 if t == "null" else sc.max_definition_level for t in toks]
                assert len(entry["def"]) == col.num_values, (stem, name, entry["def"], col.num_values)
                columns.append(entry)
                leaf += 1
        assert leaf == rg.num_columns, (stem, leaf, rg.num_columns)
        out["files"].append({"file": stem + ".uncompressed.plain.parquet", "golden": var, "columns": columns})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT, sum(len(x["columns"]) for x in out["files"]), "column chunks")


if __name__ == "__main__":
    sys.exit(main())
