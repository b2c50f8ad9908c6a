"""The oracle against the reference's own data: the column chunks of tests/test_data/alltypes/*.parquet (cut into
tests/golden/alltypes_chunks.json by tests/golden/make_alltypes_chunks.py) walked by the CPU restatement of
column_chunk_reader (oracle/chunk.py) must render to the golden CQL tokens of tests/cql_reader_alltypes_test.cc:32-72.
Pins level_decoder (v1 RLE), every PLAIN decoder (BOOLEAN, INT32, INT64, INT96, FLOAT, DOUBLE, BYTE_ARRAY,
FIXED_LEN_BYTE_ARRAY) and the page walk on bytes the reference's authors wrote."""
import numpy as np
import pytest

from oracle import chunk as ck
from tests import alltypes_util as au

COLUMNS = au.load()


def test_fixture_shape():
    assert len(COLUMNS) == 73
    assert {c["physical_type"] for c in COLUMNS} == set(range(8))
    assert any(c["max_rep"] for c in COLUMNS) and any(c["max_def"] == 2 for c in COLUMNS)


@pytest.mark.parametrize("col", COLUMNS, ids=lambda c: "%s:%s" % (c["file"].split(".")[0], c["leaf"]))
def test_oracle_chunk_matches_golden_cql(col):
    pages = ck.decode_chunk(col["physical_type"], col["chunk"], col["max_def"], col["max_rep"], col["type_length"])
    dl = np.concatenate([p["def"] for p in pages])
    rl = np.concatenate([p["rep"] for p in pages])
    vals = [v for p in pages for v in p["values"]]
    assert dl.tolist() == col["def"] and rl.tolist() == col["rep"]
    assert [au.render(col["render"], v) for v in vals] == col["tokens"]
