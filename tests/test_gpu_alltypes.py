"""column_chunk_reader<T> on the GPU against the reference's own data: the column chunks of the reference's
tests/test_data/alltypes/*.parquet (tests/golden/alltypes_chunks.json) must give the definition / repetition
levels and, rendered by the reference's formatter, the value tokens of the golden CQL text in
tests/cql_reader_alltypes_test.cc:32-72; and exactly what the CPU oracle's page walk gives."""
import numpy as np
import pytest

from oracle import chunk as ck
from parquet4seastar_amd import hostapi
from tests import alltypes_util as au

pytestmark = pytest.mark.gpu

COLUMNS = au.load()


def _values_list(v):
    return list(v) if not isinstance(v, np.ndarray) else [x for x in v]


@pytest.mark.parametrize("batch,from_device", [(2, False), (1024, False), (1024, True)])
def test_column_chunk_reader_alltypes(batch, from_device, monkeypatch):
    """batch 2 is the reference example's batch size (examples/example.cc:383-387), 1024 record_reader's.
    from_device: the chunk is put into device memory first and decoded from there (page headers walked on the device)."""
    monkeypatch.setenv("P4S_DEVICE_CHUNK", "1" if from_device else "0")
    for col in COLUMNS:
        with hostapi.Chunk(col["physical_type"], col["chunk"], col["max_def"], col["max_rep"],
                           type_length=col["type_length"]) as c:
            d_all, r_all, v_all = [], [], []
            while True:
                d, r, v = c.read_batch(batch)
                if len(d) == 0:
                    break
                d_all += d.tolist(); r_all += r.tolist(); v_all += _values_list(v)
        where = "%s:%s" % (col["file"], col["leaf"])
        assert d_all == col["def"] and r_all == col["rep"], where
        assert [au.render(col["render"], x) for x in v_all] == col["tokens"], where
        pages = ck.decode_chunk(col["physical_type"], col["chunk"], col["max_def"], col["max_rep"], col["type_length"])
        want = [x for p in pages for x in _values_list(p["values"])]
        assert len(want) == len(v_all) and all(np.array_equal(a, b) for a, b in zip(v_all, want)), where
