"""Two processes, one per "GPU" (both on GPU 0 here), each decoding its sharding.shard_plan shard of one page list
through the C ABI, with gloo carrying only a barrier and a byte count: the concatenation of what the ranks decoded
must be the whole column, bit for bit.  The ranks are started by tests/conftest.py at session start, before this
process touches the GPU (a process that has initialised the GPU must not start another program)."""
import os

import numpy as np
import pytest

from tests import mp_shard_worker as w

pytestmark = pytest.mark.gpu


def test_two_ranks_decode_their_shards_of_one_page_list(request):
    job = getattr(request.config, "_p4s_mp_job", None)
    if job is None:
        pytest.skip("the ranks are only started for GPU sessions (see tests/conftest.py)")
    for p in job["procs"]:
        assert p.wait(timeout=300) == 0, open(job["log"]).read()[-4000:]
    parts = [np.load(os.path.join(job["dir"], "rank%d.npz" % r)) for r in range(job["world"])]
    (dpages, dvals), (dictionary, ipages, idx) = w.workloads()
    want_delta = np.concatenate(dvals)
    want_dict = np.concatenate([dictionary[x] for x in idx])
    for key, want in (("delta", want_delta), ("dict", want_dict)):
        firsts = [int(p[key + "_first"]) for p in parts]
        assert firsts[0] == 0 and all(firsts[r] + len(parts[r][key]) == (firsts[r + 1] if r + 1 < len(parts) else len(want))
                                      for r in range(len(parts)))
        assert min(len(p[key]) for p in parts) > 0                     # every rank had work
        got = np.concatenate([p[key] for p in parts])
        assert got.dtype == want.dtype and np.array_equal(got, want)
    for p in parts:
        assert float(p["units"]) == float(p["total_bytes"])            # the ranks' bytes add up to the list's
