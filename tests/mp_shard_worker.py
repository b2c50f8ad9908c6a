"""One rank of the two-process sharded decode (tests/test_gpu_multiprocess.py): started as a fresh process before
its parent has touched the GPU, it joins a gloo group, takes its sharding.shard_plan shard of two page lists that
every rank generates from the same seed (DELTA_BINARY_PACKED int64 pages; dictionary-encoded INT32 data pages over
one dictionary, replicated per rank), decodes the shard through the C ABI on GPU 0 and writes what it decoded, and
where that belongs in the whole column, to a file.
    python tests/mp_shard_worker.py RANK WORLD PORT OUTDIR"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEED = 2024
DELTA_PAGES, DELTA_VALUES = 96, 20000
DICT_PAGES, DICT_VALUES, DICT_SIZE = 48, 50000, 65535


def workloads():
    """(delta pages, their values), (dictionary, index pages, their indices): the same on every rank and in the parent."""
    from parquet4seastar_amd import pagegen as g
    rng = np.random.default_rng(SEED)
    dvals = [rng.integers(-100, 3996, size=DELTA_VALUES + 7 * (i % 5)).astype(np.int64).cumsum() for i in range(DELTA_PAGES)]
    dpages = [g.delta_encode(v) for v in dvals]
    dictionary = rng.integers(-2**31, 2**31 - 1, size=DICT_SIZE).astype(np.int32)
    idx = [rng.integers(0, DICT_SIZE, size=DICT_VALUES + 11 * (i % 3)).astype(np.uint32) for i in range(DICT_PAGES)]
    ipages = [g.dict_page_values_encode(x, 16) for x in idx]
    return (dpages, dvals), (dictionary, ipages, idx)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import torch.distributed as dist
    from parquet4seastar_amd import capi, sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    (dpages, dvals), (dictionary, ipages, idx) = workloads()
    L = capi.lib()
    out = {}
    with capi.Context(0) as ctx:
        # ---- DELTA_BINARY_PACKED int64
        nv = [len(v) for v in dvals]
        plan = sharding.shard_plan([len(p) for p in dpages], nv, 8, world, rank)
        mine = dpages[plan["lo"]:plan["hi"]]
        if mine:
            buf, offs, sizes = capi.pack_pages(mine, align=16)
            descs = capi.make_descs(offs, sizes, plan["out_offsets"], nv[plan["lo"]:plan["hi"]])
            d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
            d_out, d_status = ctx.alloc(8 * max(plan["values"], 1)), ctx.alloc(48 * len(mine))
            ctx._check(L.p4s_decode_delta_binary_packed(ctx.handle, d_in.ptr, buf.size, d_pages.ptr, len(mine), d_out.ptr,
                                                        d_status.ptr, 8))
            ctx.sync()
            capi.raise_on_page_errors(d_status.download(capi.PAGE_STATUS_DTYPE, len(mine)))
            out["delta"] = d_out.download(np.int64, plan["values"])
        else:
            out["delta"] = np.zeros(0, dtype=np.int64)
        out["delta_first"] = np.int64(plan["first_value"])
        # ---- dictionary pages: the dictionary is replicated on every rank
        nv = [len(x) for x in idx]
        plan2 = sharding.shard_plan([len(p) for p in ipages], nv, 4, world, rank)
        mine = ipages[plan2["lo"]:plan2["hi"]]
        if mine:
            buf, offs, sizes = capi.pack_pages(mine, align=16)
            descs = capi.make_descs(offs, sizes, plan2["out_offsets"], nv[plan2["lo"]:plan2["hi"]])
            d_in, d_pages = ctx.to_device(buf), ctx.to_device(descs.view(np.uint8))
            d_dict = ctx.to_device(np.concatenate([dictionary, np.zeros(1, dtype=np.int32)]).view(np.uint8))
            d_out, d_status = ctx.alloc(4 * max(plan2["values"], 1)), ctx.alloc(48 * len(mine))
            ctx._check(L.p4s_decode_dictionary(ctx.handle, d_in.ptr, buf.size, d_pages.ptr, len(mine), d_dict.ptr, DICT_SIZE, 4,
                                               d_out.ptr, d_status.ptr))
            ctx.sync()
            capi.raise_on_page_errors(d_status.download(capi.PAGE_STATUS_DTYPE, len(mine)))
            out["dict"] = d_out.download(np.int32, plan2["values"])
        else:
            out["dict"] = np.zeros(0, dtype=np.int32)
        out["dict_first"] = np.int64(plan2["first_value"])
    # the only things the ranks exchange: a barrier and their byte counts (the whole-job rate of bench.py)
    units, seconds, _ = sharding.whole_job_rate(float(plan["bytes"] + plan2["bytes"]), 1.0, dist)
    out["units"] = np.float64(units)
    out["total_bytes"] = np.float64(plan["total_bytes"] + plan2["total_bytes"])
    dist.barrier()
    dist.destroy_process_group()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)


if __name__ == "__main__":
    main()
