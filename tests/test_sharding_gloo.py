"""Multi-process bookkeeping of the page-sharded decode, world_size 2 over gloo on the CPU:
shards partition the page list, every rank derives the same global output offsets without
communication, and the whole-job rate is total units over the slowest rank's time."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parquet4seastar_amd import sharding


def test_shard_bounds_partition_and_balance():
    rng = np.random.default_rng(3)
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 8, 1000):
            w = rng.integers(1, 1000, size=n)
            b = sharding.shard_bounds(w, world)
            assert len(b) == world + 1 and b[0] == 0 and b[-1] == n
            assert all(b[i] <= b[i + 1] for i in range(world))
            if n >= 100 * world:
                loads = [w[b[r]:b[r + 1]].sum() for r in range(world)]
                assert max(loads) - min(loads) <= 2 * w.max(), (world, n, loads)
    assert sharding.shard_bounds([5, 5, 5, 5], 2) == [0, 2, 4]


def _worker(rank, world, port, n_pages, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(99)                      # every rank sees the same page headers
    num_values = rng.integers(1, 5000, size=n_pages)
    enc = rng.integers(100, 9000, size=n_pages)
    plan = sharding.shard_plan(enc, num_values, 8, world, rank)       # the function bench.py shards its page lists with
    mine = list(range(plan["lo"], plan["hi"]))
    # stand-in for the decode of the shard: the rank "writes" the value index range of its pages
    covered = torch.zeros(int(num_values.sum()), dtype=torch.int32)
    for p, o in zip(mine, plan["out_offsets"]):
        covered[plan["first_value"] + o: plan["first_value"] + o + num_values[p]] += 1
    assert plan["values"] == int(num_values[mine].sum()) and plan["total_values"] == int(num_values.sum())
    dist.all_reduce(covered, op=dist.ReduceOp.SUM)         # test only: the product never exchanges decoded data
    units, seconds, rate = sharding.whole_job_rate(float(sum(enc[p] + 8 * num_values[p] for p in mine)),
                                                   0.5 if rank == 0 else 2.0, dist)
    out_q.put((rank, bool((covered == 1).all()), units, seconds, rate, len(mine)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_cover_every_value_once():
    world, n_pages = 2, 257
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pages, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(99)
    num_values = rng.integers(1, 5000, size=n_pages)
    enc = rng.integers(100, 9000, size=n_pages)
    total = float((enc + 8 * num_values).sum())
    assert sum(r[5] for r in results) == n_pages
    for rank, ok, units, seconds, rate, _ in results:
        assert ok, "a value was decoded twice or not at all"
        assert units == total and seconds == 2.0 and abs(rate - total / 2.0) < 1e-6
