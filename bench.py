#!/usr/bin/env python3
"""Benchmark of the page-value decoding hot path on MI355X.

Headline workload (BASELINE.json configs[1]): DELTA_BINARY_PACKED int64, 256Mi
values in 2048 pages of 131072 values (1 MiB decoded each), reference encoder
shape (block 256 / 8 miniblocks), deltas uniform in [-100, 3996).  One "step"
decodes every page of the rank's shard once, device-resident (encoded pages and
page descriptors already in HBM).  With --gpus N every rank decodes its own
shard of the same size (pages are independent: weak scaling, no collective on
the data path); the reported value is the whole-job rate.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pages P] [--page-values V]
                    [--encodings delta,bss,...] [--no-cpu-baseline] [--no-verify]

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (6.3 TB/s is what a plain copy reaches)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=2048)
    ap.add_argument("--page-values", type=int, default=131072)
    ap.add_argument("--seed", type=int, default=20261111)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--extra", default="", help="comma list of extra encodings to time: bss32,bss64,rle,dict,dlba,dba")
    return ap.parse_args()


def gen_delta_pages(n_pages, page_values, seed, block=256, nmb=8):
    """Synthetic INT64 column: per page a random walk with steps uniform in [-100, 3996)."""
    from parquet4seastar_amd import pagegen
    rng = np.random.default_rng(seed)
    pages = []
    values = np.empty(n_pages * page_values, dtype=np.int64)
    for p in range(n_pages):
        steps = rng.integers(-100, 3996, size=page_values, dtype=np.int64)
        steps[0] = rng.integers(-2**40, 2**40)
        v = np.cumsum(steps)
        values[p * page_values:(p + 1) * page_values] = v
        pages.append(pagegen.delta_encode(v, block, nmb, as_array=True))
    return pages, values


def pack(pages, align=16):
    offs, pos = [], 0
    for pg in pages:
        pos = (pos + align - 1) // align * align
        offs.append(pos)
        pos += len(pg)
    buf = np.zeros(pos + 64, dtype=np.uint8)
    for o, pg in zip(offs, pages):
        buf[o:o + len(pg)] = pg
    return buf, offs, [len(pg) for pg in pages]


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from parquet4seastar_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decode path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- synthetic input, distinct per rank ----
    t0 = time.time()
    pages, values = gen_delta_pages(args.pages, args.page_values, args.seed + 1000 * rank)
    buf, offs, sizes = pack(pages)
    enc_bytes = int(sum(sizes))
    n_values = args.pages * args.page_values
    dec_bytes = n_values * 8
    descs = capi.make_descs(offs, sizes, [p * args.page_values for p in range(args.pages)], args.page_values)
    gen_s = time.time() - t0

    ctx = capi.Context(local_rank)
    # a dedicated (non-null) torch stream carries the kernels, so that torch.cuda.Event
    # records bracket exactly what the library launches
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    d_in = torch.from_numpy(buf).to(dev)
    d_pages = torch.from_numpy(descs.view(np.uint8)).to(dev)
    d_status = torch.zeros(48 * args.pages, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n_values, dtype=torch.int64, device=dev)
    L = capi.lib()

    def step():
        ctx._check(L.p4s_decode_delta_binary_packed(ctx.handle, d_in.data_ptr(), buf.size, d_pages.data_ptr(),
                                                    args.pages, d_out.data_ptr(), d_status.data_ptr(), 8))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    # per-launch kernel time from HIP events on the launch stream, wall time around the whole loop
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for a, b in evs:
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t_start
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # ---- parity of what was just timed ----
    verified = None
    if not args.no_verify:
        status = np.frombuffer(d_status.cpu().numpy().tobytes(), dtype=capi.PAGE_STATUS_DTYPE)
        capi.raise_on_page_errors(status)
        assert int(status["values_decoded"].sum()) == n_values
        want = torch.from_numpy(values).to(dev)
        verified = bool(torch.equal(d_out, want))
        del want
        if not verified:
            raise SystemExit("GPU decode differs from the generated column")

    total_bytes_all = (enc_bytes + dec_bytes) * world
    ms_per_step = wall / args.steps * 1e3
    value_gbs = total_bytes_all / (wall / args.steps) / 1e9
    avg_kernel_ms = float(np.mean(kern_ms))
    achieved = (enc_bytes + dec_bytes) / (avg_kernel_ms * 1e-3) / 1e9

    result = {
        "metric": "device-resident decoded GB/s (DELTA_BINARY_PACKED int64; encoded+decoded bytes per second)",
        "value": round(value_gbs, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": "DELTA_BINARY_PACKED int64, %d pages x %d values per GPU (block 256 / 8 miniblocks, "
                        "deltas uniform [-100,3996))" % (args.pages, args.page_values),
            "values_per_gpu": n_values,
            "encoded_bytes_per_gpu": enc_bytes,
            "decoded_bytes_per_gpu": dec_bytes,
            "sharding": "pages split across ranks, no collective",
        },
        "values_per_sec": round(n_values * world / (wall / args.steps), 1),
        "verified_bit_exact": verified,
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": "k_delta_binary_packed<uint64_t>",
            "avg_kernel_ms": round(avg_kernel_ms, 4),
            "algorithmic_bytes_per_launch": enc_bytes + dec_bytes,
        },
        "gen_seconds": round(gen_s, 2),
    }

    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        # bounded sample: the rank-0 shard decoded once, single thread, by the C restatement of the
        # reference's per-value loop (src/encoding.cc:207-239)
        n_sample = min(args.pages, 512)
        out = np.empty(args.page_values, dtype=np.int64)
        lib = orc.lib()
        import ctypes as C
        err = orc.OrcError()
        left = C.c_int64(0)
        t0 = time.perf_counter()
        sample_bytes = 0
        for p in range(n_sample):
            pg = pages[p]
            got = lib.orc_delta_decode(pg.ctypes.data, pg.size, out.ctypes.data, 8, args.page_values,
                                       C.byref(left), C.byref(err))
            assert got == args.page_values
            sample_bytes += pg.size + got * 8
        dt = time.perf_counter() - t0
        assert np.array_equal(out, values[(n_sample - 1) * args.page_values: n_sample * args.page_values])
        result["cpu_baseline"] = {
            "value": round(sample_bytes / dt / 1e9, 3),
            "unit": "GB/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d of the %d pages (%d values), one thread, C restatement of the reference's "
                      "per-value DELTA loop, gcc -O2" % (n_sample, args.pages, n_sample * args.page_values),
            "values_per_sec": round(n_sample * args.page_values / dt, 1),
        }

    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
