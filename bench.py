#!/usr/bin/env python3
"""Benchmark of the page-value decoding hot path on MI355X.

Headline workload (BASELINE.json configs[1]): DELTA_BINARY_PACKED int64, 256Mi
values in 2048 pages of 131072 values (1 MiB decoded each), reference encoder
shape (block 256 / 8 miniblocks), deltas uniform in [-100, 3996).  One "step"
decodes every page of the rank's shard once, device-resident (encoded pages and
page descriptors already in HBM).

With --gpus N the ONE page list is partitioned over the ranks (contiguous, byte-balanced shards,
parquet4seastar_amd.sharding.shard_plan; no collective on the data path): strong scaling, the reported value
is the whole list's bytes over the slowest rank's time.  --scaling weak gives every rank a list of its own
(the previous behaviour).

The other configurations of BASELINE.json (hybrid RLE widths 1-20, BYTE_STREAM_SPLIT
float/double, the 64K-entry dictionary column through column_chunk_decoder, the
DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY string columns) are timed in the same run
and reported under "encodings" (device time of the kernels, HIP events on the launch
stream; --encodings none skips them), sharded the same way; "sweep" repeats RLE width 12, the headline DELTA
pages and the dictionary chunk at 1, 8, 64, 256 and 2048 pages, each next to the one-core CPU figure.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pages P] [--page-values V] [--scaling strong|weak]
                    [--encodings all|none|rle,bss,dict,strings,delta,snappy,sweep] [--no-cpu-baseline] [--no-verify]

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (about 6.3 TB/s is what a plain copy reaches)
MI = 1 << 20


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
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host (PCIe inclusive) measurement")
    ap.add_argument("--only", default="", help="comma list of entry names: of the chosen encodings, run just these (counter passes)")
    ap.add_argument("--encodings", default="all", help="all, none or a comma list of rle,bss,dict,strings,delta,snappy,sweep")
    ap.add_argument("--rle-widths", default="1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20")
    ap.add_argument("--rle-pages", type=int, default=2048, help="pages of 1Mi values per RLE / dictionary workload")
    ap.add_argument("--enc-steps", type=int, default=5, help="timed steps per extra encoding")
    ap.add_argument("--scale", type=float, default=1.0, help="scales the extra encodings' sizes (tests use < 1)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="with --gpus N: one page list partitioned over the ranks (strong) or a list per rank (weak)")
    ap.add_argument("--sweep-pages", default="1,8,64,256,2048", help="page counts of the small-batch sweep")
    return ap.parse_args()


# --------------------------------------------------------------------------------------
# inputs, shards, timing
# --------------------------------------------------------------------------------------
class Shard:
    """Which ranks there are and how a page list is divided among them."""

    def __init__(self, rank, world, strong):
        self.rank, self.world, self.strong = rank, world, strong

    def plan(self, sizes, num_values, elem):
        """shard_plan of this rank over a list that every rank builds identically (strong), or the whole list of a
        rank's own (weak: every rank decodes a list of the same shape, the job's total is world times it)."""
        from parquet4seastar_amd import sharding
        if self.strong:
            return sharding.shard_plan(sizes, num_values, elem, self.world, self.rank)
        pl = sharding.shard_plan(sizes, num_values, elem, 1, 0)
        pl["total_bytes"] *= self.world
        pl["total_values"] *= self.world
        return pl


def pack(pages, align=16):
    offs, pos = [], 0
    for pg in pages:
        pos = (pos + align - 1) // align * align
        offs.append(pos)
        pos += len(pg)
    total = (pos + 63) // 64 * 64
    buf = np.zeros(total, dtype=np.uint8)
    for o, pg in zip(offs, pages):
        buf[o:o + len(pg)] = np.frombuffer(bytes(pg), dtype=np.uint8) if not isinstance(pg, np.ndarray) else pg
    return buf, offs, [len(pg) for pg in pages]


class Timer:
    """HIP-event stopwatch of the library, on the stream the kernels are launched on."""

    def __init__(self, ctx_handle):
        from parquet4seastar_amd import capi
        self.L, self.h = capi.lib(), ctx_handle

    def time(self, fn, steps, warmup=1):
        for _ in range(warmup):
            fn()
        ms = C.c_float()
        assert self.L.p4s_timer_start(self.h) == 0
        for _ in range(steps):
            fn()
        assert self.L.p4s_timer_stop(self.h, C.byref(ms)) == 0
        return ms.value / steps


def replicated(ctx, pages, lo, hi):
    """Pages lo .. hi-1 of the list pages[0], pages[1], ..., pages[d-1], pages[0], ... (replicas of a few distinct
    pages at distinct addresses) in device memory.  Returns (device buffer, in_bytes, offsets, sizes)."""
    from parquet4seastar_amd import capi
    buf, offs, sizes = pack(pages)
    block, d = buf.size, len(pages)
    r_lo, r_hi = lo // d, (hi + d - 1) // d
    n_blocks = max(r_hi - r_lo, 1)
    d_in = ctx.alloc(block * n_blocks + 64)
    for r in range(n_blocks):
        ctx._check(capi.lib().p4s_copy_h2d(ctx.handle, d_in.ptr + r * block, buf.ctypes.data, block))
    ctx.sync()
    all_offs = [(i // d - r_lo) * block + offs[i % d] for i in range(lo, hi)]
    return d_in, block * n_blocks, all_offs, [sizes[i % d] for i in range(lo, hi)]


def entry(ms, enc, dec, values, extra_bytes=0):
    alg = enc + dec + extra_bytes
    gbs = alg / (ms * 1e-3) / 1e9
    return {"ms": round(ms, 4), "GB/s": round(gbs, 1), "values_per_sec": round(values / (ms * 1e-3), 1),
            "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "encoded_bytes": enc, "decoded_bytes": dec, "values": values}


def run_fixed_family(ctx, timer, shard, fn, pages, n_total, n_per_page, elem, truths, steps, extra=(), param0=0, verify=True,
                     e2e=None):
    """Times one fixed-width-output entry point over this rank's shard of the list of n_total pages made of
    replicas of `pages`; truths[i] = values of pages[i].  Returns (ms, enc bytes, dec bytes, values) of the SHARD
    plus the plan (totals of the whole list)."""
    from parquet4seastar_amd import capi
    d = len(pages)
    sizes_all = [len(pages[i % d]) for i in range(n_total)]
    pl = shard.plan(sizes_all, [n_per_page] * n_total, elem)
    lo, hi = pl["lo"], pl["hi"]
    n_pages = hi - lo
    if n_pages == 0:
        return 0.0, 0, 0, 0, pl
    d_in, in_bytes, offs, sizes = replicated(ctx, pages, lo, hi)
    descs = capi.make_descs(offs, sizes, [int(x) for x in pl["out_offsets"]], n_per_page, param0)
    d_pages = ctx.to_device(descs.view(np.uint8))
    d_status = ctx.alloc(48 * n_pages)
    d_out = ctx.alloc(max(pl["values"], 1) * elem)
    f = getattr(capi.lib(), fn)

    def step():
        ctx._check(f(ctx.handle, d_in.ptr, in_bytes, d_pages.ptr, n_pages, d_out.ptr, d_status.ptr, *extra))

    ms = timer.time(step, steps)
    if os.environ.get("P4S_BENCH_DUMP_AUX"):       # kernel timing experiments (-DP4S_EXP_TIMING builds)
        stt = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
        a0, a1 = stt["aux0"].astype(np.uint64), stt["aux1"].astype(np.uint64)
        print(fn, "aux cycles per page (mean):", [float(x.mean()) for x in
              (a0 & 0xFFFFFFFF, a0 >> 32, a1 & 0xFFFFFFFF, a1 >> 32, stt["bytes_consumed"].astype(np.uint64))], file=sys.stderr)
    if verify:
        status = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
        capi.raise_on_page_errors(status)
        assert int(status["values_decoded"].sum()) == n_pages * n_per_page
        for i in sorted(set([0, n_pages // 2, n_pages - 1])):      # first, middle and last page of the shard
            got = d_out.download(np.uint8, n_per_page * elem, byte_offset=int(pl["out_offsets"][i]) * elem)
            if not np.array_equal(got, truths[(lo + i) % d].view(np.uint8).reshape(-1)):
                raise SystemExit("%s: GPU decode differs from the generated column" % fn)
    if e2e is not None:          # a dict that receives the PCIe-inclusive figure of the same launch
        dec_b = n_pages * n_per_page * elem
        ent, (h_out,) = pcie_inclusive(ctx, d_in, in_bytes, step, [(d_out, dec_b)], int(sum(sizes)) + dec_b)
        i = n_pages - 1
        lo_b = int(pl["out_offsets"][i]) * elem
        ent["verified_last_page"] = bool(np.array_equal(h_out[lo_b:lo_b + n_per_page * elem], truths[(lo + i) % d].view(np.uint8).reshape(-1)))
        e2e.update(ent)
    for b in (d_in, d_pages, d_status, d_out):
        b.free()
    return ms, int(sum(sizes)), n_pages * n_per_page * elem, n_pages * n_per_page, pl


def pcie_inclusive(ctx, d_in, in_bytes, step, outs, alg_bytes, reps=3):
    """The same step with the copies the reference's path implies, on one stream without overlap: the encoded pages
    start in pinned host memory, every output buffer in `outs` = [(device Buffer, bytes)] ends there.
    Returns (the entry for the bench line, [pinned host arrays of the outputs])."""
    import ctypes as C
    from parquet4seastar_amd import capi
    L = capi.lib()
    h_in = C.c_void_p()
    ctx._check(L.p4s_host_alloc(ctx.handle, max(in_bytes, 16), C.byref(h_in)))
    ctx._check(L.p4s_copy_d2h(ctx.handle, h_in.value, d_in.ptr, in_bytes))
    h_outs = []
    for _, nbytes in outs:
        h = C.c_void_p()
        ctx._check(L.p4s_host_alloc(ctx.handle, max(nbytes, 16), C.byref(h)))
        h_outs.append(h)
    ctx.sync()

    def e2e():
        ctx._check(L.p4s_copy_h2d(ctx.handle, d_in.ptr, h_in.value, in_bytes))
        step()
        for (d, nbytes), h in zip(outs, h_outs):
            ctx._check(L.p4s_copy_d2h(ctx.handle, h.value, d.ptr, nbytes))

    e2e()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        e2e()
    ctx.sync()
    sec = (time.perf_counter() - t0) / reps
    arrays = [np.ctypeslib.as_array(C.cast(h.value, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),)).copy()[:nbytes]
              for (_, nbytes), h in zip(outs, h_outs)]
    L.p4s_host_free(ctx.handle, h_in)
    for h in h_outs:
        L.p4s_host_free(ctx.handle, h)
    return {"GB/s": round(alg_bytes / sec / 1e9, 2), "ms_per_step": round(sec * 1e3, 3),
            "what": "pinned host -> device copy of the encoded pages, decode, device -> pinned host copy of the output; one "
                    "stream, no overlap, per GPU"}, arrays


class DistCtx:
    def __init__(self, world, rdev):
        self.world, self.rdev = world, rdev

    def max_ms(self, ms):
        if self.world == 1:
            return ms
        import torch
        import torch.distributed as dist
        t = torch.tensor([ms], dtype=torch.float64, device=self.rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, *vals):
        if self.world == 1:
            return list(vals)
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(vals), dtype=torch.float64, device=self.rdev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(x) for x in t.tolist()]


def cpu_time(fn, min_seconds=0.5):
    """Seconds per call of fn (one thread), repeated until min_seconds have passed."""
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            return dt / n


def cpu_info():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "cores_visible": len(os.sched_getaffinity(0))}


def wanted(args, name):
    return not args.only or name in args.only.split(",")


def traffic_of(name):
    """HBM bytes per launch measured with rocprofv3 --pmc (profiles/traffic.json: FETCH_SIZE doubled as gfx950 needs
    for 16-byte-per-lane streaming reads, plus WRITE_SIZE, separate passes), or None when no pass covers `name`."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f).get(name)
            return t.get("hbm_bytes") if isinstance(t, dict) else t
    except (OSError, ValueError):
        return None


def finish_entry(name, out, dctx, ms, enc, dec, nv, pl, workload, extra_bytes=0, offset_bytes=None):
    """One "encodings" entry: this rank's time reduced to the slowest rank's, bytes of the whole list."""
    ms_all = dctx.max_ms(ms)
    enc_all, dec_all, nv_all, extra_all = dctx.sum_int(enc, dec, nv, extra_bytes)
    e = entry(ms_all, enc_all, dec_all, nv_all, extra_all)
    e["hbm_frac"] = round(e["GB/s"] / dctx.world / HBM_PEAK_GBS, 4)      # per GPU
    if offset_bytes is not None:
        e["offset_bytes"] = extra_all
    e["workload"] = workload
    t = traffic_of(name)
    if t is not None and dctx.world == 1:
        e["traffic"] = t
    out[name] = e
    return e


# --------------------------------------------------------------------------------------
# the other encodings (device time per launch)
# --------------------------------------------------------------------------------------
def rle_cpu(orc, page, w, n):
    ref = orc.ref()
    if ref is not None:      # the reference's own RleDecoder::GetBatch (oracle/_ref)
        dt = cpu_time(lambda: orc.rle_decode_batches(page, w, [n], 4, use_ref=True))
        kind = "reference"
    else:
        dt = cpu_time(lambda: orc.rle_decode(page, w, n, 4))
        kind = "port"
    return {"GB/s": round((len(page) + 4 * n) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1),
            "cores": 1, "kind": kind, "sample": "one page of %d values" % n}


def bench_rle(ctx, timer, shard, dctx, args, rng, out, cpu):
    from oracle import oracle as orc
    from parquet4seastar_amd import pagegen
    from parquet4seastar_amd.pagegen import synth
    n = int(MI * args.scale) // 8 * 8
    distinct, total = 8, max(8, int(args.rle_pages * args.scale) // 8 * 8)
    for w in [int(x) for x in args.rle_widths.split(",") if x]:
        if not wanted(args, "rle_w%d" % w):
            continue
        vals = [synth.run_values(rng, n, w).astype(np.uint32) for _ in range(distinct)]
        pages = [pagegen.rle_encode(v, w) for v in vals]
        ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, shard, "p4s_decode_rle", pages, total, n, 4, vals, args.enc_steps,
                                                extra=(4,), param0=w, verify=not args.no_verify)
        e = finish_entry("rle_w%d" % w, out, dctx, ms, enc, dec, nv, pl,
                         "%d pages x %d int32 values, width %d, alternating repeated/literal runs of 8-107 values" % (total, n, w))
        if cpu and w in (1, 8, 16, 20):
            e["cpu"] = rle_cpu(orc, pages[0], w, n)
    # the two extremes of the run structure, at one width: literal runs only (as long as the reference's encoder
    # makes them: 63 groups = 504 values per header) and repeated runs of 1024 values
    w = 12
    for name, make in (("rle_w12_all_literal", lambda: rng.integers(0, 1 << w, size=n, dtype=np.uint64).astype(np.uint32)),
                       ("rle_w12_all_repeated", lambda: np.repeat(rng.integers(0, 1 << w, size=n // 1024 + 1, dtype=np.uint64)
                                                                  .astype(np.uint32), 1024)[:n])):
        if not wanted(args, name):
            continue
        vals = [make() for _ in range(distinct)]
        pages = [pagegen.rle_encode(v, w) for v in vals]
        ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, shard, "p4s_decode_rle", pages, total, n, 4, vals, args.enc_steps,
                                                extra=(4,), param0=w, verify=not args.no_verify)
        finish_entry(name, out, dctx, ms, enc, dec, nv, pl, "%d pages x %d int32 values, width %d, %s" % (
            total, n, w, "literal runs of 504 values" if "literal" in name else "repeated runs of 1024 values"))
    # the write path's counterpart: the same values encoded on the GPU (values in, hybrid streams out), byte for byte
    # what the reference's RleEncoder writes
    from parquet4seastar_amd import capi
    for name, w, make, what in (
            ("rle_encode_w12", 12, lambda: synth.run_values(rng, n, 12).astype(np.uint32), "alternating repeated/literal runs of 8-107 values"),
            ("rle_encode_w1_levels", 1, lambda: (rng.random(n) > 0.1).astype(np.uint32), "definition levels with 10 % nulls")):
        if not wanted(args, name):
            continue
        vals = [make() for _ in range(distinct)]
        want = [orc.rle_encode(v, w) for v in vals[:2]]
        sizes_all = [n * 4] * total
        pl = shard.plan(sizes_all, [n] * total, 4)
        lo, hi = pl["lo"], pl["hi"]
        n_pages = hi - lo
        ms = enc_b = dec_b = nv = 0
        if n_pages:
            # every page reads values of its own (replicas of the distinct columns, each at its own address)
            d_vals = ctx.alloc(n_pages * n * 4)
            for i in range(n_pages):
                a = vals[(lo + i) % distinct]
                ctx._check(capi.lib().p4s_copy_h2d(ctx.handle, d_vals.ptr + i * n * 4, a.ctypes.data, a.nbytes))
            ctx.sync()
            cap = capi.rle_encode_bound(n, w)
            slot = (cap + 15) // 16 * 16
            descs = capi.make_descs([i * n for i in range(n_pages)], [n * 4] * n_pages, [i * slot for i in range(n_pages)], n,
                                    param0=cap, param1=w)
            d_pages = ctx.to_device(descs.view(np.uint8))
            d_status, d_out = ctx.alloc(48 * n_pages), ctx.alloc(slot * n_pages)

            def step():
                ctx._check(capi.lib().p4s_encode_rle(ctx.handle, d_vals.ptr, n_pages * n, d_pages.ptr, n_pages, d_out.ptr,
                                                     slot * n_pages, d_status.ptr))
            ms = timer.time(step, args.enc_steps)
            status = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
            capi.raise_on_page_errors(status)
            if os.environ.get("P4S_PROBE_AUX"):      # -DP4S_EXP_TIMING builds: cycle counts of the kernel's phases in the status words
                a = [status[k].astype(np.uint64) for k in ("aux0", "aux1", "bytes_consumed")]
                sys.stderr.write("%s cycles/page: stage %.0f bits %.0f list %.0f entries %.0f pack %.0f flush %.0f\n" % (
                    name, (a[0] & 0xFFFFFFFF).mean(), (a[0] >> 32).mean(), (a[1] & 0xFFFFFFFF).mean(), (a[1] >> 32).mean(),
                    (a[2] & 0xFFFFFFFF).mean(), (a[2] >> 32).mean()))
            if not args.no_verify:
                for i in range(n_pages):
                    k = (lo + i) % distinct
                    if k < len(want):
                        got = d_out.download(np.uint8, len(want[k]), byte_offset=i * slot)
                        if int(status[i]["out_bytes"]) != len(want[k]) or got.tobytes() != want[k]:
                            raise SystemExit("%s: the GPU's stream differs from the reference encoder's" % name)
                        break
            enc_b, dec_b, nv = int(status["out_bytes"].sum()), n_pages * n * 4, n_pages * n
            for b in (d_vals, d_pages, d_status, d_out):
                b.free()
        e = finish_entry(name, out, dctx, ms, enc_b, dec_b, nv, pl,
                         "ENCODER: %d pages x %d uint32 values, width %d, %s -> RLE / bit-packed hybrid streams" % (total, n, w, what))
        if cpu:
            dt = cpu_time(lambda: orc.ref_rle_encode(vals[0], w) if orc.ref() is not None else orc.rle_encode(vals[0], w))
            e["cpu"] = {"GB/s": round((len(want[0]) + n * 4) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1), "cores": 1,
                        "kind": "reference" if orc.ref() is not None else "port", "sample": "one page of %d values" % n}


def bench_bss(ctx, timer, shard, dctx, args, rng, out, cpu):
    """Config 3: 512Mi float32 and 512Mi float64 values in pages of 1 MiB (encoded), raw bit patterns."""
    from oracle import oracle as orc
    from parquet4seastar_amd import pagegen
    for name, dt_, elem in (("bss_float32", np.uint32, 4), ("bss_float64", np.uint64, 8)):
        n = int(MI * args.scale) // elem // 16 * 16            # 1 MiB encoded pages
        distinct = 32
        total = max(distinct, int(512 * MI * args.scale * args.scale) // n // distinct * distinct)   # 512Mi values
        if not wanted(args, name):
            continue
        vals = [rng.integers(0, np.iinfo(dt_).max, size=n, dtype=dt_) for _ in range(distinct)]    # raw bit patterns (NaNs too)
        pages = [pagegen.bss_encode(v) for v in vals]
        e2e = {} if (elem == 4 and dctx.world == 1 and not args.no_e2e) else None
        ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, shard, "p4s_decode_byte_stream_split", pages, total, n, elem, vals,
                                                args.enc_steps, extra=(elem,), verify=not args.no_verify, e2e=e2e)
        e = finish_entry(name, out, dctx, ms, enc, dec, nv, pl,
                         "%d pages x %d values (1 MiB encoded each, %d values in all), random bit patterns" % (total, n, total * n))
        if e2e:
            e["pcie_inclusive"] = e2e
        if cpu:
            dt = cpu_time(lambda: orc.bss_decode(pages[0], elem, n))
            e["cpu"] = {"GB/s": round(2 * n * elem / dt / 1e9, 3), "values_per_sec": round(n / dt, 1), "cores": 1,
                        "kind": "port", "sample": "one page of %d values" % n}


def bench_delta_variants(ctx, timer, shard, dctx, args, rng, out, cpu):
    """Config 2's other shapes: constant stride (width 0), full-range 64-bit deltas (width 64), block 128 / 4
    miniblocks, INT32; and the copy ceiling (PLAIN int64 pages of the same decoded size through the copy kernel)."""
    from oracle import oracle as orc
    from parquet4seastar_amd import pagegen
    n = int(131072 * args.scale) // 1024 * 1024
    distinct, total = 16, max(16, int(2048 * args.scale) // 16 * 16)
    cases = (
        ("delta_int64_w0", np.int64, 256, 8, "constant stride (every width 0)",
         lambda: (np.arange(n, dtype=np.int64) * 7 + int(rng.integers(-2**40, 2**40)))),
        ("delta_int64_w64", np.int64, 256, 8, "full-range values (widths up to 64, wrap-around sums)",
         lambda: rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)),
        ("delta_int64_128x4", np.int64, 128, 4, "deltas uniform [-100,3996), block 128 / 4 miniblocks",
         lambda: np.cumsum(rng.integers(-100, 3996, size=n, dtype=np.int64))),
        ("delta_int32", np.int32, 256, 8, "INT32, deltas uniform [-100,3996)",
         lambda: np.cumsum(rng.integers(-100, 3996, size=n, dtype=np.int64)).astype(np.int32)),
    )
    for name, dt_, block, nmb, what, gen in cases:
        elem = np.dtype(dt_).itemsize
        if not wanted(args, name):
            continue
        vals = [gen() for _ in range(distinct)]
        pages = [pagegen.delta_encode(v, block, nmb, as_array=True) for v in vals]
        ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, shard, "p4s_decode_delta_binary_packed", pages, total, n, elem, vals,
                                                args.enc_steps, extra=(elem,), verify=not args.no_verify)
        e = finish_entry(name, out, dctx, ms, enc, dec, nv, pl, "%d pages x %d values, %s" % (total, n, what))
        if cpu:
            dt = cpu_time(lambda: orc.delta_decode(bytes(pages[0]), n, elem))
            e["cpu"] = {"GB/s": round((len(pages[0]) + n * elem) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1),
                        "cores": 1, "kind": "port", "sample": "one page of %d values" % n}
    if wanted(args, "delta_encode_int64"):
        # the write path's counterpart of the headline: the same columns encoded on the GPU (values in, pages out)
        from parquet4seastar_amd import capi
        vals = [np.cumsum(rng.integers(-100, 3996, size=n, dtype=np.int64)) for _ in range(distinct)]
        want = [pagegen.delta_encode(v, 256, 8, as_array=True) for v in vals]
        sizes_all = [len(want[i % distinct]) for i in range(total)]
        pl = shard.plan(sizes_all, [n] * total, 8)
        lo, hi = pl["lo"], pl["hi"]
        n_pages = hi - lo
        ms = enc_b = dec_b = nv = 0
        if n_pages:
            # every page reads values of its own (replicas of the distinct columns, each at its own address): input
            # that all pages shared would come out of the caches, not out of HBM
            have = n_pages
            d_vals = ctx.alloc(n_pages * n * 8)
            for i in range(n_pages):
                a = vals[(lo + i) % distinct]
                ctx._check(capi.lib().p4s_copy_h2d(ctx.handle, d_vals.ptr + i * n * 8, a.ctypes.data, a.nbytes))
            ctx.sync()
            cap = capi.delta_encode_bound(n, 8)
            slot = (cap + 15) // 16 * 16
            descs = capi.make_descs([(i % have) * n for i in range(n_pages)], [n * 8] * n_pages,
                                    [i * slot for i in range(n_pages)], n, param0=cap)
            d_pages = ctx.to_device(descs.view(np.uint8))
            d_status, d_out = ctx.alloc(48 * n_pages), ctx.alloc(slot * n_pages)

            def step():
                ctx._check(capi.lib().p4s_encode_delta_binary_packed(ctx.handle, d_vals.ptr, have * n, d_pages.ptr, n_pages,
                                                                     d_out.ptr, slot * n_pages, d_status.ptr, 8))
            ms = timer.time(step, args.enc_steps)
            status = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
            capi.raise_on_page_errors(status)
            if not args.no_verify:
                for i in sorted(set([0, n_pages // 2, n_pages - 1])):
                    w = want[(lo + (i % have)) % distinct]
                    got = d_out.download(np.uint8, len(w), byte_offset=i * slot)
                    if int(status[i]["out_bytes"]) != len(w) or not np.array_equal(got, np.asarray(w).view(np.uint8).reshape(-1)):
                        raise SystemExit("delta_encode_int64: GPU pages differ from the generator's")
            enc_b, dec_b, nv = int(status["out_bytes"].sum()), n_pages * n * 8, n_pages * n
            for b in (d_vals, d_pages, d_status, d_out):
                b.free()
        e = finish_entry("delta_encode_int64", out, dctx, ms, enc_b, dec_b, nv, pl,
                         "ENCODER: %d pages x %d int64 values, deltas uniform [-100,3996) -> DELTA_BINARY_PACKED pages (256 / 8)" % (total, n))
        if cpu:
            dt = cpu_time(lambda: orc.delta_encode(vals[0], 8))
            e["cpu"] = {"GB/s": round((len(want[0]) + n * 8) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1),
                        "cores": 1, "kind": "port", "sample": "one page of %d values" % n}
    if not wanted(args, "plain_int64_copy"):
        return
    # the practical ceiling the scope table asks for: a plain copy of as many bytes as the headline decodes
    vals = [rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64) for _ in range(distinct)]
    pages = [v.view(np.uint8) for v in vals]
    ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, shard, "p4s_decode_plain_fixed", pages, total, n, 8, vals,
                                            args.enc_steps, extra=(8,), verify=not args.no_verify)
    finish_entry("plain_int64_copy", out, dctx, ms, enc, dec, nv, pl,
                 "%d PLAIN int64 pages x %d values: the copy kernel, read + write" % (total, n))


def dict_chunk_decode(timer_of, args, chunk, max_def, n, n_pages, truth_vals, truth_def, name, verify):
    """column_chunk_decoder<INT32> over one chunk: (ms per launch, counts)."""
    from parquet4seastar_amd import capi, hostapi, pagegen as g
    distinct = len(truth_vals)
    with hostapi.Chunk(g.TYPE_INT32, chunk, max_def, 0) as c:
        c.plan()
        timer = timer_of(hostapi.current_ctx())
        ms = timer.time(c.launch, args.enc_steps)
        counts = c.collect()
        if counts["failed_pages"]:
            raise SystemExit("%s: %d pages failed: %s" % (name, counts["failed_pages"], c.page_error(1)))
        if verify:
            L, h = capi.lib(), hostapi.current_ctx()
            for p in sorted(set([0, n_pages // 2, n_pages - 1])):
                tv = truth_vals[p % distinct]
                got = np.empty(len(tv), dtype=np.int32)
                assert L.p4s_copy_d2h(h, got.ctypes.data, c.device_ptr("values") + 4 * p * n, got.nbytes) == 0
                gd = np.empty(n, dtype=np.int32)
                if max_def:
                    assert L.p4s_copy_d2h(h, gd.ctypes.data, c.device_ptr("def") + 4 * p * n, gd.nbytes) == 0
                assert L.p4s_ctx_sync(h) == 0
                if not np.array_equal(got, tv) or (max_def and not np.array_equal(gd, truth_def[p % distinct])):
                    raise SystemExit("%s: GPU decode differs from the generated column" % name)
    return ms, counts


def dict_chunk_e2e(args, chunk, max_def, n, n_pages, truth_vals, truth_def, alg_bytes, reps=2):
    """The dictionary chunk from host memory to host memory: page-header walk, upload, decode (column_chunk_decoder),
    values and levels copied back to pinned host memory; wall clock per chunk."""
    import ctypes as C
    from parquet4seastar_amd import capi, hostapi, pagegen as g
    L = capi.lib()
    distinct = len(truth_vals)
    h_val, h_def = C.c_void_p(), C.c_void_p()
    secs, ok = [], True
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        with hostapi.Chunk(g.TYPE_INT32, chunk, max_def, 0) as c:
            c.plan()
            c.launch()
            counts = c.collect()
            h = hostapi.current_ctx()
            if not h_val:
                assert L.p4s_host_alloc(h, 4 * n * n_pages, C.byref(h_val)) == 0
                assert L.p4s_host_alloc(h, 4 * n * n_pages, C.byref(h_def)) == 0
                continue                                                  # (the first pass also pays for the pinned buffers)
            assert L.p4s_copy_d2h(h, h_val.value, c.device_ptr("values"), 4 * n * n_pages) == 0
            if max_def:
                assert L.p4s_copy_d2h(h, h_def.value, c.device_ptr("def"), 4 * n * n_pages) == 0
            assert L.p4s_ctx_sync(h) == 0
            secs.append(time.perf_counter() - t0)
            p = n_pages - 1
            tv = truth_vals[p % distinct]
            got = np.ctypeslib.as_array(C.cast(h_val.value, C.POINTER(C.c_int32)), shape=(n * n_pages,))[p * n:p * n + len(tv)]
            ok = ok and bool(np.array_equal(got, tv)) and not counts["failed_pages"]
    h = hostapi.current_ctx()
    L.p4s_host_free(h, h_val)
    L.p4s_host_free(h, h_def)
    sec = min(secs)
    return {"GB/s": round(alg_bytes / sec / 1e9, 2), "ms_per_step": round(sec * 1e3, 3), "verified_last_page": ok,
            "what": "chunk in host memory -> header walk, upload, decode -> values (and levels) in pinned host memory; "
                    "wall clock, one stream, no overlap, per GPU"}


def bench_dict(timer_of, shard, dctx, args, rng, out, cpu):
    """Config 4: dictionary-encoded INT32 column chunk through column_chunk_decoder (C++ host layer).  Every rank
    gets the dictionary page and its shard of the data pages (the dictionary is replicated, as a reader would)."""
    from oracle import oracle as orc
    from parquet4seastar_amd import pagegen as g
    from parquet4seastar_amd.pagegen import synth
    n = int(MI * args.scale) // 8 * 8
    distinct, total = 8, max(8, int(args.rle_pages * args.scale) // 8 * 8)
    for name, dsize, dist, max_def in (("dict_65535_uniform", 65535, "uniform", 0), ("dict_65536_zipf_nulls", 65536, "zipf", 1)):
        if not wanted(args, name):
            continue
        w = synth.dict_index_width(dsize)
        dict_page, pages, dictionary, truth_vals, truth_def = synth.dict_chunk_pages(rng, dsize, dist, max_def, n, distinct)
        sizes_all = [len(pages[i % distinct]) for i in range(total)]
        pl = shard.plan(sizes_all, [n] * total, 4)
        lo, hi = pl["lo"], pl["hi"]
        body = [np.frombuffer(pages[i % distinct], dtype=np.uint8) for i in range(lo, hi)]
        chunk = np.concatenate([np.frombuffer(dict_page, dtype=np.uint8)] + body)
        # the generated columns in shard order
        tv = [truth_vals[(lo + i) % distinct] for i in range(min(hi - lo, distinct))]
        td = [truth_def[(lo + i) % distinct] for i in range(min(hi - lo, distinct))]
        ms, counts = dict_chunk_decode(timer_of, args, chunk, max_def, n, hi - lo, tv, td, name, not args.no_verify)
        n_levels, n_values = counts["levels"], counts["values"]
        assert n_levels == n * (hi - lo)
        dec = 4 * n_values + (4 * n_levels if max_def else 0)
        e = finish_entry(name, out, dctx, ms, int(chunk.size), dec, n_values, pl,
                         "column chunk: %d-entry INT32 dictionary page + %d v1 data pages x %d values, %d-bit RLE indices (%s)%s, "
                         "decoded by column_chunk_decoder in one launch per section" % (
                             dsize, total, n, w, dist, ", max_def 1 with 10% nulls" if max_def else ""))
        if dctx.world == 1 and not args.no_e2e and max_def == 0:
            e["pcie_inclusive"] = dict_chunk_e2e(args, chunk, max_def, n, hi - lo, tv, td, int(chunk.size) + dec)
        if cpu:
            sec = g.dict_page_values_encode(np.minimum(rng.zipf(1.1, size=n) - 1, dsize - 1).astype(np.uint32)
                                            if dist == "zipf" else rng.integers(0, dsize, size=n).astype(np.uint32), w)
            dt = cpu_time(lambda: orc.dict_decode(sec, dictionary, n))
            e["cpu"] = {"GB/s": round((len(sec) + 4 * n) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1), "cores": 1,
                        "kind": "port", "sample": "the value section of one page (%d values), no levels" % n}


def bench_sweep(ctx, timer, timer_of, args, rng, out, cpu):
    """Few pages in flight: RLE width 12, the headline DELTA pages and the dictionary chunk at 1 ... 2048 pages
    (rank 0's GPU only), each with the one-core CPU figure of the same page."""
    from oracle import oracle as orc
    from parquet4seastar_amd import pagegen
    from parquet4seastar_amd.pagegen import synth
    counts = [int(x) for x in args.sweep_pages.split(",") if x]
    one = Shard(0, 1, True)
    solo = DistCtx(1, None)
    n = int(MI * args.scale) // 8 * 8
    w = 12
    vals = [synth.run_values(rng, n, w).astype(np.uint32) for _ in range(8)]
    pages = [pagegen.rle_encode(v, w) for v in vals]
    rcpu = rle_cpu(orc, pages[0], w, n) if cpu else None
    nd = int(131072 * args.scale) // 1024 * 1024
    dvals = [np.cumsum(rng.integers(-100, 3996, size=nd, dtype=np.int64)) for _ in range(8)]
    dpages = [pagegen.delta_encode(v, 256, 8, as_array=True) for v in dvals]
    dcpu = None
    if cpu:
        dt = cpu_time(lambda: orc.delta_decode(bytes(dpages[0]), nd, 8))
        dcpu = {"GB/s": round((len(dpages[0]) + nd * 8) / dt / 1e9, 3), "values_per_sec": round(nd / dt, 1), "cores": 1,
                "kind": "port", "sample": "one page of %d values" % nd}
    dict_page, cpages, dictionary, tv, td = synth.dict_chunk_pages(rng, 65535, "uniform", 0, n, 8)
    ccpu = None
    if cpu:
        sec = pagegen.dict_page_values_encode(rng.integers(0, 65535, size=n).astype(np.uint32), 16)
        dt = cpu_time(lambda: orc.dict_decode(sec, dictionary, n))
        ccpu = {"GB/s": round((len(sec) + 4 * n) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1), "cores": 1,
                "kind": "port", "sample": "the value section of one page (%d values)" % n}
    sweep = {"rle_w12": {}, "delta_int64": {}, "dict_65535_uniform": {}}
    for k in counts:
        tmp = {}
        ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, one, "p4s_decode_rle", pages, k, n, 4, vals, args.enc_steps,
                                                extra=(4,), param0=w, verify=not args.no_verify)
        sweep["rle_w12"][str(k)] = finish_entry("x", tmp, solo, ms, enc, dec, nv, pl, "%d pages x %d values" % (k, n))
        ms, enc, dec, nv, pl = run_fixed_family(ctx, timer, one, "p4s_decode_delta_binary_packed", dpages, k, nd, 8, dvals,
                                                args.enc_steps, extra=(8,), verify=not args.no_verify)
        sweep["delta_int64"][str(k)] = finish_entry("x", tmp, solo, ms, enc, dec, nv, pl, "%d pages x %d values" % (k, nd))
        chunk = np.concatenate([np.frombuffer(dict_page, dtype=np.uint8)] +
                               [np.frombuffer(cpages[i % 8], dtype=np.uint8) for i in range(k)])
        ms, cnt = dict_chunk_decode(timer_of, args, chunk, 0, n, k, tv[:min(k, 8)], td[:min(k, 8)], "sweep dict", not args.no_verify)
        sweep["dict_65535_uniform"][str(k)] = finish_entry("x", tmp, solo, ms, int(chunk.size), 4 * cnt["values"], cnt["values"], None,
                                                           "dictionary page + %d data pages x %d values" % (k, n))
    for key, c in (("rle_w12", rcpu), ("delta_int64", dcpu), ("dict_65535_uniform", ccpu)):
        if c:
            sweep[key]["cpu"] = c
    out["sweep"] = sweep


def bench_snappy(ctx, timer, shard, dctx, args, rng, out, cpu):
    """Snappy page decompression, the step in front of the decoders for compressed chunks.  The inputs are made with
    libsnappy (pyarrow's copy, the library the reference wraps); without pyarrow the entry is left out."""
    try:
        import pyarrow as pa
    except ImportError:
        return
    from parquet4seastar_amd import capi
    codec = pa.Codec("snappy")
    page_bytes = int(MI * args.scale) // 64 * 64
    distinct, total = 16, max(16, int(1024 * args.scale) // 16 * 16)        # 1 GiB decompressed
    works = {}
    # PLAIN int64 pages of the headline's value distribution (running sums of small deltas)
    works["snappy_plain_int64"] = [np.cumsum(rng.integers(-100, 3996, size=page_bytes // 8)).astype(np.int64).tobytes()
                                   for _ in range(distinct)]
    # PLAIN BYTE_ARRAY pages of strings drawn from a vocabulary of 4096 words of 8-24 bytes: many matches
    vocab = [bytes(rng.integers(97, 123, size=int(rng.integers(8, 25)), dtype=np.uint8)) for _ in range(4096)]
    vocab = [len(w).to_bytes(4, "little") + w for w in vocab]
    works["snappy_plain_strings"] = [
        b"".join(vocab[i] for i in rng.integers(0, 4096, size=page_bytes // 12))[:page_bytes] for _ in range(distinct)]
    for name, raw in works.items():
        if not wanted(args, name):
            continue
        comp = [codec.compress(x, asbytes=True) for x in raw]
        pl = shard.plan([len(comp[i % distinct]) for i in range(total)], [page_bytes] * total, 1)
        lo, hi = pl["lo"], pl["hi"]
        n_pages = hi - lo
        d_in, in_bytes, offs, sizes = replicated(ctx, comp, lo, hi)
        descs = capi.make_descs(offs, sizes, [i * page_bytes for i in range(n_pages)], 0, page_bytes)
        d_pages = ctx.to_device(descs.view(np.uint8))
        d_status = ctx.alloc(48 * max(n_pages, 1))
        d_out = ctx.alloc(max(n_pages, 1) * page_bytes)

        def step():
            ctx._check(capi.lib().p4s_decompress_snappy(ctx.handle, d_in.ptr, in_bytes, d_pages.ptr, n_pages, d_out.ptr,
                                                        n_pages * page_bytes, d_status.ptr))

        ms = timer.time(step, max(2, args.enc_steps // 2))
        if not args.no_verify:
            status = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
            capi.raise_on_page_errors(status)
            for i in sorted(set([0, n_pages - 1])):
                got = d_out.download(np.uint8, page_bytes, byte_offset=i * page_bytes)
                if not np.array_equal(got, np.frombuffer(raw[(lo + i) % distinct], dtype=np.uint8)):
                    raise SystemExit("%s: GPU output differs from the compressed bytes' source" % name)
        for bb in (d_in, d_pages, d_status, d_out):
            bb.free()
        e = finish_entry(name, out, dctx, ms, int(sum(sizes)), n_pages * page_bytes, n_pages * page_bytes, pl,
                         "%d pages x %d bytes, compressed by libsnappy to %.0f %%" % (
                             total, page_bytes, 100.0 * sum(len(c) for c in comp) / (distinct * page_bytes)))
        if cpu:
            dt = cpu_time(lambda: codec.decompress(comp[0], decompressed_size=page_bytes, asbytes=True))
            e["cpu"] = {"GB/s": round((len(comp[0]) + page_bytes) / dt / 1e9, 3), "cores": 1, "kind": "reference",
                        "sample": "one page, libsnappy RawUncompress (pyarrow's copy of the library the reference calls)"}


def bench_strings(ctx, timer, shard, dctx, args, rng, out, cpu):
    """Config 5: 64Mi short strings with shared prefixes, as DELTA_LENGTH_BYTE_ARRAY and as DELTA_BYTE_ARRAY."""
    from oracle import oracle as orc
    from parquet4seastar_amd import capi
    from parquet4seastar_amd.pagegen import synth
    n = int(65536 * args.scale) // 8 * 8
    distinct, total = 16, max(16, int(1024 * args.scale) // 16 * 16)          # 1024 pages x 64Ki strings = 64Mi strings
    gen = synth.string_pages(rng, n, distinct)
    L = capi.lib()
    for name, key in (("delta_length_byte_array", "dlba"), ("delta_byte_array", "dba")):
        if not wanted(args, name):
            continue
        pages = [p[key] for p in gen]
        totals = [int(p["offsets"][-1]) for p in gen]
        pl = shard.plan([len(pages[i % distinct]) for i in range(total)], [n] * total, 8)
        lo, hi = pl["lo"], pl["hi"]
        n_pages = hi - lo
        d_in, in_bytes, offs, sizes = replicated(ctx, pages, lo, hi)
        caps = [(totals[(lo + i) % distinct] + 15) // 16 * 16 for i in range(n_pages)]
        aux = np.concatenate([[0], np.cumsum(caps)[:-1]]) if n_pages else np.zeros(0, dtype=np.int64)
        descs = capi.make_descs(offs, sizes, [i * (n + 1) for i in range(n_pages)], n, caps, aux_offsets=aux)
        d_pages = ctx.to_device(descs.view(np.uint8))
        d_status = ctx.alloc(48 * max(n_pages, 1))
        n_slots = n_pages * (n + 1) + 1
        d_offsets = ctx.alloc(n_slots * 8)
        bytes_cap = int(sum(caps)) + 64
        d_bytes = ctx.alloc(bytes_cap)
        d_scratch = ctx.alloc(n_slots * 8)
        if key == "dlba":
            def step():
                ctx._check(L.p4s_decode_delta_length_byte_array(ctx.handle, d_in.ptr, in_bytes, d_pages.ptr, n_pages,
                                                                d_offsets.ptr, d_bytes.ptr, bytes_cap, d_status.ptr))
        else:
            def step():
                ctx._check(L.p4s_decode_delta_byte_array(ctx.handle, d_in.ptr, in_bytes, d_pages.ptr, n_pages, d_offsets.ptr,
                                                         d_bytes.ptr, bytes_cap, d_scratch.ptr, n_slots * 8, d_status.ptr))
        ms = timer.time(step, args.enc_steps)
        status = d_status.download(capi.PAGE_STATUS_DTYPE, n_pages)
        capi.raise_on_page_errors(status)
        if os.environ.get("P4S_BENCH_DUMP_AUX"):       # kernel timing experiments (-DP4S_EXP_TIMING builds)
            a0, a1 = status["aux0"].astype(np.uint64), status["aux1"].astype(np.uint64)
            print(name, "aux cycles per page (mean):", float((a0 & 0xFFFFFFFF).mean()), float((a0 >> 32).mean()),
                  float((a1 & 0xFFFFFFFF).mean()), float((a1 >> 32).mean()), file=sys.stderr)
            if key == "dba" and hasattr(L, "p4s_debug_dba_ticks"):
                import ctypes
                t = (ctypes.c_uint64 * 8)()
                L.p4s_debug_dba_ticks.argtypes = [ctypes.c_void_p]
                L.p4s_debug_dba_ticks(t)
                launches = args.enc_steps + 1
                print("k_dba_assemble cycles per launch and workgroup slot (thread 0):",
                      [round(x / launches / 8192) for x in t], file=sys.stderr)
        out_bytes = int(status["out_bytes"].sum())
        assert int(status["values_decoded"].sum()) == n_pages * n
        if not args.no_verify:
            # against the generated strings, and (one page) against the oracle
            for p in sorted(set([0, n_pages - 1])):
                src = gen[(lo + p) % distinct]
                got_off = d_offsets.download(np.int64, n + 1, byte_offset=8 * p * (n + 1))
                got_bytes = d_bytes.download(np.uint8, totals[(lo + p) % distinct], byte_offset=int(aux[p]))
                if not (np.array_equal(got_off, src["offsets"]) and np.array_equal(got_bytes, src["bytes"])):
                    raise SystemExit("%s: GPU decode differs from the generated strings" % name)
            want_off, want_bytes = (orc.dlba_decode if key == "dlba" else orc.dba_decode)(pages[lo % distinct], n)
            got_off = d_offsets.download(np.int64, n + 1)
            if not np.array_equal(got_off, want_off):
                raise SystemExit("%s: GPU decode differs from the oracle" % name)
        e = finish_entry(name, out, dctx, ms, int(sum(sizes)), out_bytes, n_pages * n, pl,
                         "%d pages x %d strings of 8-24 bytes, about 60%% shared prefix, into int64 offsets + bytes" % (total, n),
                         extra_bytes=8 * n_pages * (n + 1), offset_bytes=True)
        if key == "dba" and dctx.world == 1 and not args.no_e2e:
            alg = int(sum(sizes)) + out_bytes + 8 * n_pages * (n + 1)
            ent, (h_off, h_by) = pcie_inclusive(ctx, d_in, in_bytes, step, [(d_offsets, 8 * n_pages * (n + 1)), (d_bytes, int(sum(caps)))], alg)
            src = gen[(lo + n_pages - 1) % distinct]
            p = n_pages - 1
            ent["verified_last_page"] = bool(
                np.array_equal(h_off.view(np.int64)[p * (n + 1):(p + 1) * (n + 1)], src["offsets"]) and
                np.array_equal(h_by[int(aux[p]):int(aux[p]) + totals[(lo + p) % distinct]], src["bytes"]))
            e["pcie_inclusive"] = ent
        if cpu:
            fn = orc.dlba_decode if key == "dlba" else orc.dba_decode
            dt = cpu_time(lambda: fn(pages[0], n))
            e["cpu"] = {"GB/s": round((sizes[0] + totals[0] + 8 * (n + 1)) / dt / 1e9, 3), "values_per_sec": round(n / dt, 1),
                        "cores": 1, "kind": "port", "sample": "one page of %d strings" % n}
        for b in (d_in, d_pages, d_status, d_offsets, d_bytes, d_scratch):
            b.free()


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from parquet4seastar_amd import capi
    from parquet4seastar_amd.pagegen import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # P4S_BENCH_BACKEND=gloo and P4S_BENCH_DEVICE=<index> let the multi-rank path be rehearsed on a one-GPU box
    backend = os.environ.get("P4S_BENCH_BACKEND", "nccl")
    if "P4S_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["P4S_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decode path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)
    os.environ["P4S_DEVICE"] = str(local_rank)       # the C++ host layer's device
    dev = torch.device("cuda", local_rank)
    strong = args.scaling == "strong"
    shard = Shard(rank, world, strong)
    rdev = dev if backend == "nccl" else torch.device("cpu")      # where the timing reductions live
    dctx = DistCtx(world, rdev)

    # ---- synthetic input: ONE page list that every rank generates identically and takes its shard of (strong), or a
    # list of the rank's own (weak)
    t0 = time.time()
    pages, values = synth.delta_pages(args.pages, args.page_values, args.seed + (0 if strong else 1000 * rank))
    sizes_all = [int(p.size) for p in pages]
    pl = shard.plan(sizes_all, [args.page_values] * args.pages, 8)
    lo, hi = pl["lo"], pl["hi"]
    my_pages = hi - lo
    if my_pages == 0:
        raise SystemExit("rank %d of %d has no page to decode: --pages must be at least the number of GPUs" % (rank, world))
    buf, offs, sizes = pack(pages[lo:hi])
    enc_bytes = int(sum(sizes))
    n_values = my_pages * args.page_values
    dec_bytes = n_values * 8
    my_values = values[lo * args.page_values: hi * args.page_values]
    descs = capi.make_descs(offs, sizes, [int(x) for x in pl["out_offsets"]], args.page_values)
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
    d_status = torch.zeros(48 * my_pages, dtype=torch.uint8, device=dev)
    d_out = torch.empty(n_values, dtype=torch.int64, device=dev)
    L = capi.lib()

    def step():
        ctx._check(L.p4s_decode_delta_binary_packed(ctx.handle, d_in.data_ptr(), buf.size, d_pages.data_ptr(),
                                                    my_pages, d_out.data_ptr(), d_status.data_ptr(), 8))

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
        t = torch.tensor([wall], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())

    # kernel timing experiments (-DP4S_EXP_TIMING builds) report cycle counts in the status records' aux fields
    if os.environ.get("P4S_BENCH_DUMP_AUX"):
        stt = np.frombuffer(d_status.cpu().numpy().tobytes(), dtype=capi.PAGE_STATUS_DTYPE)
        a0, a1 = stt["aux0"].astype(np.uint64), stt["aux1"].astype(np.uint64)
        parts = [a0 & 0xFFFFFFFF, a0 >> 32, a1 & 0xFFFFFFFF, a1 >> 32]
        print("aux cycles per page (mean):", [float(x.mean()) for x in parts], file=sys.stderr)

    # ---- parity of what was just timed
    verified = None
    if not args.no_verify:
        status = np.frombuffer(d_status.cpu().numpy().tobytes(), dtype=capi.PAGE_STATUS_DTYPE)
        capi.raise_on_page_errors(status)
        assert int(status["values_decoded"].sum()) == n_values
        want = torch.from_numpy(np.ascontiguousarray(my_values)).to(dev)
        verified = bool(torch.equal(d_out, want))
        del want
        if not verified:
            raise SystemExit("GPU decode differs from the generated column")
        if world > 1:
            (n_ok,) = dctx.sum_int(1)
            verified = n_ok == world

    total_bytes_all = pl["total_bytes"]                 # the whole list (strong) or every rank's list (weak)
    total_values_all = pl["total_values"]
    ms_per_step = wall / args.steps * 1e3
    value_gbs = total_bytes_all / (wall / args.steps) / 1e9
    avg_kernel_ms = float(np.mean(kern_ms))
    achieved = (enc_bytes + dec_bytes) / (avg_kernel_ms * 1e-3) / 1e9
    workload = ("DELTA_BINARY_PACKED int64, %d pages x %d values %s (block 256 / 8 miniblocks, deltas uniform [-100,3996))"
                % (args.pages, args.page_values,
                   "per GPU" if not strong and world > 1 else ("partitioned over %d GPUs" % world if world > 1 else "on one GPU")))

    result = {
        "metric": "device-resident decoded GB/s (DELTA_BINARY_PACKED int64; encoded+decoded bytes per second)",
        "value": round(value_gbs, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "pages_total": args.pages * (1 if strong else world),
            "pages_this_rank": my_pages,
            "values_total": total_values_all,
            "encoded_bytes_this_rank": enc_bytes,
            "decoded_bytes_this_rank": dec_bytes,
            "sharding": "contiguous byte-balanced shards of one page list, no collective" if strong else "a page list per rank, no collective",
        },
        "values_per_sec": round(total_values_all / (wall / args.steps), 1),
        "verified_bit_exact": verified,
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic_of("delta_int64_%dx%d" % (my_pages, args.page_values)),
            "kernel": "k_delta_binary_packed<uint64_t>",
            "avg_kernel_ms": round(avg_kernel_ms, 4),
            "algorithmic_bytes_per_launch": enc_bytes + dec_bytes,
            "what": "rank 0's kernel: its shard's encoded + decoded bytes over the average launch duration (HIP events on the launch stream)",
        },
        "gen_seconds": round(gen_s, 2),
    }

    cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if cpu:
        from oracle import oracle as orc
        # bounded sample: pages of the rank-0 shard decoded once, single thread, by the C restatement of the
        # reference's per-value loop (src/encoding.cc:207-239); gcc -O2, then the -O3 build where the host can run it
        n_sample = min(my_pages, 512)
        out = np.empty(args.page_values, dtype=np.int64)
        info = cpu_info()

        def time_port(lib, count):
            err, left = orc.OrcError(), C.c_int64(0)
            t0 = time.perf_counter()
            nb = 0
            for p in range(count):
                pg = pages[lo + p]
                got = lib.orc_delta_decode(pg.ctypes.data, pg.size, out.ctypes.data, 8, args.page_values,
                                           C.byref(left), C.byref(err))
                assert got == args.page_values
                nb += pg.size + got * 8
            return nb, time.perf_counter() - t0

        lib = orc.lib()
        sample_bytes, dt = time_port(lib, n_sample)
        assert np.array_equal(out, my_values[(n_sample - 1) * args.page_values: n_sample * args.page_values])
        result["cpu_baseline"] = {
            "value": round(sample_bytes / dt / 1e9, 3),
            "unit": "GB/s",
            "cores": 1,
            "kind": "port",
            "cpu_model": info["model"],
            "cores_visible": info["cores_visible"],
            "flags": "gcc -O2",
            "sample": "%d of the %d pages (%d values), one thread, C restatement of the reference's "
                      "per-value DELTA loop" % (n_sample, my_pages, n_sample * args.page_values),
            "values_per_sec": round(n_sample * args.page_values / dt, 1),
        }
        lib3 = orc.lib_o3()
        if lib3 is not None:
            b3, dt3 = time_port(lib3, min(n_sample, 256))
            result["cpu_baseline"]["value_O3"] = round(b3 / dt3 / 1e9, 3)
            result["cpu_baseline"]["flags_O3"] = orc.O3_FLAGS
        # the same port on every host core this process may use (one page partition per thread; ctypes releases
        # the GIL during the call): the scope table's second CPU figure
        import concurrent.futures as cf
        n_thr = max(1, min(info["cores_visible"], int(os.environ.get("OMP_NUM_THREADS", "16")), 64))
        n_all = min(my_pages, 64 * n_thr)

        def cpu_part(t):
            o = np.empty(args.page_values, dtype=np.int64)
            e_, l_ = orc.OrcError(), C.c_int64(0)
            nb = 0
            for p in range(t, n_all, n_thr):
                pg = pages[lo + p]
                nb += pg.size + 8 * lib.orc_delta_decode(pg.ctypes.data, pg.size, o.ctypes.data, 8, args.page_values,
                                                         C.byref(l_), C.byref(e_))
            return nb
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(n_thr) as ex:
            all_bytes = sum(ex.map(cpu_part, range(n_thr)))
        dt = time.perf_counter() - t0
        result["cpu_baseline_all_cores"] = {"value": round(all_bytes / dt / 1e9, 3), "unit": "GB/s", "cores": n_thr,
                                            "kind": "port", "flags": "gcc -O2", "cpu_model": info["model"],
                                            "sample": "%d pages, one thread per page partition" % n_all}

    # ---- the same step with the copies the reference's path implies: encoded pages start in (pinned) host
    # memory, decoded values end there.  Reported next to `value`, never as it.
    if not args.no_e2e and world == 1:
        import ctypes as _C
        h_in, h_out = _C.c_void_p(), _C.c_void_p()
        ctx._check(L.p4s_host_alloc(ctx.handle, buf.size, _C.byref(h_in)))
        ctx._check(L.p4s_host_alloc(ctx.handle, dec_bytes, _C.byref(h_out)))
        _C.memmove(h_in.value, buf.ctypes.data, buf.size)

        def e2e_step():
            ctx._check(L.p4s_copy_h2d(ctx.handle, d_in.data_ptr(), h_in.value, buf.size))
            step()
            ctx._check(L.p4s_copy_d2h(ctx.handle, h_out.value, d_out.data_ptr(), dec_bytes))

        e2e_step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            e2e_step()
        ctx.sync()
        e2e_s = (time.perf_counter() - t0) / 3
        got = np.ctypeslib.as_array(_C.cast(h_out.value, _C.POINTER(_C.c_int64)), shape=(n_values,))
        e2e_ok = bool(np.array_equal(got[-args.page_values:], my_values[-args.page_values:]))
        result["pcie_inclusive"] = {
            "GB/s": round((enc_bytes + dec_bytes) / e2e_s / 1e9, 2), "ms_per_step": round(e2e_s * 1e3, 3),
            "verified_last_page": e2e_ok,
            "what": "pinned host -> device copy of the encoded pages, decode, device -> pinned host copy of the values; "
                    "one stream, no overlap, per GPU"}
        # the same with the copies overlapped: eight page groups alternate between two contexts (two streams), so the
        # upload of a group, the decode of the one before and the download of the one before that run together
        ctx2 = capi.Context(local_rank)
        groups = 8 if my_pages % 8 == 0 and my_pages >= 8 else 1
        per = my_pages // groups
        bounds = [offs[g * per] for g in range(groups)] + [buf.size]

        def e2e_overlapped():
            for g in range(groups):
                c = (ctx, ctx2)[g & 1]
                lo_b, hi_b = bounds[g], bounds[g + 1]
                c._check(L.p4s_copy_h2d(c.handle, d_in.data_ptr() + lo_b, h_in.value + lo_b, hi_b - lo_b))
                c._check(L.p4s_decode_delta_binary_packed(c.handle, d_in.data_ptr(), buf.size, d_pages.data_ptr() + 48 * g * per,
                                                          per, d_out.data_ptr(), d_status.data_ptr() + 48 * g * per, 8))
                o_lo = 8 * g * per * args.page_values
                c._check(L.p4s_copy_d2h(c.handle, h_out.value + o_lo, d_out.data_ptr() + o_lo, 8 * per * args.page_values))
            ctx.sync()
            ctx2.sync()

        got[:] = 0
        e2e_overlapped()
        t0 = time.perf_counter()
        for _ in range(3):
            e2e_overlapped()
        ov_s = (time.perf_counter() - t0) / 3
        result["pcie_inclusive"]["overlapped"] = {
            "GB/s": round((enc_bytes + dec_bytes) / ov_s / 1e9, 2), "ms_per_step": round(ov_s * 1e3, 3),
            "verified": bool(np.array_equal(got, my_values)),
            "what": "%d page groups alternating between two streams: upload, decode and download overlap" % groups}
        ctx2.close()
        L.p4s_host_free(ctx.handle, h_in)
        L.p4s_host_free(ctx.handle, h_out)

    # ---- the other encodings: every rank times its shard, rank 0 reports the whole job
    del d_in, d_out, d_pages, d_status
    torch.cuda.empty_cache()
    want = set() if args.encodings == "none" else set(
        ["rle", "bss", "dict", "strings", "delta", "snappy", "sweep"] if args.encodings == "all" else args.encodings.split(","))
    if want:
        t0 = time.time()
        rng = np.random.default_rng(args.seed + 7 + (0 if strong else 1000 * rank))
        encs = {}
        timer = Timer(ctx.handle)
        if "rle" in want:
            bench_rle(ctx, timer, shard, dctx, args, rng, encs, cpu)
        if "delta" in want:
            bench_delta_variants(ctx, timer, shard, dctx, args, rng, encs, cpu)
        if "bss" in want:
            bench_bss(ctx, timer, shard, dctx, args, rng, encs, cpu)
        if "strings" in want:
            bench_strings(ctx, timer, shard, dctx, args, rng, encs, cpu)
        if "dict" in want:
            bench_dict(Timer, shard, dctx, args, rng, encs, cpu)
        if "snappy" in want:
            bench_snappy(ctx, timer, shard, dctx, args, rng, encs, cpu)
        if "sweep" in want and world == 1:
            bench_sweep(ctx, timer, Timer, args, rng, encs, cpu)
        if "plain_int64_copy" in encs and world == 1:
            # fraction of what a plain copy kernel reaches on this GPU, next to the fraction of the nominal peak
            result["roofline"]["copy_kernel_GBs"] = encs["plain_int64_copy"]["GB/s"]
            result["roofline"]["frac_of_copy_kernel"] = round(achieved / encs["plain_int64_copy"]["GB/s"], 4)
        result["encodings"] = encs
        result["encodings_seconds"] = round(time.time() - t0, 1)

    if rank == 0:
        print(json.dumps(result))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
