"""Partitioning a column chunk's page list across GPUs (one process per GPU).

Pages are independent (the reference builds a fresh decoder per page, src/encoding.cc:526-586;
only the read-only dictionary is shared), so the data path has no collective: every rank decodes a
contiguous shard of the page list into its own HBM, and a page's global output offset is the
exclusive prefix sum of the header value counts, which every rank can compute alone.
torch.distributed carries only barriers and the reduction of timings.
"""
import numpy as np


def shard_bounds(weights, world):
    """Contiguous shards with balanced total weight (weight = encoded + decoded bytes of a page).
    Returns world + 1 page indices; rank r owns pages [b[r], b[r + 1])."""
    w = np.asarray(weights, dtype=np.float64)
    if world <= 0:
        raise ValueError("world must be positive")
    if w.size == 0:
        return [0] * (world + 1)
    cum = np.cumsum(w)
    targets = cum[-1] * np.arange(1, world) / world
    cuts = np.searchsorted(cum, targets, side="left") + 1
    # a page goes to the side whose boundary its midpoint is closer to
    mids = cum - w / 2
    cuts = np.array([c - 1 if c - 1 >= 0 and mids[c - 1] > t else c for c, t in zip(cuts, targets)], dtype=np.int64)
    bounds = np.concatenate([[0], np.clip(cuts, 0, w.size), [w.size]])
    return [int(x) for x in np.maximum.accumulate(bounds)]


def global_output_offsets(num_values):
    """Global first-value index of every page: exclusive prefix sum of the header value counts."""
    n = np.asarray(num_values, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(n)[:-1]]) if n.size else n


def whole_job_rate(units_local, seconds_local, dist=None):
    """(units of all ranks, max seconds over ranks, units per second).  `dist` = torch.distributed
    (initialised) or None for a single process."""
    units, seconds = float(units_local), float(seconds_local)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        u = torch.tensor([units], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        units, seconds = float(u.item()), float(t.item())
    return units, seconds, units / seconds


def shard_plan(encoded_bytes, num_values, elem_size, world, rank):
    """What rank `rank` of `world` decodes of a page list that every rank knows (page sizes and header value
    counts): a contiguous run of pages balanced on encoded + decoded bytes, and where its values go in the global
    output.  Returns a dict: lo, hi (pages [lo, hi)), first_value (global index of the shard's first value),
    values (of the shard), out_offsets (per page of the shard, relative to the shard's own output buffer),
    bytes (encoded + decoded bytes of the shard), total_bytes / total_values (of the whole list)."""
    enc = np.asarray(encoded_bytes, dtype=np.int64)
    nv = np.asarray(num_values, dtype=np.int64)
    weights = enc + nv * int(elem_size)
    b = shard_bounds(weights, world)
    lo, hi = b[rank], b[rank + 1]
    offs = global_output_offsets(nv)
    first = int(offs[lo]) if lo < len(offs) else int(nv.sum())
    return {"lo": lo, "hi": hi, "first_value": first, "values": int(nv[lo:hi].sum()),
            "out_offsets": (offs[lo:hi] - first).astype(np.int64), "bytes": int(weights[lo:hi].sum()),
            "total_bytes": int(weights.sum()), "total_values": int(nv.sum())}
