"""Read sharding across the GPUs of one node.

Placement is a pure function of (read, DB): reads split into contiguous shards, one rank (process) per GPU, the DB
replicated into every GPU's HBM.  There is NO collective on the data path; the only cross-rank steps are a
max-over-ranks of the elapsed time (bench) and a host-side ordered concat of result records.
"""
import numpy as np

from .placement import Placements


def shard_range(n_items, world_size, rank):
    """Contiguous shard [start, end) of rank; sizes differ by at most one, order preserved across ranks."""
    base, rem = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_reads(seq, seq_off, world_size, rank):
    """Slice a concatenated-read batch to this rank's shard (offsets rebased to 0)."""
    a, b = shard_range(len(seq_off) - 1, world_size, rank)
    lo, hi = int(seq_off[a]), int(seq_off[b])
    return np.ascontiguousarray(seq[lo:hi]), (seq_off[a:b + 1] - seq_off[a]).astype(np.uint64)


def concat_placements(parts):
    """Ordered concat of per-shard results (host side; what the Java side does with jplace records)."""
    parts = list(parts)
    counters = {}
    for p in parts:
        for k, v in p.counters.items():
            counters[k] = counters.get(k, 0) + v
    return Placements(np.concatenate([p.n_rows for p in parts]), np.concatenate([p.branch for p in parts]),
                      np.concatenate([p.score for p in parts]), np.concatenate([p.lwr for p in parts]),
                      np.concatenate([p.flags for p in parts]), counters)


def gather_placements(local, dst=0, group=None):
    """torch.distributed: every rank contributes its shard's Placements, `dst` gets the ordered concat."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bucket = [None] * world if rank == dst else None
    dist.gather_object(local, bucket, dst=dst, group=group)
    return concat_placements(bucket) if rank == dst else None


def max_over_ranks(value, device=None, group=None):
    """Elapsed-time reduction of the bench contract (MAX over ranks)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
