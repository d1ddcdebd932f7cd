"""Instance sharding across the GPUs of one node.

Every problem instance is independent, so a batch is cut into contiguous per-rank shards and each rank solves its
own shard with no data-path collective; the only exchange is the gather of the converged costs (and, on request,
status words) at the end.  Backend-agnostic: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) of `total` instances owned by `rank`; shards differ by at most one instance."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_costs(local_cost, total: int, group=None):
    """All-gather per-shard cost vectors (1-D float64 tensors, possibly of unequal length) into the full [total]
    vector, in instance order, on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    if local_cost.numel() != sizes[rank]:
        raise ValueError(f"rank {rank}: local shard has {local_cost.numel()} instances, expected {sizes[rank]}")
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=local_cost.dtype, device=local_cost.device)
    buf[: sizes[rank]] = local_cost
    out = torch.empty(world * pad, dtype=local_cost.dtype, device=local_cost.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return torch.cat([out[r * pad : r * pad + sizes[r]] for r in range(world)])
