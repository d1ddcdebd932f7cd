"""Instance sharding across the GPUs of one node.

Every problem instance is independent, so a batch is cut into contiguous per-rank shards and each rank solves its
own shard with no data-path collective; the only exchange is the gather of the converged costs (all ranks) and, on
request, of the converged trajectories X / U (to one rank) at the end.  Backend-agnostic: "nccl" (= RCCL over xGMI) on
GPUs, "gloo" in the CPU tests.
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


def gather_trajectories(local, total: int, dst: int = 0, group=None):
    """Gather per-shard trajectory tensors [n_local, ...] (X: [n, T, n_x], U: [n, T-1, n_u]; shards may differ by one instance) to
    rank `dst` in instance order: returns the full [total, ...] tensor there, None on the other ranks.  One point-to-point style
    collective per call: every peer's shard rides its own xGMI link to `dst` (C4: 4096 x 200 x 23 doubles = 151 MB per peer)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank}: local shard has {local.shape[0]} instances, expected {sizes[rank]}")
    pad = max(sizes)
    buf = local
    if sizes[rank] != pad:  # ragged: pad to the common shard size (at most one instance)
        buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[: sizes[rank]] = local
    buf = buf.contiguous()
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gather_list=parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([parts[r][: sizes[r]] for r in range(world)], dim=0)
