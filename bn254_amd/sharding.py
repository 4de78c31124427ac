"""Multi-GPU sharding of a verify batch: one process per GPU, contiguous shards, and ONE collective —
an all-gather of the per-item status bytes (RCCL over xGMI when the backend is "nccl").

Every tuple is independent (/root/reference/src/ecdsa.rs:49-64 shares no state), so there is no
data-path exchange; constant tables are replicated per device.
"""


def shard_range(n_total, rank, world):
    """contiguous slice [lo, hi) of ceil(n_total / world) items owned by `rank`"""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def gather_status(local_status, n_total=None, out=None):
    """all-gather equal-length uint8 status shards into one tensor (rank-major order).
    local_status: 1-D uint8 torch tensor (padded to the common shard length by the caller)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_status if n_total is None else local_status[:n_total]
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(local_status.numel() * world, dtype=torch.uint8, device=local_status.device)
    dist.all_gather_into_tensor(out, local_status)
    return out if n_total is None else out[:n_total]


def failure_count(status):
    """optional 8-byte all-reduce(sum) of the number of non-Ok items"""
    import torch
    import torch.distributed as dist

    c = (status != 0).sum().to(torch.int64).reshape(1)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return int(c.item())
