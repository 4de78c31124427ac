"""Multi-GPU sharding of a verify / pairing batch: one process per GPU, contiguous shards, and ONE collective —
an all-gather of the per-item status bytes (RCCL over xGMI when the backend is "nccl"); for the pairing workload
(BASELINE config 4) additionally an 8-byte all-reduce of the additive 64-bit checksum over all Gt words.

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

    if not (dist.is_available() and dist.is_initialized()):
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
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return int(c.item())


def gt_checksum(gt_bytes):
    """additive checksum over all little-endian 64-bit words of a uint8 tensor of canonical Gt bytes, mod 2^64
    (returned as a 1-element int64 tensor on the tensor's device; int64 addition wraps like uint64 addition)"""
    import torch

    assert gt_bytes.dtype == torch.uint8 and gt_bytes.numel() % 8 == 0
    return gt_bytes.view(torch.int64).sum().reshape(1)


def allreduce_checksum(local_sum):
    """sum of the per-rank checksums mod 2^64 (8-byte all-reduce); returns a Python int in [0, 2^64)"""
    import torch.distributed as dist

    t = local_sum.clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item()) & 0xFFFFFFFFFFFFFFFF
